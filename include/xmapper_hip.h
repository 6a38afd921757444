/* xmapper_hip.h — C ABI of libxmapper_hip.so: the MI355X-native drop-in for X-Mapper's per-read seed-and-extend path.
 *
 * The reference (mathjeff/Mapper, 100 % Java) has no FFI seam; the narrowest stable boundary on this path is
 *     AlignerWorker.align(Query) -> QueryAlignments          src/main/java/mapper/AlignerWorker.java:256-261
 *     (batch form: the loop of AlignerWorker.process()        src/main/java/mapper/AlignerWorker.java:177-231)
 * which is also the public Api.align(Query, ReferenceDatabase, AlignmentParameters, Logger)   Api.java:79-92.
 * Every entry point below cites the reference code it replaces.  INTEGRATION.md shows the JNI binding a maintainer
 * would add on the Java side.  Plain pointers and sizes only; no Java, torch or C++ types cross this boundary.
 *
 * Bases are 4-bit IUPAC masks, one per byte: A=1 C=2 G=4 T=8, ambiguity = OR (QuickVariants Basepairs encoding, see
 * HashBlock_Matcher.java:184-196).  Contigs are passed forward-only in the order Mapper.sortAndComplementReference
 * produces (Mapper.java:1151-1172: length-descending); the reverse complements are implied.
 *
 * Errors: like the reference (any worker exception aborts the run, AlignerWorker.java:195-197, Mapper.java:1070-1077)
 * a failing call returns non-zero, produces no partial result, and xm_last_error() describes it.
 * Threading: the reference shares one HashBlock_Database between all AlignerWorker threads through per-thread views
 * (HashBlock_Database.java:129-133, Mapper.java:1026-1040; Api.java:78 "thread-safe").  Here an xm_index handle is one such view - a CONTEXT:
 * its own HIP stream, batch buffers, scratch and result pool over tables that every context of the index shares (host tables: all contexts;
 * tables in HBM: the contexts of one GPU).  Calls on one context serialise; contexts run side by side, one host thread each
 * (xm_context_new for a further context on the same GPU, xm_index_replicate for one on another GPU).  Tables that grow
 * (xm_index_ensure_length, a batch with longer mates) grow for every context; the launches that read them are waited for.
 */
#ifndef XMAPPER_HIP_H
#define XMAPPER_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* AlignmentParameters (AlignmentParameters.java:8-35).  StartingInsertionStartFree is always false at entry. */
typedef struct xm_params {
  double MutationPenalty, InsertionStart_Penalty, InsertionExtension_Penalty, DeletionStart_Penalty, DeletionExtension_Penalty,
      MaxErrorRate, UnalignedPenalty, AmbiguityPenalty, Max_PenaltySpan;
  int32_t MaxNumMatches;
  int32_t reserved;
} xm_params;

typedef struct xm_ref {
  int32_t num_contigs;
  const char* const* names;      /* may be NULL */
  const uint8_t* const* codes;   /* [num_contigs] forward strand */
  const int64_t* lengths;        /* [num_contigs] */
} xm_ref;

/* How the reference database is assembled (Mapper.java:657-692 vs Api.java:41-69). */
typedef struct xm_build_opts {
  int32_t enable_gapmers;        /* 1 (default); 0 = --no-gapmers (Mapper.java:245) */
  int32_t min_interesting_size;  /* <=0: (int)max(log4(N+1)-2, 1)  (HashBlock_Database.java:52) */
  int32_t max_hashed_length;     /* <=0: chooseMaxDuplicationLength = 2*ceil(log2 N) (DuplicationDetector.java:17-36); grows on demand */
  int32_t dup_window;            /* 1000 for Mapper.run (Mapper.java:691), 1 for Api.newDatabase (Api.java:66) */
  int32_t dup_min_copies;        /* 2 */
  int32_t dup_min_length;        /* <=0: chooseMinDuplicationLength */
  int32_t dup_max_length;        /* <=0: chooseMaxDuplicationLength */
  int32_t device;                /* HIP device ordinal; <0: current device */
  int32_t host_only;             /* 1: build the host-side tables only (no GPU touched): index inspection on CPU-only machines */
  int32_t reserved;
} xm_build_opts;

typedef struct xm_index xm_index;

/* A batch of queries = the List<QueryBuilder> one AlignerWorker.process() call consumes (AlignerWorker.java:177). */
typedef struct xm_query_batch {
  int64_t num_queries;
  const int32_t* mate_count;     /* [nq] 1 or 2 (Query.getNumSequences()) */
  const int64_t* mate_offset;    /* [2*nq] offset of each mate in `codes` */
  const int32_t* mate_length;    /* [2*nq] */
  const uint8_t* codes;          /* mates as given in the FASTQ (mate 2 NOT reverse-complemented) */
  int64_t codes_length;
  const double* expected_inner;  /* [nq] Query.getExpectedInnerDistance() (paired only; --spacing, Mapper.java:34) */
  const double* deviation;       /* [nq] Query.getSpacingDeviationPerUnitPenalty() (Mapper.java:35) */
} xm_query_batch;

/* Results = List<QueryAlignments> (AlignerWorker.java:231,652-656) flattened into two streams.
 * For query q, ints[int_off[q] .. int_off[q+1]) and dbls[dbl_off[q] .. dbl_off[q+1]) hold, in this order:
 *   ints: numComponents (1; 2 when a pair fell back to unpaired alignments, AlignerWorker.java:643)
 *         per component: numAlignments
 *           per alignment (= QueryAlignment ctor args, QueryMatch_Aligner.java:267): innerDistance, numSequences
 *             per sequence (= SequenceAlignment): contigIndex, referenceReversed, numBlocks,
 *               per block (= AlignedBlock): startA, startB, lengthA, lengthB
 *   dbls:   per alignment: spacingPenalty, overlapMultiplier, duplicationBonus, totalPenalty
 *             per sequence: totalPenalty, alignedPenalty
 * An unaligned query is one component with zero alignments (QueryAlignments.unaligned, AlignerWorker.java:480). */
typedef struct xm_result {
  int64_t num_queries, num_ints, num_dbls;
  int32_t* ints;
  double* dbls;
  int64_t* int_off;  /* [nq+1] */
  int64_t* dbl_off;  /* [nq+1] */
  /* counters of this call: 0 reads, 1 bucket-header probes (PackedMap.getNumMatchesLowerBound), 2 bucket fetches (PackedMap.get),
   * 3 positions fetched, 4 candidates extended (QueryMatch_Aligner.doAlign), 5 PathAligner calls, 6 PathAligner nodes, 7 quick accepts,
   * 8 alignments written, 9 reference-window bytes (4-bit), 10 read bytes (4-bit), 11 reads rerun with a larger scratch scale,
   * 12-15 kernel microseconds by pass: 12 wave-per-read light tier (+ lane-per-read light pass of what the wave form left), 13 wave-per-read
   * chain tier, 14 wave-per-read search tier, 15 lane-per-read gapped / rerun passes */
  int64_t counters[16];
  double kernel_ms;   /* sum of the align (and search) kernels' launch durations (HIP events on the launch stream) */
  double h2d_ms, d2h_ms;  /* batch upload; prefix sums + query-order gather + copy of the four streams to the host */
  int32_t kernel_launches;  /* align + search kernel launches of this call */
  int32_t reserved;
  int64_t prof[16];   /* diagnostic builds (-DXM_PROFILE=1: summed over lanes, =2: per wave) only: shader-clock ticks per phase; otherwise 0 */
  /* (appended in ABI version 2; xm_abi_version())  the rejection filter in front of PathAligner (batches of long reads): 0 searches it examined, 1 searches it
   * proved null without running them (PathAligner.java:169: the search would have returned null after exploring every node within the budget; their nodes
   * are not in counters[6]), 2 cells of the bounding recurrence it computed, 3 = 1 when a pass of this call ran with the filter; 4 pieces (BlockAligner.alignPiece,
   * BlockAligner.java:215-249) the filter examined, 5 pieces it proved unalignable within their budget before their chain ran (their PathAligner calls and nodes are in neither counters[5] nor
   * counters[6]); 6-7 reserved (0) */
  int64_t extra[8];
} xm_result;

typedef struct xm_index_info_t {
  int32_t num_contigs, min_interesting_size, max_hashed_length, enable_gapmers, dup_window, position_bytes;
  int64_t total_forward_size, index_bytes, num_positions;
  double dup_granularity;
  int32_t built_on_device;       /* 1: the tables were hashed on the GPU (xm_index_device.hip), 0: by the host builder (or read from a file) */
  int32_t bucket_line_bytes;     /* 32 / 64: the index has bucket lines of that size in HBM (a probe is one access: xm_seed_probe_lines_kernel); 0: CSR arrays only (xm_seed_probe_kernel) */
  double hash_seconds;           /* wall time of hashing the reference into tables (all calls so far), */
  double duplication_seconds;    /* and of the duplication map */
} xm_index_info_t;

const char* xm_last_error(void);
/* First 16 hex digits of the SHA-256 over the library's sources (every .h and .hip file of mapper_amd/csrc in name order, then this header) at build
 * time: lets a caller check that the loaded library was built from the sources it sits beside. */
const char* xm_build_stamp(void);
/* Version of this header's structs and entry points: 2 = xm_result.extra[] appended, xm_seed_probe_packed replaces xm_seed_probe.  A binding checks it once
 * after loading the library (mapper_amd/_capi.py, bindings/java/xmapper_jni.c). */
int32_t xm_abi_version(void);
/* Page-locked host memory this process holds through the library's pool of result buffers (in use + kept for reuse; at most 4 GiB are kept idle), and - through
 * high_water, if not NULL - the most it ever held.  Eight ranks of a node, each with several contexts, all pin host memory: bench.py prints the mark per rank. */
int64_t xm_pinned_host_bytes(int64_t* high_water);
int xm_device_count(void);

/* Replaces new SequenceDatabase + new HashBlock_Database(...).prepare() + new DuplicationDetector(...).helpSetup()
 * (Mapper.java:657-692, Api.java:41-69, HashBlock_Database.java:490-665, PackedMap.java:54-153, DuplicationDetector.java:97-436)
 * and uploads the result to HBM. */
int xm_index_build(const xm_ref* ref, const xm_build_opts* opts, xm_index** out);
/* A further context of a built index (see "Threading" above).  xm_context_new: on the source's GPU; host tables and tables in HBM are the
 * source's own (no copy of either: a 3 Gb reference's ~130 GB of tables and bucket lines exist once per GPU however many contexts align on it).
 * xm_index_replicate: on `device`; the host tables are shared, and when `device` is another GPU the tables are copied from the source's HBM
 * (hipMemcpyPeer: xGMI between GPUs) instead of being built or uploaded again (device == the source's: the same as xm_context_new).  The
 * reference shares one HashBlock_Database between all AlignerWorkers of a run (Mapper.java:912-1134); one copy per GPU is that sharing here
 * (index replicated, reads sharded, no collective).  Every handle is released with xm_index_free, in any order: the
 * shared tables go with the last one. */
int xm_context_new(xm_index* source, xm_index** out);
int xm_index_replicate(xm_index* source, int32_t device, xm_index** out);
/* Upper limit of the scratch (HBM) a context may allocate for its passes (0: the default, up to 200 GiB).  A context never takes more than 3/4
 * of what is free when it sizes its scratch, and settles for less when the allocation fails; callers that run several contexts on one GPU divide
 * what xm_device_memory reports as free (after the index is resident) between them.  No reference counterpart (the JVM's -Xmx is the analogue). */
int xm_context_set_scratch(xm_index* context, int64_t bytes);
int xm_device_memory(int32_t device, int64_t* free_bytes, int64_t* total_bytes);
/* Binary index cache, in the spirit of --cache-dir (DirCache.java:19-60, HashBlock_Database.java:106-114,477-487, PackedMap.java:249-279:
 * the reference writes one "length-<n>" file per PackedMap under a directory keyed by its property map).  xm_index_save writes the
 * reference, every table hashed so far and the duplication map into ONE file (beside `path`, then renamed: concurrent writers are safe).
 * xm_index_load reads it back and uploads it; with `ref` non-null the file must hold exactly that reference and have been built with the
 * settings `opts` asks for (the reference's cache keys: enableGapmers, minInterestingSize, maxNumShortMatches, format version; plus the
 * duplication settings), else the call fails and the caller builds.  opts->max_hashed_length beyond the file's grows the tables. */
int xm_index_save(xm_index* index, const char* path);
int xm_index_load(const char* path, const xm_ref* ref, const xm_build_opts* opts, xm_index** out);
/* Readable_HashBlock_Database.getContainingMap's lazy growth (Readable_HashBlock_Database.java:108-113): hash tables
 * through gapmers that use `length` bases.  xm_align_batch calls this itself for the longest mate of the batch. */
int xm_index_ensure_length(xm_index* index, int32_t length);
void xm_index_free(xm_index* index);
int xm_index_get_info(const xm_index* index, xm_index_info_t* info);
/* Diagnostics: buckets of all hashed tables, buckets that hold at least one position, buckets marked overfull (more than max(L^2, 5) entries: their positions
 * are dropped, HashBlock_Database.java:569-577) - what a repeat-rich genome does to the index. */
int xm_index_bucket_stats(const xm_index* index, int64_t* buckets, int64_t* occupied, int64_t* overfull);
/* inspection (parity tests): one PackedMap as (counts per bucket or -1 if overfull, concatenated encoded positions) */
int xm_index_table_info(const xm_index* index, int32_t used_length, int32_t* capacity, int32_t* max_count_per_key, int64_t* num_stored, int64_t* num_overfull);
int xm_index_table_dump(const xm_index* index, int32_t used_length, int32_t* counts, int64_t* positions);
/* the PackedMap of one gapmer length: its keyCapacity and per-key limit (HashBlock_Database.java:569-577,620-665) without walking the buckets */
int xm_index_table_shape(const xm_index* index, int32_t used_length, int32_t* capacity, int32_t* max_count_per_key);
int64_t xm_index_dup_keys(const xm_index* index, int32_t contig, int32_t* out, int64_t cap);

/* Replaces the per-read loop of AlignerWorker.process() / Api.align (AlignerWorker.java:177-231,306-644): every query is
 * aligned on the GPU; *out is allocated by the library (its four streams are pinned host buffers from a pool the library keeps) and
 * released with xm_result_free.  Reads may contain IUPAC ambiguity codes (at most 128 ambiguous bases per mate). */
int xm_align_batch(xm_index* index, const xm_params* params, const xm_query_batch* batch, xm_result** out);
void xm_result_free(xm_result* result);
/* The same in two steps, for callers that keep a batch in HBM (and for measuring the path without the PCIe copy):
 * xm_batch_upload validates and copies the batch to the device, xm_align_resident aligns the resident batch. */
int xm_batch_upload(xm_index* index, const xm_query_batch* batch);
int xm_align_resident(xm_index* index, const xm_params* params, xm_result** out);
/* The reference's workers take the next batch of queries while the previous one is being aligned (AlignerWorker.run / requestMoreWork,
 * AlignerWorker.java:92-175).  Here: xm_batch_stage copies the NEXT batch into a second set of device buffers on its own stream and may
 * run (from another host thread) while xm_align_resident is aligning the resident batch; xm_batch_commit then makes the staged batch
 * the resident one (it waits for a running xm_align_resident).  xm_batch_upload = stage + commit without the overlap. */
int xm_batch_stage(xm_index* index, const xm_query_batch* batch);
int xm_batch_commit(xm_index* index);

/* Bulk form of Readable_HashBlock_Database.getNumMatchesLowerBound + matchBlock / PackedMap.get (PackedMap.java:160-172,
 * 228-236) for n (used_length, lookup key) pairs: counts[i] = number of stored positions, -1 when the bucket is overfull or
 * holds more than the table's limit, -2 when used_length is not a hashed length.  Positions (at most max_per_probe <= 15 per probe, not reverse-complemented): the
 * probes of a chunk of 64 consecutive ones (i = 64c ... 64c + 63: a wavefront's) store theirs one behind the other, in probe order, from
 * out_positions[64c * max_per_probe] on; probe i's start within its chunk is the sum of min(max(counts[i'], 0), max_per_probe) over the chunk's probes
 * before it (out_positions holds n * max_per_probe entries; what no probe fills is not written).  Packed because the outputs are most of what such a
 * kernel moves: rows of max_per_probe slots would be 56 bytes a probe at 7 slots, of which a genome's buckets fill ~12.  Device-resident micro-kernel used for
 * the seed-lookup roofline measurement. */
int xm_seed_probe_packed(xm_index* index, int64_t n, const int32_t* used_length, const int32_t* keys, int32_t max_per_probe, int32_t* counts, int64_t* out_positions, double* kernel_ms);
/* (ABI version 2: this entry was xm_seed_probe with out_positions[j * n + i]; the packed layout has its own name so that a caller built against the old
 * header fails to link instead of reading garbage.  out_positions may be NULL: the kernel still fetches the positions - that is what is measured - and only
 * the counts are copied back.) */

/* Measurement helper for the seed-lookup roofline (SURVEY.md section 8d asks for the achieved rate next to "a measured random-64 B-gather
 * ceiling on the same GPU"): `accesses` reads of one random 64-byte sector each from a zero-filled device table of table_bytes;
 * kernel_ms = best of two timed launches.  No reference counterpart. */
int xm_measure_random_gather(int device, int64_t table_bytes, int64_t accesses, double* kernel_ms);

/* Pile-up of the alignments on the reference: what MatchDatabase.addAlignments / groupByPosition hand the
 * mutation and VCF writers (Mapper.java:700-708,758-785; behaviour pinned by MatchDatabase_Test.java:12-69 and MutationsWriter_Test.java:18-134).
 * xm_pileup_add_last accumulates, on the device, the alignments of the index's last xm_align_batch / xm_align_resident call (result streams and
 * batch still in HBM): per forward reference position the depth and the counts of differing query bases (A, C, G, T), in integer units of
 * 1 / XM_PILEUP_UNIT read bases (a query with n alignments adds 1/n per alignment; the mates of a pair add 1/2 each where they overlap), and one
 * event per insertion / deletion block: eight int64 = contig, position (startB of the block), type (1 insertion, 2 deletion), length, query
 * ordinal (over all batches added), mate | reversed << 1 | near-query-end << 2, startA, weight.  Several GPUs: one pile-up per replica, summed by the host in rank order. */
#define XM_PILEUP_UNIT 1441440ull
typedef struct xm_pileup xm_pileup;
int xm_pileup_new(xm_index* index, xm_pileup** out);
/* MatchDatabase(queryEndFraction) (Mapper.java:76,351-353,700; --distinguish-query-ends, default 0.1 in Mapper.main): set before the first add.  With a
 * fraction > 0 the pile-up also keeps the "middle" depth - the depth from query bases that are not within that fraction of the query's length of
 * either query end - which is what the indel thresholds of the writers look at (Mapper.java:532-542, pinned by MutationsWriter_Test.java:114-134),
 * and events carry bit 2 of their flags word when they lie near a query end.  xm_pileup_read_middle reads it (fraction 0: equal to the depth). */
int xm_pileup_set_query_ends(xm_pileup* pileup, double fraction);
int xm_pileup_read_middle(xm_pileup* pileup, int32_t contig, int64_t first, int64_t n, uint64_t* depth);
int xm_pileup_add_last(xm_pileup* pileup, int64_t* num_events);
int xm_pileup_read(xm_pileup* pileup, int32_t contig, int64_t first, int64_t n, uint64_t* depth, uint64_t* alt /* [4][n] */);
int64_t xm_pileup_events(xm_pileup* pileup, int64_t first, int64_t n, int64_t* out /* [8 * n] */);
void xm_pileup_free(xm_pileup* pileup);

/* TEST-ONLY entry (tests/test_gpu_kat.py; not part of the drop-in boundary): the reference's component-level known-answer tests run by the
 * device code of the align kernels over two given texts.  chain 0 = PathAligner alone (PathAligner_Test.java:10-39): mode 0 the lane-per-read
 * search in the wave's LDS slot, 1 the same search in HBM mode, 2 the wave-cooperative search with the search kernel's capacities, 3 with the
 * capacities of the chain tiers' inline searches, 4 the lane-private form of the search (xm_wsearch.h: what the chains of long reads and the reruns use).  chain 1 = HashBlock_Aligner -> StraightAligner -> PathAligner_Runner
 * (HashBlockAligner_Test.java:10-48): mode 0 searches slot-first as in the kernel, 1 all searches in HBM mode, 4 all in the lane-private form.
 * Returns 0 with blocks[4 * num_blocks] = (startA, startB, lengthA, lengthB) and penalties[2] = (total, aligned); 1 = no alignment (null); -1 = error. */
int xm_test_local_align(int32_t device, int32_t chain, int32_t mode, const xm_params* params, const uint8_t* query, int32_t query_length, const uint8_t* reference,
                        int32_t reference_length, double max_ins_ext, double max_del_ext, int32_t block_cap, int32_t* blocks, int32_t* num_blocks, double* penalties,
                        int64_t* nodes_put);
/* Test-only: what the rejection filter in front of PathAligner (mapper_amd/csrc/xm_bound.h) did in this thread's last xm_test_local_align call with mode + 8
 * (the search behind the filter): searches it took, searches it proved null, cells of its recurrence. */
void xm_test_bound_counters(int64_t* out3);
/* Test-only: the rejection filter alone on one problem - query[start_a, end_a) (query_rc: of the reverse complement of `query`) against
 * reference[start_b, end_b), the search's predictedBestOffset - run by one lane (pair: by the two lanes of a pair) of a wave, as the gapped passes of
 * long reads run it in front of PathAligner.align (PathAligner.java:55-293).  out3: 1 if the filter takes the problem, 1 if it proves the search null
 * (PathAligner.java:169), cells it computed.  tests/test_gpu_bound.py compares with the oracle's observer of the same bound, which also runs the search. */
int xm_test_bound(int32_t device, const xm_params* params, const uint8_t* query, int32_t query_length, int32_t query_rc, int32_t start_a, int32_t end_a, const uint8_t* reference,
                  int32_t reference_length, int32_t start_b, int32_t end_b, int32_t predicted_best_offset, int32_t pair, int64_t* out3);


#ifdef __cplusplus
}
#endif
#endif

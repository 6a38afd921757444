/* libxm_hostio.so - host-side I/O of the standalone harness (python -m mapper_amd): FASTA / FASTQ in, SAM out, in native code.
 *
 * NOT part of the drop-in boundary (include/xmapper_hip.h is): in the drop-in deployment the Java host keeps reading the queries and writing the
 * SAM / VCF files (BASELINE.json north_star: "all I/O stays Java"; Mapper.java:699-732 hands every batch of QueryAlignments to its writers).  This
 * library exists so that the Python harness of SURVEY.md section 8(f) rank 1 can feed and drain a kernel that aligns millions of reads per second:
 * reads go from the file buffer to the flat batch arrays of xm_query_batch without an object per read, result streams go to SAM text without an
 * object per alignment.  Formats: what SamWriter_Test.java:18-94 pins, otherwise the SAM specification ([unpinned] in mapper_amd/sam.py, whose
 * records() xmio_write_batch reproduces byte for byte). */
#ifndef XMAPPER_HOSTIO_H
#define XMAPPER_HOSTIO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct xmio_reader xmio_reader;

/* One batch of queries: the arrays of xm_query_batch (mate 2 of a single query has length 0) + the names (one per mate: name_off[2 * q + m] ..
 * name_off[2 * q + m + 1] in `names`) and, when the reader keeps them, the quality strings the same way (has_qual[2 * q + m] = 1 for a FASTQ record). */
typedef struct xmio_batch {
  int64_t num_queries;
  const int32_t* mate_count;
  const int64_t* mate_offset;
  const int32_t* mate_length;
  const uint8_t* codes;
  int64_t codes_length;
  const char* names;
  const int64_t* name_off;   /* [2 * nq + 1] */
  const char* quals;         /* null unless the reader was opened with keep_qualities */
  const int64_t* qual_off;
  const uint8_t* has_qual;
  void* store;               /* owner of the arrays (xmio_batch_free) */
} xmio_batch;

/* Running totals of Mapper.run's statistics lines (Mapper.java:786-796), accumulated over the batches of a job in query order. */
typedef struct xmio_stats {
  int64_t num_queries, num_aligned, total_aligned_length, num_indels;
  double total_penalty;
} xmio_stats;

const char* xmio_last_error(void);
/* path2 != NULL: paired files read in step (a pair = one query).  split_past_size > 0: --split-queries-past-size (SequenceSplitter.java:9-38: a read longer
 * than that becomes equal sections, each a query of its own; single files only).  Files may be gzip-compressed. */
xmio_reader* xmio_open(const char* path1, const char* path2, int32_t split_past_size, int32_t keep_qualities);
/* The next up to max_queries queries; *out = NULL at the end of the input.  Returns 0, or -1 on a malformed file (xmio_last_error). */
int xmio_next(xmio_reader* reader, int64_t max_queries, xmio_batch** out);
void xmio_batch_free(xmio_batch* batch);
void xmio_close(xmio_reader* reader);
/* The result streams of `batch` (xm_result.ints / dbls / int_off / dbl_off) as SAM records (no header) to sam_fd and its unaligned queries (FASTQ when
 * they came with qualities, else FASTA) to unaligned_fd; either descriptor may be -1.  Query order; `threads` formatting threads.  Adds to *stats. */
int xmio_write_batch(const xmio_batch* batch, const int32_t* ints, const double* dbls, const int64_t* int_off, const int64_t* dbl_off, int32_t num_contigs,
                     const char* const* contig_names, int32_t sam_fd, int32_t unaligned_fd, int32_t threads, xmio_stats* stats);
/* Double.toString(x) as the AS:f: / cs:f: tags print it (test hook); returns the length, -1 if cap is too small. */
int xmio_java_double(double x, char* out, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif

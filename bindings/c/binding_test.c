/* binding_test.c — libxmapper_hip.so used from plain C through the JNI shim's marshalling functions (bindings/java/xmapper_jni.c, part 1):
 * what a foreign binder of include/xmapper_hip.h does, without Python and without a JVM.
 *
 *   binding_test symbols             dlopen-free link check: every entry point resolves, xm_build_stamp() and xm_device_count() answer (no GPU needed)
 *   binding_test align IN OUT        reads a reference and a batch from IN (little-endian, written by tests/test_c_binding.py), builds the index with
 *                                    xm_index_build, aligns with xm_align_batch, writes the four result streams to OUT, frees with xm_result_free /
 *                                    xm_index_free (needs a GPU); the Python test compares OUT with the CPU oracle's streams bit for bit
 *   binding_test errors              the error contract: null arguments and a bad batch return non-zero and xm_last_error() says why
 *
 * Replaces, seen from the reference: AlignerWorker.process()'s loop (AlignerWorker.java:177-231) around align(Query) (:256-261).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "xmapper_hip.h"

void xmj_params(const double* nine, int32_t max_num_matches, xm_params* p);
int xmj_build_index(int32_t num_contigs, const uint8_t* const* codes, const int64_t* lengths, const char* const* names, int32_t enable_gapmers,
                    int32_t duplication_window, int32_t max_query_length, int32_t device, xm_index** out);
int xmj_align_batch(xm_index* index, const double* nine, int32_t max_num_matches, int64_t num_queries, const int32_t* mate_count, const int64_t* mate_offset,
                    const int32_t* mate_length, const uint8_t* codes, int64_t codes_length, const double* expected_inner, const double* deviation, xm_result** out);

static void* rd(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) { fprintf(stderr, "binding_test: short input\n"); exit(3); }
  return p;
}
static int64_t rd64(FILE* f) { int64_t v; if (fread(&v, 8, 1, f) != 1) { fprintf(stderr, "binding_test: short input\n"); exit(3); } return v; }

static int do_symbols(void) {
  /* taking the addresses makes the linker resolve every entry point of the header */
  const void* fns[] = {(const void*)xm_last_error, (const void*)xm_build_stamp, (const void*)xm_device_count, (const void*)xm_index_build, (const void*)xm_index_replicate, (const void*)xm_context_new, (const void*)xm_context_set_scratch, (const void*)xm_device_memory, (const void*)xm_index_save,
                       (const void*)xm_index_load, (const void*)xm_index_ensure_length, (const void*)xm_index_free, (const void*)xm_index_get_info,
                       (const void*)xm_index_table_info, (const void*)xm_index_table_shape, (const void*)xm_index_table_dump, (const void*)xm_index_dup_keys, (const void*)xm_align_batch,
                       (const void*)xm_result_free, (const void*)xm_batch_upload, (const void*)xm_batch_stage, (const void*)xm_batch_commit,
                       (const void*)xm_align_resident, (const void*)xm_seed_probe_packed, (const void*)xm_measure_random_gather, (const void*)xm_test_local_align, (const void*)xm_pileup_new, (const void*)xm_pileup_set_query_ends, (const void*)xm_pileup_read_middle, (const void*)xm_pileup_add_last, (const void*)xm_pileup_read,
                       (const void*)xm_pileup_events, (const void*)xm_pileup_free};
  size_t i;
  for (i = 0; i < sizeof(fns) / sizeof(fns[0]); i++) if (!fns[i]) return 1;
  printf("symbols %d stamp %s devices %d\n", (int)(sizeof(fns) / sizeof(fns[0])), xm_build_stamp(), xm_device_count());
  return 0;
}

static int do_errors(void) {
  xm_index* idx = NULL;
  xm_result* r = NULL;
  double nine[9] = {1.0, 1.5, 0.6, 1.5, 0.6, 0.1, 1.0, 0.1, 0.5};
  int32_t mc[1] = {3};  /* a query has 1 or 2 mates */
  int64_t mo[2] = {0, 0};
  int32_t ml[2] = {4, 0};
  uint8_t codes[4] = {1, 2, 4, 8};
  double z[1] = {0.0}, one[1] = {1.0};
  if (xm_index_build(NULL, NULL, &idx) == 0) { fprintf(stderr, "xm_index_build(NULL) succeeded\n"); return 1; }
  if (!xm_last_error() || !*xm_last_error()) { fprintf(stderr, "no error text after a failed call\n"); return 1; }
  if (xmj_align_batch(NULL, nine, 10, 1, mc, mo, ml, codes, 4, z, one, &r) == 0) { fprintf(stderr, "xm_align_batch(NULL index) succeeded\n"); return 1; }
  if (r != NULL) { fprintf(stderr, "a failing call produced a result\n"); return 1; }
  printf("errors ok: %s\n", xm_last_error());
  return 0;
}

static int do_align(const char* in, const char* outPath) {
  FILE* f = fopen(in, "rb");
  if (!f) { perror(in); return 3; }
  /* reference: n, then per contig: length, codes */
  const int64_t n = rd64(f);
  const uint8_t** codes = (const uint8_t**)calloc((size_t)n, sizeof(*codes));
  int64_t* lengths = (int64_t*)calloc((size_t)n, sizeof(*lengths));
  int64_t c;
  for (c = 0; c < n; c++) { lengths[c] = rd64(f); codes[c] = (const uint8_t*)rd(f, (size_t)lengths[c]); }
  /* batch: nq, codes_length, nine doubles, MaxNumMatches, arrays */
  const int64_t nq = rd64(f), codesLength = rd64(f);
  double* nine = (double*)rd(f, 9 * sizeof(double));
  const int64_t maxNumMatches = rd64(f);
  int32_t* mc = (int32_t*)rd(f, (size_t)nq * 4);
  int64_t* mo = (int64_t*)rd(f, (size_t)nq * 2 * 8);
  int32_t* ml = (int32_t*)rd(f, (size_t)nq * 2 * 4);
  uint8_t* qcodes = (uint8_t*)rd(f, (size_t)codesLength);
  double* ei = (double*)rd(f, (size_t)nq * 8);
  double* dv = (double*)rd(f, (size_t)nq * 8);
  fclose(f);

  xm_index* idx = NULL;
  if (xmj_build_index((int32_t)n, codes, lengths, NULL, 1, 1000, 0, -1, &idx)) { fprintf(stderr, "xm_index_build: %s\n", xm_last_error()); return 1; }
  xm_index_info_t info;
  if (xm_index_get_info(idx, &info)) { fprintf(stderr, "xm_index_get_info: %s\n", xm_last_error()); return 1; }
  xm_result* r = NULL;
  int round;
  for (round = 0; round < 2; round++) {  /* twice: the second call reuses the pooled result buffers the first one gave back */
    if (r) xm_result_free(r);
    r = NULL;
    if (xmj_align_batch(idx, nine, (int32_t)maxNumMatches, nq, mc, mo, ml, qcodes, codesLength, ei, dv, &r)) { fprintf(stderr, "xm_align_batch: %s\n", xm_last_error()); return 1; }
  }
  if (r->num_queries != nq || r->int_off[nq] != r->num_ints || r->dbl_off[nq] != r->num_dbls) { fprintf(stderr, "inconsistent result header\n"); return 1; }
  FILE* o = fopen(outPath, "wb");
  if (!o) { perror(outPath); return 3; }
  fwrite(&r->num_queries, 8, 1, o); fwrite(&r->num_ints, 8, 1, o); fwrite(&r->num_dbls, 8, 1, o);
  fwrite(r->ints, 4, (size_t)r->num_ints, o);
  fwrite(r->dbls, 8, (size_t)r->num_dbls, o);
  fwrite(r->int_off, 8, (size_t)nq + 1, o);
  fwrite(r->dbl_off, 8, (size_t)nq + 1, o);
  fclose(o);
  printf("aligned %lld queries on %d contigs (%lld bases): %lld ints, %lld doubles, kernel %.3f ms in %d launches\n", (long long)nq, info.num_contigs,
         (long long)info.total_forward_size, (long long)r->num_ints, (long long)r->num_dbls, r->kernel_ms, r->kernel_launches);
  xm_result_free(r);
  xm_index_free(idx);
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && !strcmp(argv[1], "symbols")) return do_symbols();
  if (argc >= 2 && !strcmp(argv[1], "errors")) return do_errors();
  if (argc >= 4 && !strcmp(argv[1], "align")) return do_align(argv[2], argv[3]);
  fprintf(stderr, "usage: binding_test symbols | errors | align IN OUT\n");
  return 2;
}

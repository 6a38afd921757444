/* xmapper_jni.c — the JNI shim between mapper.NativeAligner (bindings/java/mapper/NativeAligner.java) and libxmapper_hip.so.
 *
 * The drop-in point in the reference is the per-read call inside the batch loop of AlignerWorker.process()
 * (src/main/java/mapper/AlignerWorker.java:177-231, align(Query) :256-261): NativeAligner.alignBatch replaces that loop, this file
 * carries its arguments across.  No Java object crosses: primitive arrays and one direct ByteBuffer in, four primitive arrays out.
 *
 * Two parts:
 *  1. xmj_* — the marshalling itself on plain C arrays (always compiled; tests/test_c_binding.py builds bindings/c/binding_test.c
 *     against it, so the exact code the JNI functions run is exercised without a JVM);
 *  2. Java_mapper_NativeAligner_* — the JNI entry points (compiled with -DXM_HAVE_JNI on a machine that has a JDK: the image this
 *     repository is developed in has neither jni.h nor javac).
 *
 * Build (maintainer):
 *   gcc -shared -fPIC -DXM_HAVE_JNI -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude bindings/java/xmapper_jni.c \
 *       -Lmapper_amd/_lib -lxmapper_hip -Wl,-rpath,'$ORIGIN' -o libxmapper_jni.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "xmapper_hip.h"

#define XMJ_ABI_VERSION 2  /* include/xmapper_hip.h as of round 6 */

/* ---------------------------------------------------------------- part 1: marshalling on plain C arrays ---------------------------------------- */

/* AlignmentParameters (AlignmentParameters.java:8-33) as NativeAligner passes them: nine doubles in field order + MaxNumMatches. */
void xmj_params(const double* nine, int32_t max_num_matches, xm_params* p) {
  memset(p, 0, sizeof(*p));
  p->MutationPenalty = nine[0];
  p->InsertionStart_Penalty = nine[1];
  p->InsertionExtension_Penalty = nine[2];
  p->DeletionStart_Penalty = nine[3];
  p->DeletionExtension_Penalty = nine[4];
  p->MaxErrorRate = nine[5];
  p->UnalignedPenalty = nine[6];
  p->AmbiguityPenalty = nine[7];
  p->Max_PenaltySpan = nine[8];
  p->MaxNumMatches = max_num_matches;
}

/* new HashBlock_Database(...) + prepare() + new DuplicationDetector(...) as Mapper.run assembles them (Mapper.java:657-692):
 * contigs forward-only, in the order of Mapper.sortAndComplementReference (Mapper.java:1151-1172). */
int xmj_build_index(int32_t num_contigs, const uint8_t* const* codes, const int64_t* lengths, const char* const* names, int32_t enable_gapmers,
                    int32_t duplication_window, int32_t max_query_length, int32_t device, xm_index** out) {
  xm_ref ref;
  xm_build_opts o;
  /* the library must implement the header this shim was compiled against (version 2: xm_result.extra[], xm_seed_probe_packed); -2 = it does not */
  if (xm_abi_version() != XMJ_ABI_VERSION) return -2;
  memset(&ref, 0, sizeof(ref));
  memset(&o, 0, sizeof(o));
  ref.num_contigs = num_contigs;
  ref.names = names;
  ref.codes = codes;
  ref.lengths = lengths;
  o.enable_gapmers = enable_gapmers ? 1 : 0;
  o.min_interesting_size = -1;
  o.max_hashed_length = max_query_length;
  o.dup_window = duplication_window;
  o.dup_min_copies = 2;
  o.dup_min_length = -1;
  o.dup_max_length = -1;
  o.device = device;
  o.host_only = 0;
  return xm_index_build(&ref, &o, out);
}

/* One AlignerWorker.process() batch. */
int xmj_align_batch(xm_index* index, const double* nine, int32_t max_num_matches, int64_t num_queries, const int32_t* mate_count, const int64_t* mate_offset,
                    const int32_t* mate_length, const uint8_t* codes, int64_t codes_length, const double* expected_inner, const double* deviation, xm_result** out) {
  xm_params p;
  xm_query_batch b;
  xmj_params(nine, max_num_matches, &p);
  memset(&b, 0, sizeof(b));
  b.num_queries = num_queries;
  b.mate_count = mate_count;
  b.mate_offset = mate_offset;
  b.mate_length = mate_length;
  b.codes = codes;
  b.codes_length = codes_length;
  b.expected_inner = expected_inner;
  b.deviation = deviation;
  return xm_align_batch(index, &p, &b, out);
}

/* ---------------------------------------------------------------- part 2: the JNI entry points -------------------------------------------------- */
#ifdef XM_HAVE_JNI
#include <jni.h>

static void xmj_throw(JNIEnv* env, const char* prefix) {
  char msg[1024];
  const char* e = xm_last_error();
  jclass rte = (*env)->FindClass(env, "java/lang/RuntimeException");
  size_t n = strlen(prefix);
  if (n > sizeof(msg) - 1) n = sizeof(msg) - 1;
  memcpy(msg, prefix, n);
  strncpy(msg + n, e ? e : "", sizeof(msg) - 1 - n);
  msg[sizeof(msg) - 1] = 0;
  if (rte) (*env)->ThrowNew(env, rte, msg);  /* takes the reference's abort path: AlignerWorker.java:195-197, Mapper.java:1070-1077 */
}

/* private static native long buildIndex(byte[][] contigCodes, String[] names, boolean enableGapmers, int duplicationWindow, int maxQueryLength, int device) */
JNIEXPORT jlong JNICALL Java_mapper_NativeAligner_buildIndex(JNIEnv* env, jclass cls, jobjectArray contigCodes, jobjectArray names, jboolean enableGapmers,
                                                              jint duplicationWindow, jint maxQueryLength, jint device) {
  (void)cls;
  const jsize n = (*env)->GetArrayLength(env, contigCodes);
  const uint8_t** codes = (const uint8_t**)calloc((size_t)n + 1, sizeof(*codes));
  int64_t* lengths = (int64_t*)calloc((size_t)n + 1, sizeof(*lengths));
  const char** cnames = (const char**)calloc((size_t)n + 1, sizeof(*cnames));
  jbyteArray* arrays = (jbyteArray*)calloc((size_t)n + 1, sizeof(*arrays));
  jstring* strings = (jstring*)calloc((size_t)n + 1, sizeof(*strings));
  xm_index* idx = NULL;
  int rc = -1;
  jsize i, pinned = 0;
  if (!codes || !lengths || !cnames || !arrays || !strings) goto done;
  for (i = 0; i < n; i++) {
    arrays[i] = (jbyteArray)(*env)->GetObjectArrayElement(env, contigCodes, i);
    lengths[i] = (int64_t)(*env)->GetArrayLength(env, arrays[i]);
    codes[i] = (const uint8_t*)(*env)->GetByteArrayElements(env, arrays[i], NULL);  /* (a copy or a pin: released below, JNI_ABORT = nothing to write back) */
    if (!codes[i]) goto done;
    pinned = i + 1;
    if (names) {
      strings[i] = (jstring)(*env)->GetObjectArrayElement(env, names, i);
      cnames[i] = strings[i] ? (*env)->GetStringUTFChars(env, strings[i], NULL) : NULL;
    }
  }
  rc = xmj_build_index((int32_t)n, codes, lengths, names ? cnames : NULL, enableGapmers ? 1 : 0, (int32_t)duplicationWindow, (int32_t)maxQueryLength, (int32_t)device, &idx);
done:
  for (i = 0; i < pinned; i++) {
    (*env)->ReleaseByteArrayElements(env, arrays[i], (jbyte*)codes[i], JNI_ABORT);
    if (cnames && cnames[i]) (*env)->ReleaseStringUTFChars(env, strings[i], cnames[i]);
  }
  free(codes); free(lengths); free(cnames); free(arrays); free(strings);
  if (rc != 0) { xmj_throw(env, "Failed to build the reference index: "); return 0; }
  return (jlong)(intptr_t)idx;
}

/* private static native long newContext(long handle): a further context of the index for another AlignerWorker thread (xm_context_new: own stream,
 * batch buffers and scratch over the same tables, as the workers share one HashBlock_Database through views, HashBlock_Database.java:129-133) */
JNIEXPORT jlong JNICALL Java_mapper_NativeAligner_newContext(JNIEnv* env, jclass cls, jlong handle) {
  xm_index* ctx = NULL;
  (void)cls;
  if (xm_context_new((xm_index*)(intptr_t)handle, &ctx) != 0) { xmj_throw(env, "Failed to make a context of the reference index: "); return 0; }
  return (jlong)(intptr_t)ctx;
}

/* private static native void freeIndex(long handle) */
JNIEXPORT void JNICALL Java_mapper_NativeAligner_freeIndex(JNIEnv* env, jclass cls, jlong handle) {
  (void)env; (void)cls;
  xm_index_free((xm_index*)(intptr_t)handle);
}

/* private static native boolean alignBatch(long handle, double[] parameters9, int maxNumMatches, int[] mateCount, long[] mateOffset, int[] mateLength,
 *                                          ByteBuffer codes, double[] expectedInnerDistance, double[] spacingDeviationPerUnitPenalty, ResultStreams out)
 * Fills out.ints / out.dbls / out.intOff / out.dblOff (List<QueryAlignments> flattened as include/xmapper_hip.h documents). */
JNIEXPORT jboolean JNICALL Java_mapper_NativeAligner_alignBatch(JNIEnv* env, jclass cls, jlong handle, jdoubleArray parameters9, jint maxNumMatches, jintArray mateCount,
                                                                 jlongArray mateOffset, jintArray mateLength, jobject codes, jdoubleArray expectedInner, jdoubleArray deviation,
                                                                 jobject out) {
  (void)cls;
  const jsize nq = (*env)->GetArrayLength(env, mateCount);
  if ((*env)->GetArrayLength(env, parameters9) != 9 || (*env)->GetArrayLength(env, mateOffset) != 2 * nq || (*env)->GetArrayLength(env, mateLength) != 2 * nq ||
      (*env)->GetArrayLength(env, expectedInner) != nq || (*env)->GetArrayLength(env, deviation) != nq) {
    jclass iae = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (iae) (*env)->ThrowNew(env, iae, "alignBatch: array lengths do not describe one batch (9 parameters; mateOffset and mateLength 2 per query; one inner distance and deviation per query)");
    return JNI_FALSE;
  }
  const uint8_t* codeBase = (const uint8_t*)(*env)->GetDirectBufferAddress(env, codes);
  const jlong codeBytes = (*env)->GetDirectBufferCapacity(env, codes);
  if (!codeBase || codeBytes < 0) {
    jclass iae = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
    if (iae) (*env)->ThrowNew(env, iae, "alignBatch: codes must be a direct ByteBuffer");
    return JNI_FALSE;
  }
  jdouble* p9 = (*env)->GetDoubleArrayElements(env, parameters9, NULL);
  jint* mc = (*env)->GetIntArrayElements(env, mateCount, NULL);
  jlong* mo = (*env)->GetLongArrayElements(env, mateOffset, NULL);
  jint* ml = (*env)->GetIntArrayElements(env, mateLength, NULL);
  jdouble* ei = (*env)->GetDoubleArrayElements(env, expectedInner, NULL);
  jdouble* dv = (*env)->GetDoubleArrayElements(env, deviation, NULL);
  xm_result* r = NULL;
  int rc = -1;
  if (p9 && mc && mo && ml && ei && dv)
    rc = xmj_align_batch((xm_index*)(intptr_t)handle, (const double*)p9, (int32_t)maxNumMatches, (int64_t)nq, (const int32_t*)mc, (const int64_t*)mo, (const int32_t*)ml, codeBase,
                         (int64_t)codeBytes, (const double*)ei, (const double*)dv, &r);
  if (p9) (*env)->ReleaseDoubleArrayElements(env, parameters9, p9, JNI_ABORT);
  if (mc) (*env)->ReleaseIntArrayElements(env, mateCount, mc, JNI_ABORT);
  if (mo) (*env)->ReleaseLongArrayElements(env, mateOffset, mo, JNI_ABORT);
  if (ml) (*env)->ReleaseIntArrayElements(env, mateLength, ml, JNI_ABORT);
  if (ei) (*env)->ReleaseDoubleArrayElements(env, expectedInner, ei, JNI_ABORT);
  if (dv) (*env)->ReleaseDoubleArrayElements(env, deviation, dv, JNI_ABORT);
  if (rc != 0) { xmj_throw(env, "Failed to align: "); return JNI_FALSE; }

  /* the four streams -> four Java arrays (one copy each; the library's pinned buffers go back to its pool) */
  jboolean ok = JNI_FALSE;
  if (r->num_ints > 0x7FFFFFF0ll || r->num_dbls > 0x7FFFFFF0ll) {
    jclass ise = (*env)->FindClass(env, "java/lang/IllegalStateException");
    if (ise) (*env)->ThrowNew(env, ise, "alignBatch: the result of this batch does not fit Java arrays; use smaller batches");
  } else {
    jintArray ints = (*env)->NewIntArray(env, (jsize)r->num_ints);
    jdoubleArray dbls = (*env)->NewDoubleArray(env, (jsize)r->num_dbls);
    jlongArray intOff = (*env)->NewLongArray(env, (jsize)(r->num_queries + 1));
    jlongArray dblOff = (*env)->NewLongArray(env, (jsize)(r->num_queries + 1));
    if (ints && dbls && intOff && dblOff) {  /* (else: OutOfMemoryError is pending) */
      (*env)->SetIntArrayRegion(env, ints, 0, (jsize)r->num_ints, (const jint*)r->ints);
      (*env)->SetDoubleArrayRegion(env, dbls, 0, (jsize)r->num_dbls, (const jdouble*)r->dbls);
      (*env)->SetLongArrayRegion(env, intOff, 0, (jsize)(r->num_queries + 1), (const jlong*)r->int_off);
      (*env)->SetLongArrayRegion(env, dblOff, 0, (jsize)(r->num_queries + 1), (const jlong*)r->dbl_off);
      jclass rc2 = (*env)->GetObjectClass(env, out);
      jfieldID fInts = (*env)->GetFieldID(env, rc2, "ints", "[I"), fDbls = (*env)->GetFieldID(env, rc2, "dbls", "[D");
      jfieldID fIntOff = (*env)->GetFieldID(env, rc2, "intOff", "[J"), fDblOff = (*env)->GetFieldID(env, rc2, "dblOff", "[J");
      jfieldID fKernelMs = (*env)->GetFieldID(env, rc2, "kernelMillis", "D");
      if (fInts && fDbls && fIntOff && fDblOff && fKernelMs) {
        (*env)->SetObjectField(env, out, fInts, ints);
        (*env)->SetObjectField(env, out, fDbls, dbls);
        (*env)->SetObjectField(env, out, fIntOff, intOff);
        (*env)->SetObjectField(env, out, fDblOff, dblOff);
        (*env)->SetDoubleField(env, out, fKernelMs, (jdouble)r->kernel_ms);
        ok = JNI_TRUE;
      }
    }
  }
  xm_result_free(r);
  return ok;
}
#endif /* XM_HAVE_JNI */

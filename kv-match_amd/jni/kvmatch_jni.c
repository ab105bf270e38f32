/*
 * kvmatch_jni.c -- JNI shim over include/kvmatch.h for cn.edu.fudan.dsm.kvmatch.gpu.Phase2Gpu.
 * Not part of the default build: this image has no JDK.  Compiles only where <jni.h> is available
 * (cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux kvmatch_jni.c -L../csrc -lkvmatch_hip).
 * tests/test_abi.py type-checks it against a minimal declaration-only jni.h (tests/jni_stub/) and compares every
 * export's C signature with the `native` declaration in Phase2Gpu.java.
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#define KVM_HAVE_JNI 1
#endif
#endif

#ifdef KVM_HAVE_JNI
#include <jni.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/kvmatch.h"

static void throw_for(JNIEnv *env, kvm_ctx *ctx, int rc) {
    const char *msg = kvm_last_error(ctx);
    /* range/parameter errors -> IllegalArgumentException (K/operator/file/TimeSeriesFileOperator.java:55-57); where the
     * reference's DtwUtils.lowerUpperLemire indexes below its arrays (K/utils/DtwUtils.java:81-90; KVM_OPT_STRICT_BOUNDS) ->
     * ArrayIndexOutOfBoundsException with the index as its message, as JDK 8 words it; a host allocation that failed inside
     * the library (KVM_ERR_NO_MEMORY: no C++ exception crosses the C ABI) -> OutOfMemoryError, what the JVM's own `new`
     * would have thrown; everything else -> IOException (what query() declares, K/QueryEngine.java:155) */
    jclass cls = (*env)->FindClass(env, rc == KVM_ERR_ILLEGAL_ARGUMENT      ? "java/lang/IllegalArgumentException"
                                        : rc == KVM_ERR_INDEX_OUT_OF_BOUNDS ? "java/lang/ArrayIndexOutOfBoundsException"
                                        : rc == KVM_ERR_NO_MEMORY           ? "java/lang/OutOfMemoryError"
                                                                            : "java/io/IOException");
    (*env)->ThrowNew(env, cls, msg ? msg : "kvmatch error");
}

static void throw_new(JNIEnv *env, const char *cls_name, const char *msg) {
    (*env)->ThrowNew(env, (*env)->FindClass(env, cls_name), msg);
}

JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_open(JNIEnv *env, jclass c, jint device) {
    (void)c;
    kvm_ctx *ctx = 0;
    int rc = kvm_open(device, &ctx);
    if (rc) { throw_for(env, 0, rc); return 0; }
    return (jlong)(intptr_t)ctx;
}

JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_loadSeriesFile(JNIEnv *env, jclass c, jlong h,
                                                                               jstring path, jlong n) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    const char *p = (*env)->GetStringUTFChars(env, path, 0);
    int rc = kvm_load_series_file(ctx, p, n);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_for(env, ctx, rc);
}

JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_close(JNIEnv *env, jclass c, jlong h) {
    (void)env; (void)c;
    kvm_close((kvm_ctx *)(intptr_t)h);
}

static jobject make_result(JNIEnv *env, kvm_ctx *ctx, int rc, kvm_result *res) {
    if (rc) { throw_for(env, ctx, rc); return 0; }
    jclass cls = (*env)->FindClass(env, "cn/edu/fudan/dsm/kvmatch/gpu/Phase2Gpu$Result");
    jobject o = cls ? (*env)->AllocObject(env, cls) : 0;
    jsize cnt = (jsize)kvm_result_count(res);
    jintArray off = o ? (*env)->NewIntArray(env, cnt) : 0;
    jdoubleArray dist = off ? (*env)->NewDoubleArray(env, cnt) : 0;
    if (!dist) {  /* (a Java exception -- NoClassDefFoundError / OutOfMemoryError -- is pending) */
        kvm_result_free(res);
        return 0;
    }
    if (cnt) {
        (*env)->SetIntArrayRegion(env, off, 0, cnt, (const jint *)kvm_result_offsets(res));
        (*env)->SetDoubleArrayRegion(env, dist, 0, cnt, kvm_result_distances(res));
    }
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, cls, "offsets", "[I"), off);
    (*env)->SetObjectField(env, o, (*env)->GetFieldID(env, cls, "distances", "[D"), dist);
    (*env)->SetLongField(env, o, (*env)->GetFieldID(env, cls, "cntCandidate", "J"), kvm_result_candidates(res));
    kvm_result_free(res);
    return o;
}

/* Inputs are COPIED out of the Java heap (Get<Type>ArrayRegion) before the library is called: no array is pinned and no
 * JNI critical region is open while the GPU works (a phase-2 call takes 0.2 ms .. 0.4 s; GetPrimitiveArrayCritical
 * would stall the collector for all of it).  The copies are m*8 + 2k*4 bytes -- what the library itself moves into its
 * pinned staging buffer in the first microseconds of the call. */
typedef struct {
    double *q;
    jint *left, *right;
    jsize m, k;
} kvm_jni_in;

static void in_free(kvm_jni_in *in) {
    free(in->q);
    free(in->left);
    free(in->right);
}

/* returns 0 and leaves a pending Java exception on failure */
static int in_copy(JNIEnv *env, jdoubleArray q, jintArray left, jintArray right, kvm_jni_in *in) {
    memset(in, 0, sizeof *in);
    if (!q || !left || !right) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "null array");
        return 0;
    }
    in->m = (*env)->GetArrayLength(env, q);
    in->k = (*env)->GetArrayLength(env, left);
    if ((*env)->GetArrayLength(env, right) != in->k) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "left/right length mismatch");
        return 0;
    }
    in->q = (double *)malloc((size_t)(in->m > 0 ? in->m : 1) * sizeof(double));
    in->left = (jint *)malloc((size_t)(in->k > 0 ? in->k : 1) * sizeof(jint));
    in->right = (jint *)malloc((size_t)(in->k > 0 ? in->k : 1) * sizeof(jint));
    if (!in->q || !in->left || !in->right) {
        in_free(in);
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "kvmatch_jni: staging copy");
        return 0;
    }
    (*env)->GetDoubleArrayRegion(env, q, 0, in->m, in->q);
    (*env)->GetIntArrayRegion(env, left, 0, in->k, in->left);
    (*env)->GetIntArrayRegion(env, right, 0, in->k, in->right);
    if ((*env)->ExceptionCheck(env)) {
        in_free(in);
        return 0;
    }
    return 1;
}

#define KVM_JNI_BODY(CALL)                                                                         \
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;                                                         \
    kvm_jni_in in;                                                                                 \
    if (!in_copy(env, q, left, right, &in)) return 0;                                              \
    const double *pq = in.q;                                                                       \
    const jint *pl = in.left, *pr = in.right;                                                      \
    const jsize m = in.m, k = in.k;                                                                \
    kvm_result *res = 0;                                                                           \
    int rc = CALL;                                                                                 \
    in_free(&in);                                                                                  \
    return make_result(env, ctx, rc, res);

JNIEXPORT jobject JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_rsmEd(JNIEnv *env, jobject self, jlong h,
        jdoubleArray q, jintArray left, jintArray right, jint shift, jdouble eps) {
    (void)self;
    KVM_JNI_BODY(kvm_verify_rsm_ed(ctx, pq, m, (const int32_t *)pl, (const int32_t *)pr, k, shift, eps, &res))
}
JNIEXPORT jobject JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_cnsmEd(JNIEnv *env, jobject self, jlong h,
        jdoubleArray q, jintArray left, jintArray right, jint shift, jdouble eps, jdouble alpha, jdouble beta) {
    (void)self;
    KVM_JNI_BODY(kvm_verify_cnsm_ed(ctx, pq, m, (const int32_t *)pl, (const int32_t *)pr, k, shift, eps, alpha, beta, &res))
}
JNIEXPORT jobject JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_rsmDtw(JNIEnv *env, jobject self, jlong h,
        jdoubleArray q, jintArray left, jintArray right, jint shift, jdouble eps, jint rho) {
    (void)self;
    KVM_JNI_BODY(kvm_verify_rsm_dtw(ctx, pq, m, (const int32_t *)pl, (const int32_t *)pr, k, shift, eps, rho, &res))
}
JNIEXPORT jobject JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_cnsmDtw(JNIEnv *env, jobject self, jlong h,
        jdoubleArray q, jintArray left, jintArray right, jint shift, jdouble eps, jint rho, jdouble alpha,
        jdouble beta) {
    (void)self;
    KVM_JNI_BODY(kvm_verify_cnsm_dtw(ctx, pq, m, (const int32_t *)pl, (const int32_t *)pr, k, shift, eps, rho, alpha, beta, &res))
}

/* TimeSeriesOperator.readTimeSeries(left, length) from HBM (K/operator/TimeSeriesOperator.java:38) */
JNIEXPORT jdoubleArray JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_readTimeSeries(JNIEnv *env, jclass c, jlong h,
                                                                                       jint left, jint length) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    const jsize cnt = length > 0 ? length : 0;
    double *buf = (double *)malloc((size_t)(cnt > 0 ? cnt : 1) * sizeof(double));
    if (!buf) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/OutOfMemoryError"), "kvmatch_jni: readTimeSeries");
        return 0;
    }
    int rc = kvm_read_time_series(ctx, left, length, buf);  /* (range errors as the file operator raises them) */
    jdoubleArray out = 0;
    if (rc) {
        throw_for(env, ctx, rc);
    } else if ((out = (*env)->NewDoubleArray(env, cnt)) != 0) {
        (*env)->SetDoubleArrayRegion(env, out, 0, cnt, buf);
    }
    free(buf);
    return out;
}

/* phase-2 time model for the early-termination test (K/QueryEngine.java:316-327) */
JNIEXPORT jdouble JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_phase2TimeEstimateMs(JNIEnv *env, jclass c, jint variant,
        jlong windows, jlong offsets, jint length) {
    (void)env; (void)c;
    return kvm_phase2_time_estimate_ms(variant, windows, offsets, length);
}

/* the early-termination test itself (K/QueryEngine.java:321-326); last[0] carries lastTotalTimeUsageEstimated */
JNIEXPORT jboolean JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_phase1ShouldStop(JNIEnv *env, jclass c, jint variant,
        jint segmentIndex, jdouble phase1Ms, jlong windows, jlong offsets, jint length, jdoubleArray last) {
    (void)c;
    if (!last || (*env)->GetArrayLength(env, last) < 1) {
        throw_new(env, "java/lang/IllegalArgumentException", "phase1ShouldStop: last must hold one double");
        return 0;
    }
    jdouble v = 0;
    (*env)->GetDoubleArrayRegion(env, last, 0, 1, &v);
    const int32_t rc = kvm_phase1_should_stop(variant, segmentIndex, phase1Ms, windows, offsets, length, &v);
    if (rc < 0) {
        throw_new(env, "java/lang/IllegalArgumentException", "phase1ShouldStop: unknown variant");
        return 0;
    }
    (*env)->SetDoubleArrayRegion(env, last, 0, 1, &v);
    return (jboolean)(rc ? 1 : 0);
}

/* Index build step 1 (K/IndexBuilder.java:186-300): build, bulk accessors, free */
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsBuild(JNIEnv *env, jclass c, jlong h, jint w) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    kvm_index_rows *rows = 0;
    int rc = kvm_index_rows_build(ctx, w, &rows);
    if (rc) { throw_for(env, ctx, rc); return 0; }
    return (jlong)(intptr_t)rows;
}
/* grouped == true: intervals ordered by (key descending, position), rows via indexRowsGroups / indexRowsGetGroups */
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsBuildGrouped(JNIEnv *env, jclass c, jlong h, jint w) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    kvm_index_rows *rows = 0;
    int rc = kvm_index_rows_build_grouped(ctx, w, &rows);
    if (rc) { throw_for(env, ctx, rc); return 0; }
    return (jlong)(intptr_t)rows;
}
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsGroups(JNIEnv *env, jclass c, jlong rows) {
    (void)env; (void)c;
    return kvm_index_rows_groups((const kvm_index_rows *)(intptr_t)rows);
}
/* keys[g], offsets[g] for all groups (offsets has groups + 1 entries) */
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsGetGroups(JNIEnv *env, jclass c, jlong rows,
        jdoubleArray keys, jintArray offsets) {
    (void)c;
    const kvm_index_rows *r = (const kvm_index_rows *)(intptr_t)rows;
    jsize ng = (jsize)kvm_index_rows_groups(r);
    if ((*env)->GetArrayLength(env, keys) != ng || (*env)->GetArrayLength(env, offsets) != ng + 1) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "index rows groups");
        return;
    }
    (*env)->SetDoubleArrayRegion(env, keys, 0, ng, kvm_index_rows_group_keys(r));
    (*env)->SetIntArrayRegion(env, offsets, 0, ng + 1, (const jint *)kvm_index_rows_group_offsets(r));
}
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsCount(JNIEnv *env, jclass c, jlong rows) {
    (void)env; (void)c;
    return kvm_index_rows_count((const kvm_index_rows *)(intptr_t)rows);
}
/* copies intervals [from, from + keys.length) into the caller's arrays */
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsGet(JNIEnv *env, jclass c, jlong rows, jlong from,
        jdoubleArray keys, jintArray first, jintArray last) {
    (void)c;
    const kvm_index_rows *r = (const kvm_index_rows *)(intptr_t)rows;
    jsize cnt = (*env)->GetArrayLength(env, keys);
    if (from < 0 || from + cnt > kvm_index_rows_count(r)) {
        (*env)->ThrowNew(env, (*env)->FindClass(env, "java/lang/IllegalArgumentException"), "index rows range");
        return;
    }
    (*env)->SetDoubleArrayRegion(env, keys, 0, cnt, kvm_index_rows_keys(r) + from);
    (*env)->SetIntArrayRegion(env, first, 0, cnt, (const jint *)(kvm_index_rows_first(r) + from));
    (*env)->SetIntArrayRegion(env, last, 0, cnt, (const jint *)(kvm_index_rows_last(r) + from));
}
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexRowsFree(JNIEnv *env, jclass c, jlong rows) {
    (void)env; (void)c;
    kvm_index_rows_free((kvm_index_rows *)(intptr_t)rows);
}
/* IndexBuilder.SingleIndexBuilder.run + IndexFileOperator.writeAll for one window length (K/IndexBuilder.java:186-347,
 * K/operator/file/IndexFileOperator.java:127-165): the bytes of files/index-{N}-{Wu} */
JNIEXPORT jbyteArray JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_indexBuild(JNIEnv *env, jclass c, jlong h, jint w) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    kvm_index_file *f = 0;
    int rc = kvm_index_build(ctx, w, &f);
    if (rc) { throw_for(env, ctx, rc); return 0; }
    const jsize size = (jsize)kvm_index_file_size(f);      /* < 2^31: kvm_index_build refuses longer images */
    jbyteArray out = (*env)->NewByteArray(env, size);
    if (out) (*env)->SetByteArrayRegion(env, out, 0, size, (const jbyte *)kvm_index_file_bytes(f));
    kvm_index_file_free(f);
    return out;
}
/* kvm_set_option, e.g. KVM_OPT_SORT_VIOLATION (6) */
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_setOption(JNIEnv *env, jclass c, jlong h, jint option, jlong value) {
    (void)c;
    kvm_ctx *ctx = (kvm_ctx *)(intptr_t)h;
    int rc = kvm_set_option(ctx, option, value);
    if (rc) throw_for(env, ctx, rc);
}

/* ---- one engine over the GPUs of a node (kvm_group): one call per query ---------------------------------------------- */
static void throw_for_group(JNIEnv *env, kvm_group *g, int rc) {
    const char *msg = kvm_group_last_error(g);
    /* as throw_for; KVM_ERR_SHARD_FAILED (another rank failed on its shard) is an IOException like every device failure */
    jclass cls = (*env)->FindClass(env, rc == KVM_ERR_ILLEGAL_ARGUMENT      ? "java/lang/IllegalArgumentException"
                                        : rc == KVM_ERR_INDEX_OUT_OF_BOUNDS ? "java/lang/ArrayIndexOutOfBoundsException"
                                        : rc == KVM_ERR_NO_MEMORY           ? "java/lang/OutOfMemoryError"
                                                                            : "java/io/IOException");
    (*env)->ThrowNew(env, cls, msg ? msg : "kvmatch group error");
}
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupOpen(JNIEnv *env, jclass c, jintArray devices, jint inFlight) {
    (void)c;
    if (!devices) { throw_new(env, "java/lang/IllegalArgumentException", "groupOpen: null device list"); return 0; }
    const jsize n = (*env)->GetArrayLength(env, devices);
    jint devs[64];
    if (n < 1 || n > 64) { throw_new(env, "java/lang/IllegalArgumentException", "groupOpen: 1 .. 64 devices"); return 0; }
    (*env)->GetIntArrayRegion(env, devices, 0, n, devs);
    kvm_group *g = 0;
    int rc = kvm_open_devices((const int32_t *)devs, n, inFlight, 0, &g);
    if (rc) { throw_for_group(env, 0, rc); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT jbyteArray JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupUniqueId(JNIEnv *env, jclass c) {
    (void)c;
    jbyte id[KVM_GROUP_ID_BYTES];
    int rc = kvm_group_unique_id(id, KVM_GROUP_ID_BYTES);
    if (rc) { throw_for_group(env, 0, rc); return 0; }
    jbyteArray out = (*env)->NewByteArray(env, KVM_GROUP_ID_BYTES);
    if (out) (*env)->SetByteArrayRegion(env, out, 0, KVM_GROUP_ID_BYTES, id);
    return out;
}
JNIEXPORT jlong JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupOpenRank(JNIEnv *env, jclass c, jint device, jint rank, jint world,
        jbyteArray id, jint inFlight) {
    (void)c;
    jbyte buf[KVM_GROUP_ID_BYTES];
    const jsize n = id ? (*env)->GetArrayLength(env, id) : 0;
    if (n == KVM_GROUP_ID_BYTES) (*env)->GetByteArrayRegion(env, id, 0, n, buf);
    kvm_group *g = 0;
    int rc = kvm_open_rank(device, rank, world, n == KVM_GROUP_ID_BYTES ? buf : 0, n, inFlight, 0, &g);
    if (rc) { throw_for_group(env, 0, rc); return 0; }
    return (jlong)(intptr_t)g;
}
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupLoadSeriesFile(JNIEnv *env, jclass c, jlong h, jstring path, jlong n) {
    (void)c;
    kvm_group *g = (kvm_group *)(intptr_t)h;
    const char *p = (*env)->GetStringUTFChars(env, path, 0);
    int rc = kvm_group_load_series_file(g, p, n);
    (*env)->ReleaseStringUTFChars(env, path, p);
    if (rc) throw_for_group(env, g, rc);
}
/* the phase-2 pass of one query on every GPU of the group: ONE call from the engine (K/QueryEngine.java:338-363 and siblings) */
JNIEXPORT jobject JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupVerify(JNIEnv *env, jclass c, jlong h, jint variant,
        jdoubleArray q, jintArray left, jintArray right, jint shift, jdouble eps, jint rho, jdouble alpha, jdouble beta) {
    (void)c;
    kvm_group *g = (kvm_group *)(intptr_t)h;
    kvm_jni_in in;
    if (!in_copy(env, q, left, right, &in)) return 0;
    kvm_result *res = 0;
    int rc = kvm_verify_sharded(g, variant, in.q, in.m, (const int32_t *)in.left, (const int32_t *)in.right, in.k, shift, eps, rho, alpha,
                                beta, &res);
    in_free(&in);
    if (rc) { throw_for_group(env, g, rc); return 0; }
    return make_result(env, 0, KVM_OK, res);
}
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupSetOption(JNIEnv *env, jclass c, jlong h, jint option, jlong value) {
    (void)c;
    kvm_group *g = (kvm_group *)(intptr_t)h;
    int rc = kvm_group_set_option(g, option, value);
    if (rc) throw_for_group(env, g, rc);
}
JNIEXPORT void JNICALL Java_cn_edu_fudan_dsm_kvmatch_gpu_Phase2Gpu_groupClose(JNIEnv *env, jclass c, jlong h) {
    (void)env; (void)c;
    kvm_group_close((kvm_group *)(intptr_t)h);
}
#else
/* no <jni.h> in this toolchain: nothing to build */
typedef int kvm_jni_not_built;
#endif

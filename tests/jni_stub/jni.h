/* Declaration-only stand-in for <jni.h>, for TYPE-CHECKING kv-match_amd/jni/kvmatch_jni.c in an image without a JDK
 * (gcc -fsyntax-only; tests/test_abi.py).  It declares the JNI types and exactly the JNIEnv entry points the shim uses,
 * with the signatures of the JNI specification (Java SE 8, "JNI Functions"); nothing here is ever linked or run, and a
 * real build uses the JDK's own header.  TEST INFRASTRUCTURE ONLY. */
#ifndef KVM_TEST_JNI_STUB_H
#define KVM_TEST_JNI_STUB_H
#include <stdint.h>

typedef int32_t jint;
typedef int64_t jlong;
typedef double jdouble;
typedef signed char jbyte;
typedef uint8_t jboolean;
typedef jint jsize;

struct _jobject;
typedef struct _jobject *jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jthrowable;
typedef jobject jarray;
typedef jarray jintArray;
typedef jarray jdoubleArray;
typedef jarray jbyteArray;
struct _jfieldID;
typedef struct _jfieldID *jfieldID;

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2

struct JNINativeInterface_;
typedef const struct JNINativeInterface_ *JNIEnv;

struct JNINativeInterface_ {
    jclass (JNICALL *FindClass)(JNIEnv *env, const char *name);
    jint (JNICALL *ThrowNew)(JNIEnv *env, jclass clazz, const char *msg);
    jboolean (JNICALL *ExceptionCheck)(JNIEnv *env);
    jobject (JNICALL *AllocObject)(JNIEnv *env, jclass clazz);
    jfieldID (JNICALL *GetFieldID)(JNIEnv *env, jclass clazz, const char *name, const char *sig);
    void (JNICALL *SetObjectField)(JNIEnv *env, jobject obj, jfieldID fieldID, jobject val);
    void (JNICALL *SetLongField)(JNIEnv *env, jobject obj, jfieldID fieldID, jlong val);
    const char *(JNICALL *GetStringUTFChars)(JNIEnv *env, jstring str, jboolean *isCopy);
    void (JNICALL *ReleaseStringUTFChars)(JNIEnv *env, jstring str, const char *chars);
    jsize (JNICALL *GetArrayLength)(JNIEnv *env, jarray array);
    jintArray (JNICALL *NewIntArray)(JNIEnv *env, jsize len);
    jdoubleArray (JNICALL *NewDoubleArray)(JNIEnv *env, jsize len);
    jbyteArray (JNICALL *NewByteArray)(JNIEnv *env, jsize len);
    void (JNICALL *GetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize l, jint *buf);
    void (JNICALL *GetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize l, jdouble *buf);
    void (JNICALL *GetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize l, jbyte *buf);
    void (JNICALL *SetIntArrayRegion)(JNIEnv *env, jintArray array, jsize start, jsize l, const jint *buf);
    void (JNICALL *SetDoubleArrayRegion)(JNIEnv *env, jdoubleArray array, jsize start, jsize l, const jdouble *buf);
    void (JNICALL *SetByteArrayRegion)(JNIEnv *env, jbyteArray array, jsize start, jsize l, const jbyte *buf);
};
#endif

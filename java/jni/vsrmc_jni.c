/* SOURCE ONLY — compiled only where <jni.h> exists (not in the build image).
 * JNI glue between java/tlc2/tool/fp/GpuFPSet.java and the C ABI of include/vsrmc.h.
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include vsrmc_jni.c -L../../vsr-tlaplus_amd -lvsrmc -o libvsrmc_jni.so
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#include <jni.h>
#include <stdint.h>

#include "vsrmc.h"

JNIEXPORT jlong JNICALL Java_tlc2_tool_fp_GpuFPSet_create0(JNIEnv* env, jclass cls, jint device, jint log2_slots) {
  (void)env; (void)cls;
  vsrmc_fpset* h = NULL;
  if (vsrmc_fpset_create(device, log2_slots, &h) != 0) return 0;
  return (jlong)(intptr_t)h;
}

static jint block(JNIEnv* env, jlong handle, jlongArray fps, jbyteArray out, int put) {
  jsize n = (*env)->GetArrayLength(env, fps);
  jlong* f = (*env)->GetLongArrayElements(env, fps, NULL);
  jbyte* o = (*env)->GetByteArrayElements(env, out, NULL);
  int32_t rc = put ? vsrmc_fpset_put_batch((vsrmc_fpset*)(intptr_t)handle, (const uint64_t*)f, (uint64_t)n, (uint8_t*)o)
                   : vsrmc_fpset_contains_batch((vsrmc_fpset*)(intptr_t)handle, (const uint64_t*)f, (uint64_t)n, (uint8_t*)o);
  (*env)->ReleaseLongArrayElements(env, fps, f, JNI_ABORT);
  (*env)->ReleaseByteArrayElements(env, out, o, 0);
  return rc;
}
JNIEXPORT jint JNICALL Java_tlc2_tool_fp_GpuFPSet_putBlock0(JNIEnv* env, jclass cls, jlong h, jlongArray fps, jbyteArray out) {
  (void)cls;
  return block(env, h, fps, out, 1);
}
JNIEXPORT jint JNICALL Java_tlc2_tool_fp_GpuFPSet_containsBlock0(JNIEnv* env, jclass cls, jlong h, jlongArray fps, jbyteArray out) {
  (void)cls;
  return block(env, h, fps, out, 0);
}
JNIEXPORT jlong JNICALL Java_tlc2_tool_fp_GpuFPSet_size0(JNIEnv* env, jclass cls, jlong h) {
  (void)env; (void)cls;
  uint64_t n = 0;
  vsrmc_fpset_size((vsrmc_fpset*)(intptr_t)h, &n);
  return (jlong)n;
}
JNIEXPORT void JNICALL Java_tlc2_tool_fp_GpuFPSet_destroy0(JNIEnv* env, jclass cls, jlong h) {
  (void)env; (void)cls;
  vsrmc_fpset_destroy((vsrmc_fpset*)(intptr_t)h);
}
JNIEXPORT jstring JNICALL Java_tlc2_tool_fp_GpuFPSet_lastError0(JNIEnv* env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, vsrmc_last_error());
}
#endif
#endif

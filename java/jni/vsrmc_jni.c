/* SOURCE ONLY — compiled only where <jni.h> exists (not in the build image).
 * JNI glue between java/tlc2/tool/fp/GpuFPSet.java, java/tlc2/tool/GpuModelChecker.java and the C ABI of include/vsrmc.h.
 *   cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include vsrmc_jni.c -L../../vsr-tlaplus_amd -lvsrmc -o libvsrmc_jni.so
 */
#if defined(__has_include)
#if __has_include(<jni.h>)
#include <jni.h>
#include <stdint.h>
#include <stdio.h>

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

/* ---- java/tlc2/tool/GpuModelChecker.java: the whole Worker / StateQueue / FPSet / TLCTrace loop behind six natives ---------- */
#include <stdlib.h>
#include <string.h>

JNIEXPORT jlong JNICALL Java_tlc2_tool_GpuModelChecker_modelLoad(JNIEnv* env, jclass cls, jstring tla, jstring cfg) {
  (void)cls;
  const char* t = tla ? (*env)->GetStringUTFChars(env, tla, NULL) : NULL;
  const char* c = (*env)->GetStringUTFChars(env, cfg, NULL);
  vsrmc_model* m = NULL;
  int32_t rc = vsrmc_model_load(t, c, &m);
  if (t) (*env)->ReleaseStringUTFChars(env, tla, t);
  (*env)->ReleaseStringUTFChars(env, cfg, c);
  return rc == 0 ? (jlong)(intptr_t)m : 0;
}

JNIEXPORT jlong JNICALL Java_tlc2_tool_GpuModelChecker_checkerCreate(JNIEnv* env, jclass cls, jlong model, jint device, jint table_log2,
                                                                     jlong frontier_words, jlong frontier_states, jlong pending_entries,
                                                                     jlong trace_entries) {
  (void)env; (void)cls;
  vsrmc_options o;
  vsrmc_options_default(&o);
  o.device = device;
  o.table_log2 = table_log2;
  o.frontier_words = (uint64_t)frontier_words;
  o.frontier_states = (uint64_t)frontier_states;
  o.pending_entries = (uint64_t)pending_entries;
  o.trace_entries = (uint64_t)trace_entries;
  vsrmc_checker* c = NULL;
  if (vsrmc_checker_create((const vsrmc_model*)(intptr_t)model, &o, &c) != 0) return 0;
  return (jlong)(intptr_t)c;
}

/* out = {level, nNew, distinct, totalGenerated, violMask, violIndex, deadlocks} */
JNIEXPORT jint JNICALL Java_tlc2_tool_GpuModelChecker_checkerStep(JNIEnv* env, jclass cls, jlong checker, jlongArray out) {
  (void)cls;
  vsrmc_level_info info;
  int32_t rc = vsrmc_checker_step((vsrmc_checker*)(intptr_t)checker, &info);
  jlong v[7] = {info.level, (jlong)info.n_new, (jlong)info.distinct, (jlong)info.total_generated, info.viol_mask, (jlong)info.viol_index,
                (jlong)info.deadlocks};
  (*env)->SetLongArrayRegion(env, out, 0, 7, v);
  return rc;
}

/* "State k: <Action>\n<state in TLC's value syntax>" for every state of a path given as wire records */
static jobjectArray states_to_strings(JNIEnv* env, const vsrmc_model* m, const uint64_t* words, const uint64_t* off, const int32_t* acts,
                                      uint64_t n) {
  jobjectArray arr = (*env)->NewObjectArray(env, (jsize)n, (*env)->FindClass(env, "java/lang/String"), NULL);
  for (uint64_t t = 0; t < n; t++) {
    int64_t need = 0;
    vsrmc_model_format_state(m, words + off[t], NULL, 0, &need);
    const char* name = vsrmc_action_name(acts[t]);
    size_t head = strlen(name) + 48;
    char* buf = (char*)malloc((size_t)need + head);
    int k = snprintf(buf, head, "State %llu: <%s>\n", (unsigned long long)(t + 1), name);
    vsrmc_model_format_state(m, words + off[t], buf + k, need, &need);
    (*env)->SetObjectArrayElement(env, arr, (jsize)t, (*env)->NewStringUTF(env, buf));
    free(buf);
  }
  return arr;
}

JNIEXPORT jobjectArray JNICALL Java_tlc2_tool_GpuModelChecker_checkerTrace(JNIEnv* env, jclass cls, jlong checker, jlong model, jint level,
                                                                           jlong index) {
  (void)cls;
  vsrmc_layout lay;
  vsrmc_model_info((const vsrmc_model*)(intptr_t)model, &lay);
  uint64_t cap_states = (uint64_t)level + 2, cap_words = cap_states * (uint64_t)lay.max_record_words, n = 0;
  uint64_t* words = (uint64_t*)malloc(cap_words * 8);
  uint64_t* off = (uint64_t*)malloc(cap_states * 8);
  int32_t* acts = (int32_t*)malloc(cap_states * 4);
  jobjectArray arr = NULL;
  if (vsrmc_checker_trace((vsrmc_checker*)(intptr_t)checker, level, (uint64_t)index, words, cap_words, off, acts, cap_states, &n) == 0)
    arr = states_to_strings(env, (const vsrmc_model*)(intptr_t)model, words, off, acts, n);
  free(words); free(off); free(acts);
  return arr;
}

/* random simulation; NULL = no violation within max_seconds, else the violating behaviour as strings */
JNIEXPORT jobjectArray JNICALL Java_tlc2_tool_GpuModelChecker_simulate(JNIEnv* env, jclass cls, jlong model, jint device, jint walkers,
                                                                       jint depth, jlong seed, jdouble max_seconds) {
  (void)cls;
  const vsrmc_model* m = (const vsrmc_model*)(intptr_t)model;
  vsrmc_sim_result r;
  if (vsrmc_simulate(m, device, (uint32_t)walkers, depth, (uint64_t)seed, max_seconds, &r) != 0 || r.found != 1) return NULL;
  uint64_t cap_states = (uint64_t)r.viol_steps + 3, cap_words = cap_states * 256, n = 0;
  uint64_t* words = (uint64_t*)malloc(cap_words * 8);
  uint64_t* off = (uint64_t*)malloc(cap_states * 8);
  int32_t* acts = (int32_t*)malloc(cap_states * 4);
  jobjectArray arr = NULL;
  if (vsrmc_model_replay(m, device, r.ords, r.viol_steps, words, cap_words, off, acts, cap_states, &n) == 0)
    arr = states_to_strings(env, m, words, off, acts, n);
  free(words); free(off); free(acts);
  return arr;
}

JNIEXPORT jstring JNICALL Java_tlc2_tool_GpuModelChecker_lastError(JNIEnv* env, jclass cls) {
  (void)cls;
  return (*env)->NewStringUTF(env, vsrmc_last_error());
}
#endif
#endif

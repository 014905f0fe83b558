// host_fpset.hpp — stand-alone FPSet (≙ tlc2.tool.fp.FPSet) and StateQueue (≙ tlc2.tool.queue.StateQueue) handles (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// FPSet
// ---------------------------------------------------------------------------------------------------------------
struct vsrmc_fpset {
  int device = 0;
  u64 slots = 0;
  Slot* table = nullptr;
  u64* d_size = nullptr;
  u32* d_err = nullptr;
};

namespace {
int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(VSRMC_E_HIP, "no HIP device: libvsrmc has no CPU fallback (the GPU path is the product)");
  if (device < 0 || device >= n) return fail(VSRMC_E_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  return 0;
}
}  // namespace

extern "C" {

int32_t vsrmc_fpset_create(int32_t device, int32_t log2_slots, vsrmc_fpset** out) {
  if (!out || log2_slots < 4 || log2_slots > 36) return fail(VSRMC_E_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  vsrmc_fpset* s = new vsrmc_fpset();
  s->device = device;
  s->slots = (u64)1 << log2_slots;
  hipError_t e = hipMalloc((void**)&s->table, s->slots * sizeof(Slot));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_size, 16);
  if (e != hipSuccess) { delete s; return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  s->d_err = (u32*)(s->d_size + 1);
  HIPCHK(hipMemset(s->table, 0, s->slots * sizeof(Slot)));
  HIPCHK(hipMemset(s->d_size, 0, 16));
  *out = s;
  return 0;
}

int32_t vsrmc_fpset_put_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_was, void* stream) {
  if (!s) return fail(VSRMC_E_ARG, "NULL handle");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(k_fpset_put, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->table, s->slots - 1,
                     d_fps, n, d_was, s->d_size, s->d_err);
  HIPCHK(hipGetLastError());
  return 0;
}
int32_t vsrmc_fpset_contains_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_present, void* stream) {
  if (!s) return fail(VSRMC_E_ARG, "NULL handle");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(k_fpset_contains, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->table,
                     s->slots - 1, d_fps, n, d_present);
  HIPCHK(hipGetLastError());
  return 0;
}

static int fpset_host_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* res, bool put) {
  if (!s || (n && (!fps || !res))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  u64* d_fps = nullptr;
  uint8_t* d_res = nullptr;
  HIPCHK(hipMalloc((void**)&d_fps, n * 8));
  hipError_t e = hipMalloc((void**)&d_res, n);
  if (e != hipSuccess) { (void)hipFree(d_fps); return fail(VSRMC_E_HIP, "hipMalloc"); }
  int rc = 0;
  e = hipMemcpy(d_fps, fps, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = put ? vsrmc_fpset_put_batch_device(s, d_fps, n, d_res, nullptr) : vsrmc_fpset_contains_batch_device(s, d_fps, n, d_res, nullptr);
    if (!rc) e = hipMemcpy(res, d_res, n, hipMemcpyDeviceToHost);
  }
  u32 err = 0;
  if (e == hipSuccess) e = hipMemcpy(&err, s->d_err, 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_fps);
  (void)hipFree(d_res);
  if (rc) return rc;
  if (e != hipSuccess) return fail(VSRMC_E_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  if (err) return fail(VSRMC_E_REP, "fingerprint set is full");
  return 0;
}
int32_t vsrmc_fpset_put_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* was_present) {
  return fpset_host_batch(s, fps, n, was_present, true);
}
int32_t vsrmc_fpset_contains_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* present) {
  return fpset_host_batch(s, fps, n, present, false);
}
int32_t vsrmc_fpset_size(vsrmc_fpset* s, uint64_t* size) {
  if (!s || !size) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipMemcpy(size, s->d_size, 8, hipMemcpyDeviceToHost));
  return 0;
}
void vsrmc_fpset_destroy(vsrmc_fpset* s) {
  if (!s) return;
  if (s->table) (void)hipFree(s->table);
  if (s->d_size) (void)hipFree(s->d_size);
  delete s;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// StateQueue: a ring of record words + a ring of (offset, length) refs, both in HBM
// ---------------------------------------------------------------------------------------------------------------
struct vsrmc_queue {
  int device = 0;
  u64 cap_words = 0, cap_states = 0;
  u64* words = nullptr;
  std::vector<std::pair<u64, u32>> refs;   // (word position in the ring, length) of every queued record, oldest at `head`
  u64 head = 0;
  u64 wpos = 0;                            // next write position in the word ring
  u64 wtail() const { return head < refs.size() ? refs[head].first : wpos; }   // position of the oldest record
};

extern "C" {

int32_t vsrmc_queue_create(int32_t device, uint64_t capacity_words, uint64_t capacity_states, vsrmc_queue** out) {
  if (!out || capacity_words < 256 || capacity_states < 1) return fail(VSRMC_E_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  vsrmc_queue* q = new vsrmc_queue();
  q->device = device;
  q->cap_words = capacity_words;
  q->cap_states = capacity_states;
  hipError_t e = hipMalloc((void**)&q->words, capacity_words * 8);
  if (e != hipSuccess) { delete q; return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *out = q;
  return 0;
}

int32_t vsrmc_queue_enqueue_batch(vsrmc_queue* q, const uint64_t* words, const uint64_t* off, uint64_t n) {
  if (!q || (n && (!words || !off))) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(q->device));
  if (q->refs.size() - q->head + n > q->cap_states) return fail(VSRMC_E_REP, "state queue full (states)");
  for (u64 i = 0; i < n; i++) {
    const u64 len = off[i + 1] - off[i];
    if (len == 0 || len > 255) return fail(VSRMC_E_ARG, "bad record length");
    if (q->head == q->refs.size()) { q->refs.clear(); q->head = 0; q->wpos = 0; }   // empty: start over at the front of the ring
    const u64 tail = q->wtail();
    u64 at;
    if (q->wpos >= tail) {                                    // data in [tail, wpos): append, or wrap to the front
      if (q->wpos + len <= q->cap_words) at = q->wpos;
      else if (len < tail) at = 0;                            // records never straddle the end of the ring
      else return fail(VSRMC_E_REP, "state queue full (words)");
    } else {                                                  // wrapped: data in [tail, cap) and [0, wpos)
      if (q->wpos + len < tail) at = q->wpos;
      else return fail(VSRMC_E_REP, "state queue full (words)");
    }
    HIPCHK(hipMemcpy(q->words + at, words + off[i], len * 8, hipMemcpyHostToDevice));
    q->refs.emplace_back(at, (u32)len);
    q->wpos = at + len;
  }
  return 0;
}

int32_t vsrmc_queue_dequeue_batch(vsrmc_queue* q, uint64_t max_states, uint64_t* words, uint64_t cap_words, uint64_t* off, uint64_t* n) {
  if (!q || !words || !off || !n) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(q->device));
  u64 k = 0, pos = 0;
  off[0] = 0;
  while (k < max_states && q->head < q->refs.size()) {
    const u64 wp = q->refs[q->head].first;
    const u32 len = q->refs[q->head].second;
    if (pos + len > cap_words) break;
    HIPCHK(hipMemcpy(words + pos, q->words + wp, (u64)len * 8, hipMemcpyDeviceToHost));
    pos += len;
    off[++k] = pos;
    q->head++;
  }
  if (q->head > 4096 && q->head * 2 > q->refs.size()) {       // drop the consumed prefix of the ref list now and then
    q->refs.erase(q->refs.begin(), q->refs.begin() + (long)q->head);
    q->head = 0;
  }
  *n = k;
  return 0;
}

int32_t vsrmc_queue_size(vsrmc_queue* q, uint64_t* n_states) {
  if (!q || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  *n_states = q->refs.size() - q->head;
  return 0;
}

void vsrmc_queue_destroy(vsrmc_queue* q) {
  if (!q) return;
  (void)hipSetDevice(q->device);
  if (q->words) (void)hipFree(q->words);
  delete q;
}

}  // extern "C"


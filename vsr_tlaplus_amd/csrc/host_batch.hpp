// host_batch.hpp — host-buffer entry points: expand / fingerprint batches, TLC trace import, simulation mode (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// expand_batch / fingerprint_batch
// ---------------------------------------------------------------------------------------------------------------
namespace {

// upload n wire records as device-layout records with their H words filled in
int upload_records(const Model& M, const u64* words, const u64* off, u64 n, u64** d_words, u64** d_off, u64* total_words) {
  std::vector<u64> dev, doff(n + 1);
  dev.reserve((size_t)(off[n] + n * M.np));
  std::vector<u64> tmp(512);
  for (u64 i = 0; i < n; i++) {
    doff[i] = dev.size();
    const u64* w = words + off[i];
    int nmsg = hdr_nmsg(w[0]);
    if ((u64)(M.h0 + nmsg) != off[i + 1] - off[i]) return fail(VSRMC_E_ARG, "record length does not match its header");
    if (nmsg > M.max_bag) return fail(VSRMC_E_REP, "record bag larger than max_bag");
    int len = wire_to_device(M, w, tmp.data());
    dev.insert(dev.end(), tmp.begin(), tmp.begin() + len);
  }
  doff[n] = dev.size();
  *total_words = dev.size();
  HIPCHK(hipMalloc((void**)d_words, std::max<size_t>(dev.size(), 1) * 8));
  HIPCHK(hipMalloc((void**)d_off, (n + 1) * 8));
  HIPCHK(hipMemcpy(*d_words, dev.data(), dev.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(*d_off, doff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((M.model_id == 1 ? k_hash_records<1> : M.model_id == 2 ? k_hash_records<2> : k_hash_records<0>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, M, *d_words, *d_off, n);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int32_t vsrmc_expand_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                           uint64_t* out_words, uint64_t out_words_cap, uint64_t* out_meta, uint64_t out_cap,
                           uint64_t* n_out, uint64_t* words_out) {
  if (!m || !words || !off || !out_words || !out_meta || !n_out || !words_out) return fail(VSRMC_E_ARG, "NULL argument");
  int rc = check_device(device);
  if (rc) return rc;
  const Model& M = m->M;
  *n_out = 0;
  *words_out = 0;
  if (n == 0) return 0;
  u64 *d_words = nullptr, *d_off = nullptr, *d_ow = nullptr, *d_om = nullptr, *d_cnt = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  u64 dev_words_cap = out_words_cap + out_cap * (u64)M.np;
  HIPCHK(hipMalloc((void**)&d_ow, std::max<u64>(dev_words_cap, 1) * 8));
  HIPCHK(hipMalloc((void**)&d_om, std::max<u64>(out_cap, 1) * 64));
  HIPCHK(hipMalloc((void**)&d_cnt, 32));
  HIPCHK(hipMemset(d_cnt, 0, 32));
  hipLaunchKernelGGL((M.model_id == 1 ? k_successors<1> : M.model_id == 2 ? k_successors<2> : k_successors<0>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, M, d_words, d_off, n, d_ow, dev_words_cap,
                     d_om, out_cap, d_cnt);
  HIPCHK(hipGetLastError());
  u64 cnt[4];
  HIPCHK(hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost));
  int ret = 0;
  if (cnt[2]) {
    ret = fail(VSRMC_E_ARG, "successor buffers too small");
  } else {
    std::vector<u64> hw(cnt[1]), hm(cnt[0] * 8);
    HIPCHK(hipMemcpy(hw.data(), d_ow, cnt[1] * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hm.data(), d_om, cnt[0] * 64, hipMemcpyDeviceToHost));
    // deterministic (parent, ordinal) order
    std::vector<u64> order(cnt[0]);
    for (u64 k = 0; k < cnt[0]; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](u64 a, u64 b) {
      if (hm[8 * a] != hm[8 * b]) return hm[8 * a] < hm[8 * b];
      return hm[8 * a + 1] < hm[8 * b + 1];
    });
    u64 wpos = 0;
    for (u64 k = 0; k < cnt[0] && !ret; k++) {
      const u64* mm = &hm[8 * order[k]];
      const u64* dev = &hw[mm[7]];
      int evalerr = (int)mm[6];
      int wl = evalerr ? 0 : M.h0 + hdr_nmsg(dev[0]);
      if (wpos + (u64)wl > out_words_cap) { ret = fail(VSRMC_E_ARG, "successor word buffer too small"); break; }
      if (!evalerr) device_to_wire(M, dev, out_words + wpos);
      for (int q = 0; q < 7; q++) out_meta[8 * k + q] = mm[q];
      out_meta[8 * k + 7] = wpos;
      wpos += (u64)wl;
    }
    *n_out = cnt[0];
    *words_out = wpos;
  }
  (void)hipFree(d_words); (void)hipFree(d_off); (void)hipFree(d_ow); (void)hipFree(d_om); (void)hipFree(d_cnt);
  return ret;
}

// ---- TLC trace / state import: text in TLC's value syntax -> wire records ---------------------------------
int32_t vsrmc_model_parse_states(const vsrmc_model* m, const char* text, uint64_t* words, uint64_t cap_words, uint64_t* off,
                                 int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!m || !text || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  std::vector<ParsedState> st;
  std::string err;
  if (!parse_states_tlc(m->M, m->value_names, text, &st, &err)) return fail(VSRMC_E_CFG, "TLC state text: " + err);
  *n_states = st.size();
  u64 total = 0;
  for (const ParsedState& s : st) total += s.rec.size();
  if (!words || !off) return 0;                                 // size query
  if (cap_states < st.size() + 1 || cap_words < total) return fail(VSRMC_E_ARG, "buffer too small");
  u64 pos = 0;
  for (size_t i = 0; i < st.size(); i++) {
    off[i] = pos;
    std::copy(st[i].rec.begin(), st[i].rec.end(), words + pos);
    pos += st[i].rec.size();
    if (actions) {
      actions[i] = -1;
      for (int a = 0; a < 16; a++)
        if (st[i].action == vsrmc_action_name(a)) actions[i] = a;
    }
  }
  off[st.size()] = pos;
  return 0;
}

// Is the sequence of states a behaviour of the model?  State 0 must be Init; every later state must be among the successors
// the GPU generates for its predecessor (k_successors, the same gen() the BFS kernels run).  ords[i] / actions[i + 1] = the
// (action, binding) ordinal and action id of the step into state i + 1; *first_bad = index of the first state that does not
// follow (or -1); *inv_mask_last = invariants violated by the last state (when the whole sequence is legal).
int32_t vsrmc_model_check_trace(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n_states,
                                uint32_t* ords, int32_t* actions, int64_t* first_bad, int32_t* inv_mask_last) {
  if (!m || !words || !off || !first_bad) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  *first_bad = -1;
  if (inv_mask_last) *inv_mask_last = 0;
  if (n_states == 0) return 0;
  auto normal = [&](const u64* rec, u64 len) {
    std::vector<u64> v(rec, rec + len);
    if (len > (u64)M.h0) std::sort(v.begin() + M.h0, v.end());
    return v;
  };
  std::vector<u64> init;
  init_record_wire(M, init);
  if (normal(words + off[0], off[1] - off[0]) != init) {
    *first_bad = 0;
    return 0;
  }
  if (actions) actions[0] = 0;
  if (n_states == 1) return 0;
  // Walk like vsrmc_model_replay does: the state that is expanded next is the successor as the GPU produced it (its bag order
  // defines the ordinals of the message-bound actions), the text's states are only compared against.
  const u64 cap = 2048, capw = cap * 64;
  std::vector<u64> ow(capw), om(cap * 8), cur = init;
  for (u64 i = 0; i + 1 < n_states; i++) {
    const std::vector<u64> want = normal(words + off[i + 1], off[i + 2] - off[i + 1]);
    const u64 coff[2] = {0, cur.size()};
    u64 n_out = 0, w_out = 0;
    int rc = vsrmc_expand_batch(m, device, cur.data(), coff, 1, ow.data(), capw, om.data(), cap, &n_out, &w_out);
    if (rc) return rc;
    bool found = false;
    for (u64 k = 0; k < n_out && !found; k++) {
      if (om[8 * k + 6]) continue;                             // an instance that raises an evaluation error has no successor
      const u64 wo = om[8 * k + 7];
      const u64 len = (u64)M.h0 + (u64)hdr_nmsg(ow[wo]);
      if (normal(&ow[wo], len) != want) continue;
      found = true;
      if (ords) ords[i] = (uint32_t)om[8 * k + 1];
      if (actions) actions[i + 1] = (int32_t)om[8 * k + 2];
      if (i + 2 == n_states && inv_mask_last) *inv_mask_last = (int32_t)om[8 * k + 5];
      cur.assign(&ow[wo], &ow[wo] + len);
    }
    if (!found) {
      *first_bad = (int64_t)(i + 1);
      return 0;
    }
  }
  return 0;
}

int32_t vsrmc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                                uint64_t* fps, uint32_t* auxkeys) {
  if (!m || !words || !off || !fps) return fail(VSRMC_E_ARG, "NULL argument");
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return 0;
  const Model& M = m->M;
  u64 *d_words = nullptr, *d_off = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  std::vector<u64> dev(total), doff(n + 1);
  HIPCHK(hipMemcpy(dev.data(), d_words, total * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(doff.data(), d_off, (n + 1) * 8, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < n; i++) {   // the hashes were computed on the GPU (k_hash_records); only the final min is here
    u64 fp;
    u32 ak;
    canonical_fp(M, dev[doff[i]], &dev[doff[i] + M.h0], &fp, &ak);
    fps[i] = fp;
    if (auxkeys) auxkeys[i] = ak;
  }
  (void)hipFree(d_words); (void)hipFree(d_off);
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// simulation mode
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t vsrmc_simulate(const vsrmc_model* m, int32_t device, uint32_t n_walkers, int32_t max_depth, uint64_t seed,
                                  double max_seconds, vsrmc_sim_result* out) {
  if (!m || !out || n_walkers == 0 || max_depth < 1 || max_depth > 512) return fail(VSRMC_E_ARG, "bad argument (max_depth 1..512)");
  int rc = check_device(device);
  if (rc) return rc;
  Model M = m->M;
  M.max_bag = 255 - M.fixed;     // walkers live in HBM, not in LDS tiles: the bag may grow to what the 8-bit count can hold
  std::memset(out, 0, sizeof(*out));
  const int stride = M.fixed + M.max_bag;
  std::vector<u64> wire, dev(512);
  init_record_wire(M, wire);
  const int len = wire_to_device(M, wire.data(), dev.data());   // the H words stay 0: simulation never fingerprints
  u64 *d_init = nullptr, *d_words = nullptr, *d_rng = nullptr;
  u32* d_depth = nullptr;
  u16* d_ords = nullptr;
  SimCtl* d_ctl = nullptr;
  HIPCHK(hipMalloc((void**)&d_init, 512 * 8));
  HIPCHK(hipMalloc((void**)&d_words, (u64)n_walkers * stride * 8));
  HIPCHK(hipMalloc((void**)&d_rng, (u64)n_walkers * 8));
  HIPCHK(hipMalloc((void**)&d_depth, (u64)n_walkers * 4));
  HIPCHK(hipMalloc((void**)&d_ords, (u64)n_walkers * max_depth * 2));
  HIPCHK(hipMalloc((void**)&d_ctl, sizeof(SimCtl)));
  HIPCHK(hipMemcpy(d_init, dev.data(), len * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_depth, 0xFF, (u64)n_walkers * 4));
  HIPCHK(hipMemset(d_ctl, 0, sizeof(SimCtl)));
  std::vector<u64> rng(n_walkers);
  u64 x = seed;
  for (u32 i = 0; i < n_walkers; i++) {   // splitmix64 stream: one non-zero xorshift state per walker
    x += 0x9E3779B97F4A7C15ULL;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    rng[i] = z ? z : 1;
  }
  HIPCHK(hipMemcpy(d_rng, rng.data(), (u64)n_walkers * 8, hipMemcpyHostToDevice));
  SimCtl h;
  typedef void (*SimKernel)(Model, const u64*, int, u64*, int, u32*, u16*, u64*, u32, int, int, SimCtl*);
  SimKernel sim_kernel = k_simulate<0>;
  if (M.model_id == 1) sim_kernel = k_simulate<1000>;
  if (M.model_id == 2) sim_kernel = k_simulate<2000>;
  else
  switch (M.R * 100 + M.C * 10 + M.n) {                        // the same per-configuration instantiations as k_expand
    case 312: sim_kernel = k_simulate<312>; break;
    case 313: sim_kernel = k_simulate<313>; break;
    case 512: sim_kernel = k_simulate<512>; break;
    default: break;
  }
  double t0 = now_s();
  while (true) {
    hipLaunchKernelGGL(sim_kernel, dim3((n_walkers + 63) / 64), dim3(64), 0, 0, M, d_init, len, d_words, stride, d_depth, d_ords, d_rng,
                       n_walkers, max_depth, 64, d_ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(&h, d_ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h.found || now_s() - t0 > max_seconds) break;
  }
  out->seconds = now_s() - t0;
  out->found = (int32_t)h.found;
  out->steps = h.steps;
  out->walks = h.walks;
  if (h.found) {
    out->viol_mask = (int32_t)(h.viol_mask & 0x7FFFFFFFu);
    out->viol_steps = (int32_t)h.viol_depth;
    for (u32 k = 0; k < h.viol_depth && k < 512; k++) out->ords[k] = h.ords[k];
  }
  (void)hipFree(d_init); (void)hipFree(d_words); (void)hipFree(d_rng); (void)hipFree(d_depth); (void)hipFree(d_ords); (void)hipFree(d_ctl);
  return 0;
}


// host_tlcfp.hpp — TLC's own FP64 fingerprint as a characterisation mode (csrc/vsr_tlcfp.hpp) (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// TLC-style fingerprints (FP64 over TLC's serialisation of the view): a characterisation mode, never the identity the seen-set uses
// ---------------------------------------------------------------------------------------------------------------
#include "vsr_tlcfp.hpp"

extern "C" {

int32_t vsrmc_tlc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n, uint64_t* fps) {
  if (!m || !words || !off || !fps) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return 0;
  for (u64 i = 0; i < n; i++)
    if (hdr_nmsg(words[off[i]]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  u64 *d_words = nullptr, *d_off = nullptr, *d_out = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  bool ok = hipMalloc((void**)&d_out, n * 8) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(vsr::tlcfp::k_tlc_fingerprints, dim3((unsigned)std::min<u64>((n + 255) / 256, 4096)), dim3(256), 0, 0, M, d_words, d_off, (const u64*)nullptr, n, d_out);
    ok = hipGetLastError() == hipSuccess && hipMemcpy(fps, d_out, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d_words); (void)hipFree(d_off); (void)hipFree(d_out);
  return ok ? 0 : fail(VSRMC_E_HIP, "k_tlc_fingerprints failed");
}

int32_t vsrmc_tlc_view_bytes(const vsrmc_model* m, const uint64_t* record, int32_t permutation, uint8_t* out, uint64_t cap, uint64_t* n_bytes) {
  if (!m || !record || !n_bytes || (!out && cap)) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  if (permutation < 0 || permutation >= M.np) return fail(VSRMC_E_ARG, "no such permutation");
  if (hdr_nmsg(record[0]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  vsr::tlcfp::ByteSink s{out, cap, 0};
  vsr::tlcfp::put_view(M, s, record, record + M.h0, M.pitab[permutation]);     // wire layout: the bag follows the replica blocks
  *n_bytes = s.n;
  return 0;
}

int32_t vsrmc_tlc_min_permutation(const vsrmc_model* m, const uint64_t* record, int32_t* permutation) {
  if (!m || !record || !permutation) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  if (hdr_nmsg(record[0]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  *permutation = vsr::tlcfp::min_permutation(M, record, record + M.h0);
  return 0;
}

uint64_t vsrmc_fp64_extend(uint64_t fp, const uint8_t* bytes, uint64_t n) {
  vsr::tlcfp::FpSink s{fp, vsr::tlcfp::H_TABLE.t};
  for (u64 k = 0; k < n; k++) s.byte(bytes[k]);
  return s.fp;
}

uint64_t vsrmc_fp64_new(void) { return vsr::tlcfp::IRRED_POLY; }

int32_t vsrmc_checker_tlc_level_fps(vsrmc_checker* c, uint64_t* out, uint64_t cap, uint64_t* n, double* kernel_ms) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = c->model.M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  *n = c->n_valid;
  if (c->n_frontier == 0) return 0;
  if (out && cap < c->n_valid) return fail(VSRMC_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  u64* d_out = nullptr;
  HIPCHK(hipMalloc((void**)&d_out, c->n_frontier * 8));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, c->stream) == hipSuccess;
  hipLaunchKernelGGL(vsr::tlcfp::k_tlc_fingerprints, dim3((unsigned)std::min<u64>((c->n_frontier + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, M,
                     (const u64*)c->words[c->cur], (const u64*)nullptr, (const u64*)c->off[c->cur], c->n_frontier, d_out);
  ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
  float ms = 0;
  if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (ok && out) {
    std::vector<u64> all(c->n_frontier);
    ok = hipMemcpy(all.data(), d_out, c->n_frontier * 8, hipMemcpyDeviceToHost) == hipSuccess;
    std::vector<u64> refs(c->n_frontier);
    ok = ok && hipMemcpy(refs.data(), c->off[c->cur], c->n_frontier * 8, hipMemcpyDeviceToHost) == hipSuccess;
    u64 k = 0;
    for (u64 i = 0; i < c->n_frontier && ok; i++)
      if (refs[i] != 0 && k < cap) out[k++] = all[i];
    *n = k;
    std::sort(out, out + k);
  }
  (void)hipFree(d_out);
  return ok ? 0 : fail(VSRMC_E_HIP, "k_tlc_fingerprints failed");
}

}  // extern "C"


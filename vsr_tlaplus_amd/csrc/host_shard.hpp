// host_shard.hpp — the phases of a sharded level behind the C ABI (vsrmc_shard_*): expand, claim, verdicts, rebalancing, commit (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

extern "C" {
// ---- sharded protocol: one level = expand -> [exchange] -> claim -> [exchange] -> materialize -> [exchange] -> append -> commit
int32_t vsrmc_shard_expand(vsrmc_checker* c, const vsrmc_shard_io* io, uint64_t* cand_counts) {
  if (!c || !io || !cand_counts) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  int rc = phase_expand(c, io);
  for (int o = 0; o < c->opt.world; o++) cand_counts[o] = c->h.cand_cnt[o];
  return rc;
}

int32_t vsrmc_shard_claim(vsrmc_checker* c, const uint64_t* d_cand_recv, uint64_t n, uint8_t* d_verdict) {
  if (!c || (n && (!d_cand_recv || !d_verdict))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  if (c->opt.exact_ties && n > c->rslot_cap) {
    if (c->rslot) (void)hipFree(c->rslot);
    c->rslot = nullptr;
    c->rslot_cap = 0;
    HIPCHK(hipMalloc((void**)&c->rslot, n * 8));
    c->rslot_cap = n;
  }
  unsigned grid = (unsigned)((n + 255) / 256);
  if (!c->opt.exact_ties) {   // single-pass level: the inserting candidate wins, the verdict is known at once
    hipLaunchKernelGGL(k_claim_batch_fused, dim3(grid), dim3(256), 0, c->stream, c->table, c->tmask, d_cand_recv, n, c->level + 1, d_verdict,
                       c->ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_claim_batch, dim3(grid), dim3(256), 0, c->stream, c->table, c->tmask, d_cand_recv, n, c->level + 1, c->rslot, c->ctl);
  hipLaunchKernelGGL(k_verdict, dim3(grid), dim3(256), 0, c->stream, c->table, d_cand_recv, c->rslot, n, d_verdict);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// NOTE: every rank must have finished vsrmc_shard_claim for ALL its received candidates before any verdict is used:
// the verdict of a slot is final only when every claim of the level has landed (the orchestrator's exchange is the barrier).
int32_t vsrmc_shard_materialize(vsrmc_checker* c, const vsrmc_shard_io* io, const uint8_t* d_verdict_in) {
  if (!c || !io) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  // every winner — local owner or remote verdict — is written into THIS rank's next frontier: records stay with their
  // generator, only 16-byte candidates and verdict bytes cross ranks (rebalancing moves records in bulk when needed)
  if (c->level_fused) {
    // single-pass level: k_expand wrote the announced successors speculatively; withdraw the ones whose owner said no
    HIPCHK(hipSetDevice(c->opt.device));
    const int nxt = c->cur ^ 1;
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    for (int o = 0; o < c->opt.world; o++) {
      if (o == c->opt.rank) continue;
      const u64 n = std::min<u64>(c->h.cand_cnt[o], io->cand_cap);
      if (n == 0) continue;
      if (!d_verdict_in) return fail(VSRMC_E_ARG, "verdicts missing");
      hipLaunchKernelGGL(k_apply_verdict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, io->cand_send + 2 * (u64)o * io->cand_cap,
                         c->cand_idx + (u64)o * io->cand_cap, d_verdict_in + (u64)o * io->cand_cap, n, c->off[nxt], c->lvl_fp, c->ctl, (const WSet*)nullptr, 0);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    c->materialize_ms += ms;
    if (c->h.err) return level_error(c, c->h, c->level + 1);
    if (c->h.ties) {
      c->failed = 1;
      return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2): "
                                 "create the checker with vsrmc_options.exact_ties = 1");
    }
    c->nx_n = c->h.n_new;
    c->nx_w = c->h.words_new;
    return 0;
  }
  int rc = phase_materialize_local(c);
  const int nxt = c->cur ^ 1;
  const u64 nx_cap = c->opt.frontier_states;
  for (int o = 0; o < c->opt.world && !rc; o++) {
    if (o == c->opt.rank) continue;
    u64 n = std::min<u64>(c->h.cand_cnt[o], io->cand_cap);
    if (n && !d_verdict_in) return fail(VSRMC_E_ARG, "verdicts missing");
    rc = phase_materialize(c, io->cand_send + 2 * (u64)o * io->cand_cap, n, d_verdict_in + (u64)o * io->cand_cap, c->words[nxt],
                           c->words_cap(nxt), c->off[nxt], nx_cap, c->lvl_fp, &c->ctl->n_new, &c->ctl->words_new, 2,
                           c->cand_idx + (u64)o * io->cand_cap);
  }
  if (rc) return rc;
  HIPCHK(hipMemcpy(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost));
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  c->nx_n = c->h.n_new;
  c->nx_w = c->h.words_new;
  return 0;
}

int32_t vsrmc_shard_count(vsrmc_checker* c, uint64_t* n_valid, uint64_t* n_range) {
  if (!c || !n_valid || !n_range) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  *n_range = c->nx_n;
  *n_valid = 0;
  if (c->nx_n == 0) return 0;
  u64 zero = 0;
  HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_count_valid, dim3(1024), dim3(256), 0, c->stream, c->off[c->cur ^ 1], c->nx_n, c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(n_valid, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int32_t vsrmc_shard_export(vsrmc_checker* c, uint64_t first, uint64_t n, uint64_t* d_words, uint64_t words_cap, uint64_t* d_off,
                           uint64_t* d_fp, uint64_t cap, uint64_t* n_out, uint64_t* words_out) {
  if (!c || !n_out || !words_out) return fail(VSRMC_E_ARG, "NULL argument");
  *n_out = *words_out = 0;
  if (n == 0) return 0;
  if (!d_words || !d_off || !d_fp || first + n > c->nx_n) return fail(VSRMC_E_ARG, "bad export window");
  HIPCHK(hipSetDevice(c->opt.device));
  const int nxt = c->cur ^ 1;
  u64* d_cnt = nullptr;
  HIPCHK(hipMalloc((void**)&d_cnt, 32));
  HIPCHK(hipMemsetAsync(d_cnt, 0, 32, c->stream));
  hipLaunchKernelGGL(k_export, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->words[nxt], c->off[nxt] + 0, c->lvl_fp,
                     first, n, d_words, words_cap, d_off, d_fp, cap, d_cnt, (u32*)(d_cnt + 2));
  HIPCHK(hipGetLastError());
  u64 h[4];
  HIPCHK(hipMemcpyAsync(h, d_cnt, 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)hipFree(d_cnt);
  if ((u32)h[2]) return fail(VSRMC_E_REP, "export buffers too small");
  *n_out = h[0];
  *words_out = h[1];
  return 0;
}

int32_t vsrmc_shard_append(vsrmc_checker* c, const uint64_t* d_words, uint64_t n_words, const uint64_t* d_off,
                           const uint64_t* d_fp, uint64_t n) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  if (!d_words || !d_off || !d_fp) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  const u64 nx_cap = c->opt.frontier_states;
  if (c->nx_n + n > nx_cap || c->nx_w + n_words > c->words_cap(c->cur ^ 1)) {
    c->failed = 1;
    return fail(VSRMC_E_REP, "frontier buffers full while appending received records");
  }
  const int nxt = c->cur ^ 1;
  HIPCHK(hipMemcpyAsync(c->words[nxt] + c->nx_w, d_words, n_words * 8, hipMemcpyDefault, c->stream));
  hipLaunchKernelGGL(k_append_fixup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->off[nxt] + c->nx_n,
                     c->lvl_fp + c->nx_n, d_off, d_fp, n, c->nx_w);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  c->nx_n += n;
  c->nx_w += n_words;
  return 0;
}

int32_t vsrmc_shard_commit(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  return phase_commit(c, info);
}

}  // extern "C"

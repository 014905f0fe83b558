// host_search.hpp — probe level, the automatic level scheme (vsrmc_checker_advance / vsrmc_check), the replicated phase of a sharded run (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

extern "C" {

// Probe level: expand the newest level WITHOUT storing its successors — every successor that is not a state of an earlier
// level gets its invariants checked, nothing is inserted into the seen-set, no frontier is written.  The search cannot
// continue afterwards (the level does not exist), but a violation one level beyond what memory can hold is found and its
// counter-example reconstructed (vsrmc_checker_probe_trace).  Also valid right after a step that failed with "frontier full".
int32_t vsrmc_checker_probe(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  // sharded: the rank probes its part of the newest level against ITS part of the seen-set; the violating successors it could
  // not find there (vsrmc_checker_probe_candidates) still have to be shown to their owners (sharded.py: ShardedChecker.probe)
  if (c->opt.exact_ties) return fail(VSRMC_E_STATE, "probe levels need a single-pass checker");
  if (c->failed && c->failed_code != ERR_FRONTIER_FULL) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  c->failed = 0;
  c->probe_fp = 0;
  c->probe_level = 0;
  c->probe_extra_fp = 0;
  int rc = phase_expand(c, nullptr, MODE_PROBE);
  if (rc) return rc;
  std::memset(info, 0, sizeof(*info));
  info->level = c->level + 1;
  info->frontier = c->n_frontier;
  info->generated = c->h.generated;
  info->deadlocks = c->h.deadlocks;
  info->probes = c->h.probes;
  info->pending = c->h.n_pending;                               // violating successors seen (duplicates included)
  info->distinct = c->distinct;
  info->total_generated = c->total_generated + c->h.generated;
  info->expand_ms = c->expand_ms;
  info->seconds = now_s() - c->t_level0;
  info->viol_fp = ~(u64)0;
  info->viol_index = ~(u64)0;
  for (int a = 0; a < 16; a++) info->act_generated[a] = c->h.act_generated[a];
  for (int a = 0; a < 8; a++) info->phase_cycles[a] = c->h.phase_cycles[a];
  if (c->h.viol_fp != ~(u64)0) {
    info->viol_fp = c->h.viol_fp;
    info->viol_mask = (int32_t)c->h.viol_mask;
    // the violator that is reported: smallest fingerprint; among its (fp, key) entries the smallest key
    u64 key = ~(u64)0;
    rc = min_violator(c, c->h.viol_fp, &key);
    if (rc) return rc;
    if (key != ~(u64)0 && c->opt.world <= 1) {                  // its parent: the newest level's state with these fingerprint bits
      int found = 0;
      u64 pfp = 0, pmeta = 0;
      rc = table_lookup(c, meta_pfp(key), c->level, 1, &found, &pfp, &pmeta);
      if (rc) return rc;
      if (found > 1) return fail(VSRMC_E_STATE, "ambiguous predecessor pointer: several states of the parent's level share the 45 fingerprint bits the violating successor keeps of its parent");
      if (found) {
        c->probe_fp = pfp;
        c->probe_level = c->level;
        c->probe_extra_fp = c->h.viol_fp;
      }
    }
  }
  return 0;
}

int32_t vsrmc_checker_probe_candidates(vsrmc_checker* c, uint64_t* pairs, uint64_t cap_pairs, uint64_t* n) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  *n = c->h.n_pending;
  if (c->h.n_pending > c->opt.pending_entries) return fail(VSRMC_E_REP, "more violating successors than the pending list holds (pending_entries)");
  if (c->h.n_pending == 0 || !pairs) return 0;                 // pairs == NULL: only the number is asked for
  if (cap_pairs < c->h.n_pending) return fail(VSRMC_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  HIPCHK(hipMemcpy(pairs, c->pending, 16 * c->h.n_pending, hipMemcpyDeviceToHost));
  return 0;
}

int32_t vsrmc_checker_seen_batch(vsrmc_checker* c, const uint64_t* fps, uint64_t n, int32_t level, uint8_t* seen) {
  if (!c || (n && (!fps || !seen))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 16 * n));
  std::vector<u64> flags(n, 0);
  hipError_t e = hipMemcpy(d, fps, 8 * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_table_seen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->table, c->tmask, d, (u64)n, (int)level, d + n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(flags.data(), d + n, 8 * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(VSRMC_E_HIP, std::string("vsrmc_checker_seen_batch: ") + hipGetErrorString(e));
  for (u64 i = 0; i < n; i++) seen[i] = flags[i] ? 1 : 0;
  return 0;
}

int32_t vsrmc_checker_probe_trace(vsrmc_checker* c, uint64_t* words, uint64_t cap_words, uint64_t* off, int32_t* actions,
                                  uint64_t cap_states, uint64_t* n_states) {
  if (!c || !words || !off || !actions || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->probe_fp == 0) return fail(VSRMC_E_STATE, "no violation recorded by vsrmc_checker_probe");
  HIPCHK(hipSetDevice(c->opt.device));
  std::vector<u64> fps;
  int rc = walk_trace(c, c->probe_fp, c->probe_level, &fps);    // Init .. the deepest state of the path that is in the seen-set
  if (rc) return rc;
  if (c->probe_extra_fp) fps.push_back(c->probe_extra_fp);      // ... and the probed state beyond it
  return vsrmc_model_replay_fps(&c->model, c->opt.device, fps.data(), (int32_t)fps.size(), words, cap_words, off, actions, cap_states, n_states);
}

int32_t vsrmc_checker_step(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  if (c->deep) return fail(VSRMC_E_STATE, "levels beyond the record buffers exist in the seen-set (vsrmc_checker_deepen): the search goes on with vsrmc_checker_deepen / _advance");
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: drive the level with the vsrmc_shard_* phases");
  return step_local(c, info);
}

int32_t vsrmc_shard_local_step(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  return step_local(c, info);
}

// Sharded runs: the largest bag among the records of the newest level over ALL ranks (records move between ranks when the
// frontiers are rebalanced, so a rank's own maximum is not enough).  Lets the next k_expand size its LDS record slots for the
// level instead of the format's worst case; without this call the worst case is used.
int32_t vsrmc_shard_set_max_bag(vsrmc_checker* c, uint64_t max_bag) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  c->cur_max_bag = max_bag;
  c->bag_known = true;
  return 0;
}

int32_t vsrmc_shard_partition(vsrmc_checker* c, uint64_t* n_kept) {
  if (!c || !n_kept) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  *n_kept = c->n_valid;
  if (c->opt.world <= 1 || c->n_frontier == 0) return 0;
  u64 zero = 0;
  HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_partition, dim3((unsigned)((c->n_frontier + 255) / 256)), dim3(256), 0, c->stream, c->off[c->cur], c->lvl_fp,
                     c->n_frontier, c->opt.rank, c->opt.world, c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(n_kept, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n_valid = *n_kept;
  return 0;
}

// Will the next level fit the idle record buffer?  Level sizes of these models grow by a factor that FALLS from level to level once the
// search is past its first few levels (the level sizes under tests/golden/: r(l+1) / r(l) is 0.92 .. 0.99 everywhere beyond level 10), so
// the last level's factor bounds the next one's; records grow by at most one bag entry per level.  Small levels are bounded by the
// successors generated per state instead.  A wrong "yes" ends in ERR_FRONTIER_FULL, a wrong "no" only costs a level of re-expansion.
static bool next_level_fits(const vsrmc_checker* c) {
  const u64 n = c->n_valid;
  if (n == 0) return true;
  double pred;
  if (n < 32768 || c->hist_new[0] == 0) pred = (double)n * (double)std::max<u64>(2, c->g_last) * 1.25;
  else pred = (double)n * std::min((double)c->g_last, (double)c->hist_new[1] / (double)c->hist_new[0] * 1.02);
  const double wbar = (double)std::max<u64>(c->cur_rec_w, (u64)c->model.M.fixed * n) / (double)n + 1.0;
  const int nxt = c->cur ^ 1;
  const double blocks = 4.0 * c->num_cus;                        // every resident block leaves a partly used word and index chunk behind
  const double cap_w = (double)c->words_cap(nxt), cap_n = (double)c->opt.frontier_states;
  return pred * wbar + std::min(blocks * 262144.0, cap_w / 4) <= cap_w && pred * 1.09 + std::min(blocks * 8192.0, cap_n / 4) <= cap_n;
}

// One unit of progress of the automatic level scheme (no level numbers, no sizes from the caller): an ordinary BFS level while the
// next one is predicted to fit the record buffers (*what = 1: a = that level), otherwise one pass of the deep search — the next level
// inserted into the seen-set only, the one after it probed (*what = 2: a = the inserted level, b = the probed one, b->level == 0 when the
// pass probed nothing).  a->n_new == 0: the search is exhausted.
int32_t vsrmc_checker_advance(vsrmc_checker* c, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what) {
  if (!c || !a || !b || !what) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(b, 0, sizeof(*b));
  b->viol_fp = b->viol_index = ~(u64)0;
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: vsrmc_shard_loop_advance");
  if (!c->deep && (c->opt.exact_ties || next_level_fits(c))) {
    *what = 1;
    return vsrmc_checker_step(c, a);
  }
  *what = 2;
  return vsrmc_checker_deepen(c, a, b);
}

// ≙ ModelChecker.run: stop_reason 0 = exhausted, 1 = invariant violated (*last = the level it was found in; a probed level: see
// vsrmc_checker_probe_trace), 2 = max_depth, 3 = max_seconds, 4 = the seen-set is 85 % full (the search is incomplete: depth reached
// = last->level)
int32_t vsrmc_check(vsrmc_checker* c, int32_t max_depth, double max_seconds, int32_t* stop_reason, vsrmc_level_info* last) {
  if (!c || !stop_reason || !last) return fail(VSRMC_E_ARG, "NULL argument");
  const double t0 = now_s();
  std::memset(last, 0, sizeof(*last));
  last->level = c->level;
  last->distinct = c->distinct;
  vsrmc_level_info a, b;
  while (true) {
    if (max_depth > 0 && c->level + c->deep >= max_depth) { *stop_reason = 2; return 0; }
    if (max_seconds > 0 && now_s() - t0 > max_seconds) { *stop_reason = 3; return 0; }
    if ((double)(c->deep ? c->deep_distinct : c->distinct) > 0.85 * (double)(c->tmask + 1)) { *stop_reason = 4; return 0; }
    int32_t what = 0;
    int rc = vsrmc_checker_advance(c, &a, &b, &what);
    if (rc) return rc;
    *last = a;
    if (a.viol_mask) { *stop_reason = 1; return 0; }
    if (a.n_new == 0) { *stop_reason = 0; return 0; }
    if (what == 2 && b.level && b.viol_mask) { *last = b; *stop_reason = 1; return 0; }
  }
}

}  // extern "C"

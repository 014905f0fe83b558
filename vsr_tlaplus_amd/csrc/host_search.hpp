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
  c->probe_viol.clear();
  c->probe_viol_key.clear();
  c->probe_viol_level = 0;
  int rc = phase_expand(c, nullptr, MODE_PROBE);
  u64 rechecked = 0;
  if (!rc && c->h.limit_unchecked) {                            // (see LevelCtl::limit_unchecked: the pass once more with every action applied)
    rechecked = c->h.limit_unchecked;
    c->probe_all_actions = true;
    rc = phase_expand(c, nullptr, MODE_PROBE);
    c->probe_all_actions = false;
  }
  if (rc) return rc;
  std::memset(info, 0, sizeof(*info));
  info->limit_rechecked = rechecked;
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
    if (c->opt.world <= 1) {                                    // every violating successor the kernel listed is a state of the probed level
      const u64 n = std::min<u64>(c->h.n_pending, std::min<u64>(c->opt.pending_entries, (u64)1 << 20));
      std::vector<u64> list(2 * n);
      if (n) HIPCHK(hipMemcpy(list.data(), c->pending, 16 * n, hipMemcpyDeviceToHost));
      std::vector<std::pair<u64, u64>> pr(n);
      for (u64 i = 0; i < n; i++) pr[i] = std::make_pair(list[2 * i], list[2 * i + 1]);
      std::sort(pr.begin(), pr.end());                            // by fingerprint, then key: the first entry of a fingerprint carries its smallest key
      for (u64 i = 0; i < n; i++)
        if (i == 0 || pr[i].first != pr[i - 1].first) { c->probe_viol.push_back(pr[i].first); c->probe_viol_key.push_back(pr[i].second); }
      c->probe_viol_level = c->level + 1;
    }
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

int32_t vsrmc_checker_probe_violators(vsrmc_checker* c, uint64_t* fps, uint64_t cap, uint64_t* n) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  *n = (u64)c->probe_viol.size();
  if (!fps) return 0;                                           // fps == NULL: only the number is asked for
  if (cap < *n) return fail(VSRMC_E_ARG, "buffer too small");
  for (u64 i = 0; i < *n; i++) fps[i] = c->probe_viol[(size_t)i];
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

// TLCTrace.getTrace for a violating state of the caller's choice: the probe collects EVERY violating successor (vsrmc_checker_probe_violators) and reports the
// one with the smallest fingerprint; the reference's own counter-example (state_transfer_violation_trace.txt:555-577) ends in another one of them.  The path:
// Init .. the violator's parent — the state of the level below whose fingerprint ends in the 45 bits the smallest key among the violator's copies carries,
// the same rule that names every state's predecessor in the seen-set — then the violator itself.
int32_t vsrmc_checker_trace_to_violator(vsrmc_checker* c, uint64_t fp, uint64_t* words, uint64_t cap_words, uint64_t* off, int32_t* actions,
                                        uint64_t cap_states, uint64_t* n_states) {
  if (!c || !words || !off || !actions || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  const auto it = std::lower_bound(c->probe_viol.begin(), c->probe_viol.end(), (u64)fp);
  if (it == c->probe_viol.end() || *it != fp || c->probe_viol_key.size() != c->probe_viol.size() || c->probe_viol_level < 2)
    return fail(VSRMC_E_STATE, "not a violating state of the last probed level (vsrmc_checker_probe_violators lists them)");
  const u64 key = c->probe_viol_key[(size_t)(it - c->probe_viol.begin())];
  HIPCHK(hipSetDevice(c->opt.device));
  int found = 0;
  u64 pfp = 0, pmeta = 0;
  int rc = table_lookup(c, meta_pfp(key), c->probe_viol_level - 1, 1, &found, &pfp, &pmeta);
  if (rc) return rc;
  if (found > 1) return fail(VSRMC_E_STATE, "ambiguous predecessor pointer: several states of the parent's level share the 45 fingerprint bits the violating successor keeps of its parent");
  if (!found) return fail(VSRMC_E_STATE, "the parent of this violating state is not in the seen-set");
  std::vector<u64> fps;
  rc = walk_trace(c, pfp, c->probe_viol_level - 1, &fps);
  if (rc) return rc;
  fps.push_back((u64)fp);
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

// ---- no fatal mispredictions (round 5) ---------------------------------------------------------------------------------------------------
// (1) A stored level that ran out of record buffer.  k_expand keeps enumerating, hashing, claiming, counting and checking after the buffers are
// exhausted (LevelCtl::full: only the records written from then on are garbage), so when vsrmc_checker_step comes back with "frontier full" the
// seen-set holds the COMPLETE level L+1 with its final min-merged keys, and the control block holds its exact figures: it is what a MODE_INSERT pass
// (the first pass of the deep search, vsr_deep.hpp) would have left, except for the two fingerprint checksums, which one pass over the seen-set
// supplies (k_table_level_checksum).  The level is adopted as the first seen-set-only level and the search rolls on from the base it already has.
static int adopt_overflowed_level(vsrmc_checker* c, vsrmc_level_info* ins) {
  if (!c->failed || !c->full_recoverable || c->failed_code != ERR_FRONTIER_FULL || c->deep || c->opt.world > 1 || c->opt.exact_ties)
    return fail(VSRMC_E_STATE, "no overflowed level to adopt");
  HIPCHK(hipSetDevice(c->opt.device));
  const LevelCtl h = c->full_h;                                 // the overflowed step's own control block (snapshot: c->h may have been rewritten since)
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 24));
  u64 sums[3] = {0, 0, 0};
  bool ok = hipMemsetAsync(d, 0, 24, c->stream) == hipSuccess;
  hipLaunchKernelGGL(k_table_level_checksum, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1, c->level + 1, d);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
       hipMemcpy(sums, d, 24, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return fail(VSRMC_E_HIP, "k_table_level_checksum failed");
  if (sums[2] != h.n_written)                                   // every claim wrote (or tried to write) one record: the two counts are one number
    return fail(VSRMC_E_STATE, "adopting an overflowed level: the seen-set holds " + std::to_string((unsigned long long)sums[2]) +
                               " states of the level, the kernel claimed " + std::to_string((unsigned long long)h.n_written));
  c->failed = 0;
  c->failed_code = 0;
  c->full_recoverable = false;
  std::memset(ins, 0, sizeof(*ins));
  ins->level = c->level + (sums[2] ? 1 : 0);
  ins->frontier = c->n_valid;
  ins->n_new = sums[2];
  ins->generated = h.generated;
  ins->deadlocks = h.deadlocks;
  ins->probes = h.probes;
  ins->max_bag = h.max_bag;
  ins->fp_xor = sums[0];
  ins->fp_sum = sums[1];
  ins->pending = 1;                                             // k_expand launches of this "pass"
  ins->expand_ms = c->full_ms;
  ins->seconds = now_s() - c->full_t0;
  for (int a = 0; a < 16; a++) ins->act_generated[a] = h.act_generated[a];
  ins->viol_fp = ins->viol_index = ~(u64)0;
  c->probe_fp = 0; c->probe_level = 0; c->probe_extra_fp = 0;
  if (sums[2] == 0) {                                           // (cannot overflow with nothing written; kept for symmetry with deep_first_pass)
    ins->distinct = c->distinct;
    ins->total_generated = c->total_generated + h.generated;
    return 0;
  }
  c->deep = 1;
  c->deep_lv.assign(1, DeepLevelRec());
  if (c->claim_bits) { (void)hipFree(c->claim_bits); c->claim_bits = nullptr; c->claim_w = c->claim_parents = 0; }   // (an ordinary level leaves no claim bitmap)
  DeepLevelRec& dl = c->deep_lv[0];
  dl.n_new = dl.n_local = sums[2]; dl.generated = h.generated; dl.max_bag = h.max_bag; dl.frontier = c->n_valid;
  c->deep_g = std::max<u64>(c->deep_g, (h.generated + c->n_valid - 1) / std::max<u64>(1, c->n_valid) + 1);
  c->deep_distinct = c->distinct + sums[2];
  c->deep_generated = c->total_generated + h.generated;
  c->deep_regen_done = true;                                    // the idle buffers (and lvl_fp) hold the overflowed level's debris: states are addressed by fingerprint from here on
  ins->distinct = c->deep_distinct;
  ins->total_generated = c->deep_generated;
  if (h.viol_fp != ~(u64)0) {
    ins->viol_fp = h.viol_fp;
    ins->viol_mask = (int32_t)h.viol_mask;
    c->probe_fp = h.viol_fp;                                    // the violator is in the seen-set: the counter-example is walked from it
    c->probe_level = c->level + 1;
  }
  return 0;
}

// (2) A seen-set that fills up while device memory is free: re-hashed into a table of twice the slots (the probe sequence depends on the size,
// the content does not; predecessor pointers are fingerprint bits, not slot numbers).  0 = grown, 1 = cannot (no memory, or 2^36 slots reached).
static int table_grow(vsrmc_checker* c) {
  if (c->opt.table_log2 >= 36) return 1;
  HIPCHK(hipSetDevice(c->opt.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const u64 old_slots = c->tmask + 1, new_slots = old_slots * 2;
  deep_free_scratch(c);                                         // the deep search's scratch buffers are empty between two passes: the next pass plans them anew
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  if ((double)free_b < (double)new_slots * sizeof(Slot) + 512e6) return 1;
  Slot* nt = nullptr;
  u32* d_err = nullptr;
  if (table_alloc(&nt, new_slots) != hipSuccess) { (void)hipGetLastError(); return 1; }
  if (hipMalloc((void**)&d_err, 4) != hipSuccess || hipMemsetAsync(d_err, 0, 4, c->stream) != hipSuccess) {
    (void)hipFree(nt);
    if (d_err) (void)hipFree(d_err);
    return fail(VSRMC_E_HIP, "table_grow: hipMalloc");
  }
  hipLaunchKernelGGL(k_table_init, dim3(4096), dim3(256), 0, c->stream, nt, new_slots);
  hipLaunchKernelGGL(k_table_rehash, dim3(4096), dim3(256), 0, c->stream, c->table, old_slots, nt, new_slots - 1, d_err);
  u32 terr = 0;
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                  hipMemcpy(&terr, d_err, 4, hipMemcpyDeviceToHost) == hipSuccess && terr == 0;
  (void)hipFree(d_err);
  if (!ok) { (void)hipFree(nt); return fail(VSRMC_E_HIP, "table_grow: re-hash failed"); }
  (void)hipFree(c->table);
  c->table = nt;
  c->tmask = new_slots - 1;
  c->opt.table_log2 += 1;
  return 0;
}

// how many new states the next unit of progress may insert, at most: the newest level times the successors a state generates — bounded by the growth the
// run has shown once it is past its first levels (level sizes of these models grow by a falling factor: next_level_fits)
static double predicted_new(const vsrmc_checker* c) {
  const u64 n = c->deep ? c->deep_lv.back().n_new : c->n_valid;
  const u64 g = c->deep ? c->deep_g : std::max<u64>(2, c->g_last);
  double pred = (double)n * (double)g;
  double grow = 0;
  if (c->deep >= 2) grow = (double)c->deep_lv.back().n_new / (double)std::max<u64>(1, c->deep_lv[c->deep_lv.size() - 2].n_new);
  else if (c->deep == 1) grow = (double)c->deep_lv.back().n_new / (double)std::max<u64>(1, c->n_valid);
  else if (c->hist_new[0]) grow = (double)c->hist_new[1] / (double)c->hist_new[0];
  if (n >= 32768 && grow > 0) pred = std::min(pred, (double)n * grow * 1.05);
  return pred;
}

// The seen-set before the next unit of progress: *state = 0 room enough, 1 = it was re-hashed into a larger table (vsrmc_checker_options shows the new
// table_log2), 2 = more than 85 % full and it cannot grow — the search is incomplete at the depth reached.  vsrmc_check, the CLI and
// ModelChecker.run ask before every vsrmc_checker_advance.
int32_t vsrmc_checker_room(vsrmc_checker* c, int32_t* state) {
  if (!c || !state) return fail(VSRMC_E_ARG, "NULL argument");
  *state = 0;
  const double slots = (double)(c->tmask + 1);
  const double have = (double)(c->deep ? c->deep_distinct : c->distinct);   // (a sharded checker: this rank's share of both)
  const bool over = have > 0.85 * slots;
  if (!over && have + predicted_new(c) <= 0.85 * slots) return 0;
  if (c->opt.world > 1) { *state = over ? 2 : 0; return 0; }    // ranks grow nothing on their own: the level loop stops the run together
  while (true) {
    const int g = table_grow(c);
    if (g < 0) return g;
    if (g == 1) { *state = over ? 2 : (*state ? 1 : 0); return 0; }   // cannot grow: full only once the load itself is past 0.85
    *state = 1;
    const double s2 = (double)(c->tmask + 1);
    if (have + predicted_new(c) <= 0.85 * s2) return 0;
  }
}

// One unit of progress of the automatic level scheme (no level numbers, no sizes from the caller): an ordinary BFS level while the
// next one is predicted to fit the record buffers (*what = 1: a = that level), otherwise one pass of the deep search — the next level
// inserted into the seen-set only, the one after it probed (*what = 2: a = the inserted level, b = the probed one, b->level == 0 when the
// pass probed nothing; *what = 3: no new level — the deep search was RE-BASED, a = the level that is the stored base now, vsr_deep.hpp).
// a->n_new == 0: the search is exhausted.
int32_t vsrmc_checker_advance(vsrmc_checker* c, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what) {
  if (!c || !a || !b || !what) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(b, 0, sizeof(*b));
  b->viol_fp = b->viol_index = ~(u64)0;
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: vsrmc_shard_loop_advance");
  if (c->failed && c->full_recoverable) {                       // a vsrmc_checker_step the caller made itself ran out of record buffer: go on from the seen-set
    *what = 2;
    return adopt_overflowed_level(c, a);
  }
  if (!c->deep && (c->opt.exact_ties || next_level_fits(c))) {
    *what = 1;
    int rc = vsrmc_checker_step(c, a);
    if (rc && c->failed && c->full_recoverable) {               // the prediction was wrong: the level is complete in the seen-set, its records are not (see above)
      *what = 2;
      rc = adopt_overflowed_level(c, a);
    }
    return rc;
  }
  if (rebase_pays(c)) {                                          // the newest seen-set-only level fits the idle record buffer: make it the stored base
    const int rc = deep_rebase(c, a);
    if (rc <= 0) { *what = 3; return rc; }                       // (1 = no memory for the scratch buffers: go on without re-basing)
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
    int32_t room = 0;
    int rc = vsrmc_checker_room(c, &room);                       // grows the seen-set while the device has memory for a table of twice the size
    if (rc) return rc;
    if (room == 2) { *stop_reason = 4; return 0; }
    int32_t what = 0;
    rc = vsrmc_checker_advance(c, &a, &b, &what);
    if (rc) return rc;
    if (what == 3) continue;                                     // re-based: no new level, the next unit of progress is an ordinary level again
    *last = a;
    if (a.viol_mask) { *stop_reason = 1; return 0; }
    if (a.n_new == 0) { *stop_reason = 0; return 0; }
    if (what == 2 && b.level && b.viol_mask) { *last = b; *stop_reason = 1; return 0; }
  }
}

}  // extern "C"

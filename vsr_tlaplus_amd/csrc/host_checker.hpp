// host_checker.hpp — the checker handle (≙ ModelChecker + Worker.run): buffers, kernel selection, the phases of a BFS level, passes over arbitrary sources, trace walks (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

// ---------------------------------------------------------------------------------------------------------------
// checker
// ---------------------------------------------------------------------------------------------------------------
struct PassDst {            // where a pass writes (records, refs, fingerprints)
  u64* words = nullptr;
  u64 words_cap = 0;
  u64* off = nullptr;
  u64* fp = nullptr;
  u64 cap = 0;
  bool own_words = true;    // scratch buffers of the deep search: false = `words` is a piece of the idle record buffer (nothing to free)
  bool own_index = true;    //                                      false = `off` / `fp` are the idle frontier's own arrays
};
struct DeepLevelRec { u64 n_new = 0, n_local = 0, generated = 0, max_bag = 0, frontier = 0; };   // n_local: this rank's share (unsharded: all)   // a level that exists in the seen-set only (vsr_deep.hpp)
struct vsrmc_checker {
  vsrmc_model model;
  vsrmc_options opt;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  Slot* table = nullptr;
  u64 tmask = 0;
  u64* words[2] = {nullptr, nullptr};
  u64* off[2] = {nullptr, nullptr};
  u64* lvl_fp = nullptr;
  u64* pending = nullptr;
  LevelCtl* ctl = nullptr;
  u64* d_find = nullptr;
  int cur = 0;
  int level = 0;
  u64 n_frontier = 0;
  u64 distinct = 0, total_generated = 0;
  int num_cus = 256;
  int lds_stride = 65;
  int failed = 0;
  // TLCTrace: there is no separate log — a state's slot in the seen-set names its parent:
  // 45 bits of its parent's fingerprint (meta word, vsr_model.hpp); traces are walked through the table (k_trace_walk)
  // state of the level in flight (between the phases)
  LevelCtl h;
  double t_level0 = 0, expand_ms = 0, materialize_ms = 0;
  u64 nx_n = 0, nx_w = 0;
  u64 n_valid = 1;                       // states in the newest level (n_frontier is its index range, holes included)
  u64 cur_w = 0;                         // words of the current frontier buffer in use (chunk slack included)
  u64* rslot = nullptr;                  // sharded: slot of every received candidate
  u64 rslot_cap = 0;
  u64* filter = nullptr;                 // sharded single-pass levels: this rank's sent-filter (vsr_kernels.hpp, k_expand)
  u64 fmask = 0;
  u64* cand_idx = nullptr;               // ... and where each announced candidate was written (world x cand_cap)
  u64 cand_idx_cap = 0;
  bool level_fused = false;              // the level in flight is a single-pass level
  int failed_code = 0;                   // device ERR_* that stopped the search (failed == 1)
  // the single-pass kernel of this model (a specialised instantiation when there is one) and its launch shape
  void* fused_kernel = nullptr;
  void* plain_kernel = nullptr;          // the same without modes / sharding, when the configuration has one (ordinary unsharded levels)
  u64* redo_buf[2] = {nullptr, nullptr}; // the lists of tiles that overflowed the LDS work list of an ordinary launch (LevelCtl::redo_out), 65536 entries each
  void* plain5_kernel = nullptr;         // ... compiled for FIVE resident blocks per CU (96 VGPRs), when the configuration has one (BASELINE configs[1]): fused_shape picks it
                                         // for the launches whose LDS fits five times
  void* modes_kernel = nullptr;          // the same without sharding (expand_pass: the passes of vsrmc_checker_probe / _probe2 / _probe3), or null
  void* regen_bits_kernel = nullptr, *insert_kernel = nullptr;   // ... with ONE mode compiled in (k_expand: PLAIN == 3 / 4), or null
  void* probe_kernel = nullptr;          // ... the probe pass alone, five blocks per CU, failing successors resolved afterwards (PLAIN == 6; k_probe_resolve), or null
  u64 cur_max_bag = 0;                   // largest bag among the records of the newest level (LDS slot size of the next launch)
  bool bag_known = true;                 // false after a checkpoint was loaded or records arrived from other ranks: use the capacity
  // vsrmc_checker_probe / _probe2: where the reported violator's counter-example is walked from — the fingerprint of the deepest
  // state of the path that is IN the seen-set, its level, and the fingerprint of the one probed state beyond it (0: none)
  u64 probe_fp = 0, probe_extra_fp = 0;
  int probe_level = 0;
  std::vector<u64> probe_viol;           // the distinct violating STATES of the last probed level (fingerprints, ascending): vsrmc_checker_probe_violators
  std::vector<u64> probe_viol_key;       // ... and, beside each, the smallest key among its copies (it names the state's parent): vsrmc_checker_trace_to_violator
  int probe_viol_level = 0;              // the probed level they belong to
  int host_frontier = 0;                 // bit b: record buffer b lives in pinned host memory (zero-copy over PCIe)
  bool saw_violation = false;            // a committed level held a violating state (the caller went on): probe passes apply every action
  u64 extra_launches = 0;                // k_expand launches of the last pass beyond its first one: the lists of overflowed tiles taken again (redo_overflowed_tiles), a probe pass run again
  bool probe_all_actions = false;        // ... and so does the second run of a probe pass whose first run left instances unapplied beside a representation limit
  // levels beyond the record buffers (vsr_deep.hpp): `deep` levels above `level` are complete in the seen-set and have no frontier
  int deep = 0;
  std::vector<DeepLevelRec> deep_lv;     // [i] = level + 1 + i
  u64 deep_g = 2;                        // successors generated per expanded state, rounded up, the largest any level showed (worst-case slice sizes)
  u64 deep_distinct = 0, deep_generated = 0;
  bool deep_regen_done = false;          // a descent has set taken bits in the levels beyond the base: cleared before the next one
  bool rebase_off = false;               // a re-basing descent could not get its scratch buffers: not tried again (vsr_deep.hpp: deep_rebase)
  std::vector<PassDst> scratch;          // scratch buffers 1 .. of the descent (vsr_deep.hpp: deep_plan_scratch; planned at the start of a pass)
  PassDst scratch0;                      // buffer 0 of the descent when the plan CARVES the idle record buffer (words == nullptr: buffer 0 is that whole buffer)
  bool scratch_carved = false;
  int scratch_buf = -1;                  // which record buffer (0 / 1) the plan's carved pieces point into: a plan made while the other buffer was idle is stale
  u64 scratch_front = 0;                 // a re-basing plan: the first words of the idle record buffer are the descent's destination, not scratch
  // sharded deep search: the generator-side winner set (vsr_kernels.hpp: WSet) — which deep-level states THIS rank's candidates inserted
  WSet h_wset = {nullptr, nullptr, 0};
  WSet* d_wset = nullptr;                // the same three words on the device (what the kernels are handed); nullptr until the first deep pass
  u32 wepoch = 0;                        // the number of the descent in flight
  bool wset_used = false, wset_dirty = false;   // this search has put states into the set / a reset came after that: emptied before the next use
  // unsharded deep search: which (parent, ordinal) instance of the stored base inserted each state of the first seen-set-only level (vsr_deep.hpp)
  u32* claim_bits = nullptr;
  u64 claim_w = 0, claim_parents = 0;    // words per parent; parents covered (= the base level's index range)
  LevelCtl full_h;                       // ... the control block of THAT step (c->h is rewritten by every later pass, lookup or trace helper), its kernel time and start
  double full_ms = 0, full_t0 = 0;
  bool full_recoverable = false;         // the last vsrmc_checker_step stopped with "frontier full" and lost nothing but records: the level is complete in the
                                         // seen-set and vsrmc_checker_advance keeps it as a seen-set-only level (host_search.hpp: adopt_overflowed_level)
  u64 hist_new[2] = {0, 0};              // new states of the last two levels (growth estimate of vsrmc_checker_advance)
  u64 g_last = 16, cur_rec_w = 0;        // successors generated per expanded state of the last level (rounded up, + 1); words of the newest level's records
  u64 words_cap(int b) const { return (b == 1 && opt.frontier_words_b) ? opt.frontier_words_b : opt.frontier_words; }
};

namespace {
typedef void (*ExpandKernel)(Model, const u64*, const u64*, u64, int, int, Slot*, u64, u64*, u64, LevelCtl*, int, int, u64*, u64, u32, u64*,
                             u64, u64*, u64, u64*, u32, u32, int, u32, u64*, u64, u64*, u32, int, u64, const WSet*, u32);
// k_expand<true, SPEC>: the configurations of BASELINE.json (and their small neighbours used by the tests) have their own
// instantiation with the model constants folded in; anything else runs the generic one.
ExpandKernel exact_kernel_for(const Model& M) {               // two-kernel levels: k_expand<false, SPEC>
  if (M.model_id == 1) return k_expand<false, 1000>;
  if (M.model_id == 2) return k_expand<false, 2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_expand<false, 211>;
    case 312: return k_expand<false, 312>;
    case 313: return k_expand<false, 313>;
    case 512: return k_expand<false, 512>;
    default: return k_expand<false, 0>;
  }
}
typedef void (*MaterializeKernel)(Model, const u64*, const u64*, const u64*, u64, Slot*, u64*, u64, u64*, u64, u64*, LevelCtl*,
                                  const uint8_t*, u64*, u64*, int, u32, u32, int, const u64*);
MaterializeKernel materialize_kernel_for(const Model& M) {
  if (M.model_id == 1) return k_materialize<1000>;
  if (M.model_id == 2) return k_materialize<2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_materialize<211>;
    case 312: return k_materialize<312>;
    case 313: return k_materialize<313>;
    case 512: return k_materialize<512>;
    default: return k_materialize<0>;
  }
}
ExpandKernel plain_kernel_for(const Model& M) {               // unsharded ordinary levels: modes and sharding compiled out
  if (M.model_id == 1) return (M.R == 3 && M.n == 2) ? k_expand<true, 1302, true> : nullptr;   // the shipped VR_STATE_TRANSFER.cfg
  if (M.model_id == 2) return (M.R == 3 && M.n == 2) ? k_expand<true, 2302, true> : nullptr;   // the shipped VR_APP_STATE.cfg
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return k_expand<true, 312, true>;
    case 313: return k_expand<true, 313, true>;
    case 512: return k_expand<true, 512, true>;
    default: return nullptr;
  }
}
ExpandKernel plain5_kernel_for(const Model& M) {              // the ordinary level at five blocks per CU: pays where the kernel fits 96 registers (two permutations: 107 -> 7 spilled)
  if (M.model_id != 0 || std::getenv("VSRMC_NO_OCC5")) return nullptr;
  return (M.R * 100 + M.C * 10 + M.n) == 312 ? k_expand<true, 312, 5> : nullptr;
}
ExpandKernel modes_kernel_for(const Model& M) {               // unsharded passes with a mode (probe / virtual / regenerated / streamed levels)
  if (M.model_id != 0) return nullptr;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return k_expand<true, 312, 2>;
    case 313: return k_expand<true, 313, 2>;
    case 512: return k_expand<true, 512, 2>;
    default: return nullptr;
  }
}
// one mode compiled in (k_expand: PLAIN == 3 / 4): which = MODE_REGEN (by the claim bitmap) or MODE_INSERT
ExpandKernel one_mode_kernel_for(const Model& M, int which) {
  if (M.model_id != 0) return nullptr;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return which == MODE_REGEN ? k_expand<true, 312, 3> : k_expand<true, 312, 4>;
    case 313: return which == MODE_REGEN ? k_expand<true, 313, 3> : k_expand<true, 313, 4>;
    case 512: return which == MODE_REGEN ? k_expand<true, 512, 3> : k_expand<true, 512, 4>;
    default: return nullptr;
  }
}
ExpandKernel probe_kernel_for(const Model& M) {
  if (M.model_id != 0 || std::getenv("VSRMC_NO_PROBE_KERNEL")) return nullptr;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return k_expand<true, 312, 6>;
    case 313: return k_expand<true, 313, 6>;
    case 512: return k_expand<true, 512, 6>;
    default: return nullptr;
  }
}
ExpandKernel fused_kernel_for(const Model& M) {
  if (M.model_id == 1) return (M.R == 3 && M.n == 2) ? k_expand<true, 1302> : k_expand<true, 1000>;
  if (M.model_id == 2) return (M.R == 3 && M.n == 2) ? k_expand<true, 2302> : k_expand<true, 2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_expand<true, 211>;
    case 212: return k_expand<true, 212>;
    case 311: return k_expand<true, 311>;
    case 312: return k_expand<true, 312>;
    case 313: return k_expand<true, 313>;
    case 323: return k_expand<true, 323>;
    case 412: return k_expand<true, 412>;
    case 512: return k_expand<true, 512>;
    default: return k_expand<true, 0>;
  }
}
// Launch shape of the single-pass kernel for one launch.  The LDS slot of a record only has to hold the longest record of the
// level that is being expanded (stride = fixed words + its largest bag, made odd: conflict-free columns), not the format's
// worst case, so deep levels of small bags leave room for more resident blocks.  64-record tiles when that gives at least
// three blocks per CU (registers and LDS, asked from the runtime), else 128-record tiles (R <= 3) at two.
#ifndef VSR_CCAP64          // work-list entries of a 64-record tile, R <= 3 (24 per record; an overflow is ERR_FRONTIER_FULL, never silent)
#define VSR_CCAP64 1536
#endif
struct FusedShape {
  int blk;
  int tile;
  u32 ccap;
  int stride;
  size_t lds;
  unsigned blocks_per_cu;
  const void* kernel;       // the instantiation this shape is for (plain launches: the four- or the five-block one)
};
FusedShape fused_shape(vsrmc_checker* c, u64 max_bag_of_source, bool plain = false) {
  const Model& M = c->model.M;
  FusedShape f;
  f.stride = (int)std::min<u64>((u64)c->lds_stride, (u64)((M.fixed + (int)std::min<u64>(max_bag_of_source, 255)) | 1));
  const void* kernel = (plain && c->plain_kernel) ? c->plain_kernel : c->fused_kernel;
  const int blk = VSR_BLOCK;                                   // (one-wave and 512-thread blocks were measured and rejected: DESIGN.md §5, round 3)
  auto occupancy = [&](int tile, u32 ccap, size_t* lds) {
    *lds = (size_t)tile * f.stride * 8 + 2 * (size_t)ccap * 4;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, blk, *lds) != hipSuccess) nb = 0;
    return nb;
  };
  f.blk = blk;
  size_t lds64 = 0, lds128 = 0;
  u32 ccap64 = M.R <= 3 ? (u32)VSR_CCAP64 : (u32)VSR_CAND_CAP;            // work-list entries per tile (24 resp. 32 per record)
  if (const char* e = std::getenv("VSRMC_CCAP"))                           // tests: a work list so short that tiles overflow it and are taken again in pieces (k_expand: s_redo_*)
    ccap64 = (u32)std::max(256, std::min(2048, std::atoi(e))) & ~127u;
  const int occ64 = occupancy(64, ccap64, &lds64);
  // 128-record tiles: only the instantiations that are generic in the constants (compiled for two blocks per CU) ever have fewer than three 64-record
  // tiles resident at R <= 3 — the specialised ones size their per-record LDS arrays for 64 (vsr_kernels.hpp: TILE_MAX)
  const bool generic = kernel == (const void*)(ExpandKernel)k_expand<true, 0> || kernel == (const void*)(ExpandKernel)k_expand<true, 1000> ||
                       kernel == (const void*)(ExpandKernel)k_expand<true, 2000>;
  const int occ128 = (M.R <= 3 && generic) ? occupancy(128, 1536u, &lds128) : 0;
  if (M.R <= 3 && occ64 < 3 && occ128 >= 1) {
    f.tile = 128; f.ccap = 1536u; f.lds = lds128; f.blocks_per_cu = (unsigned)std::min(occ128, 2);
  } else {
    f.tile = 64; f.ccap = ccap64; f.lds = lds64; f.blocks_per_cu = (unsigned)std::max(1, std::min(occ64, VSR_OCC + 1));   // (what the instantiation's registers and this LDS allow)
  }
#if VSR_TILE128
  if (plain && M.R <= 3) {                                       // EXPERIMENT: 128-record tiles, a work list of 1024 (8 per record; overflow goes to the host's list), three blocks per CU
    size_t l128 = 0;
    const int o128 = occupancy(128, 1024u, &l128);
    if (o128 >= 3) { f.tile = 128; f.ccap = 1024u; f.lds = l128; f.blocks_per_cu = 3; f.kernel = kernel; return f; }
  }
#endif
  f.kernel = kernel;
  // FIVE blocks per CU: the configuration has an ordinary-level instantiation compiled for 96 registers (plain5_kernel), and this launch's tile plus a
  // work list of 896 entries (14 instances per record; the mean is 5 — a tile with more is written down and launched again in halves:
  // redo_overflowed_tiles) fits the LDS five times.  The runtime's occupancy query is necessary, not sufficient: 32.3 KB per block (1024 entries) runs as FOUR
  // resident blocks although the query says five — 31.3 KB (896) is the largest size measured at five (VSRMC_LDS5 / VSRMC_CCAP5: the sweep of DESIGN.md §8.5:
  // config 2 k_expand 132.7 / 132.9 / 132.5 / 131.2 / 141.2 ms at 512 / 640 / 768 / 896 / 1024 entries, 137.1 with the four-block instantiation).
  if (plain && c->plain5_kernel && M.R <= 3 && f.tile == 64) {
    static const u32 cc5 = std::getenv("VSRMC_CCAP5") ? (u32)std::max(256, std::atoi(std::getenv("VSRMC_CCAP5"))) & ~127u : 896u;
    static const size_t lds5 = std::getenv("VSRMC_LDS5") ? (size_t)std::atoll(std::getenv("VSRMC_LDS5")) : (size_t)31744;
    hipFuncAttributes at;
    size_t dyn = (size_t)64 * f.stride * 8 + 2 * (size_t)cc5 * 4;
    int nb5 = 0;
    if (hipFuncGetAttributes(&at, c->plain5_kernel) == hipSuccess && dyn + at.sharedSizeBytes <= lds5 &&
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb5, c->plain5_kernel, blk, dyn) == hipSuccess && nb5 >= VSR_OCC + 1) {
      f.kernel = c->plain5_kernel; f.ccap = cc5; f.lds = dyn; f.blocks_per_cu = (unsigned)(VSR_OCC + 1);
    }
  }
  if (const char* e = std::getenv("VSRMC_MAX_BPC"))            // diagnostic: fewer resident blocks per CU (occupancy sweeps)
    f.blocks_per_cu = (unsigned)std::max(1, std::min<int>((int)f.blocks_per_cu, std::atoi(e)));
  return f;
}

// The seen-set's memory.  EXPERIMENT KNOB (environment, read per allocation): VSRMC_TABLE_MEM=uncached | finegrained allocates it with
// hipExtMallocWithFlags(hipDeviceMallocUncached | hipDeviceMallocFinegrained) — every probe misses the caches anyway (the table is 100 x the L2 + Infinity
// Cache), and a cached 16-byte probe drags a 128-byte line through the fabric; unset = hipMalloc (the product's default).
hipError_t table_alloc(Slot** out, u64 slots) {
  const char* e = std::getenv("VSRMC_TABLE_MEM");
  if (e && std::strcmp(e, "uncached") == 0) return hipExtMallocWithFlags((void**)out, slots * sizeof(Slot), hipDeviceMallocUncached);
  if (e && std::strcmp(e, "finegrained") == 0) return hipExtMallocWithFlags((void**)out, slots * sizeof(Slot), hipDeviceMallocFinegrained);
  return hipMalloc((void**)out, slots * sizeof(Slot));
}

// the winner set of a sharded deep search: allocated at the first pass beyond the record buffers, half as many slots as the seen-set (the memory
// autosize_options set aside for it), emptied whenever the search starts over
int wset_ensure(vsrmc_checker* c) {
  if (c->opt.world <= 1) return 0;                              // one rank owns and generates everything: the seen-set's own taken bits do (vsr_deep.hpp)
  if (c->d_wset) {
    if (c->wset_dirty) {                                         // a search that started over (vsrmc_checker_reset): the set is emptied, the allocation kept
      HIPCHK(hipMemsetAsync(c->h_wset.fp, 0, (c->h_wset.mask + 1) * 8, c->stream));
      HIPCHK(hipMemsetAsync(c->h_wset.epoch, 0, (c->h_wset.mask + 1) * 4, c->stream));
      c->wset_dirty = false;
      c->wepoch = 0;
    }
    return 0;
  }
  u64 slots = std::max<u64>((u64)1 << 12, (c->tmask + 1) / 2);
  if (const char* e = std::getenv("VSRMC_WSET_LOG2")) slots = (u64)1 << std::max(8, std::min(36, std::atoi(e)));   // (tests: a small set that has to grow, wset_grow)
  while (true) {
    hipError_t e = hipMalloc((void**)&c->h_wset.fp, slots * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->h_wset.epoch, slots * 4);
    if (e == hipSuccess) break;
    (void)hipGetLastError();
    if (c->h_wset.fp) (void)hipFree(c->h_wset.fp);
    c->h_wset.fp = nullptr;
    if (slots <= ((u64)1 << 16)) return fail(VSRMC_E_HIP, "hipMalloc of the winner set of the sharded deep search failed");
    slots /= 2;                                                  // (a crowded device: a smaller set; it raises ERR_TABLE_FULL when it runs full)
  }
  c->h_wset.mask = slots - 1;
  HIPCHK(hipMemsetAsync(c->h_wset.fp, 0, slots * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->h_wset.epoch, 0, slots * 4, c->stream));
  HIPCHK(hipMalloc((void**)&c->d_wset, sizeof(WSet)));
  HIPCHK(hipMemcpyAsync(c->d_wset, &c->h_wset, sizeof(WSet), hipMemcpyHostToDevice, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->wepoch = 0;
  c->wset_dirty = false;
  return 0;
}
// states in the winner set: what this rank's candidates inserted into the levels beyond the base (DeepLevelRec::n_local of each)
u64 wset_count(const vsrmc_checker* c) {
  u64 n = 0;
  for (const DeepLevelRec& d : c->deep_lv) n += d.n_local;
  return n;
}
// The set has half the slots of the seen-set shard at first (what autosize_options set aside), but must hold about the rank's share of every level beyond
// the base — nearly the whole shard once the search is deep — and its probes give up after 65 536 steps: it is re-hashed into twice the slots between two
// passes while the device has the memory (round-5 advice; vsrmc_shard_loop_room stops the run cleanly when it cannot grow).  0 = grown, 1 = no memory.
int wset_grow(vsrmc_checker* c) {
  if (!c->d_wset) return 1;
  const u64 old_slots = c->h_wset.mask + 1, slots = old_slots * 2;
  u64* nfp = nullptr;
  u32* nep = nullptr;
  u32* d_err = nullptr;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)slots * 12.0 + 1.0e9 > (double)free_b) return 1;   // (the scratch plan of the next descent needs its share too)
  bool ok = hipMalloc((void**)&nfp, slots * 8) == hipSuccess && hipMalloc((void**)&nep, slots * 4) == hipSuccess && hipMalloc((void**)&d_err, 4) == hipSuccess;
  ok = ok && hipMemsetAsync(nfp, 0, slots * 8, c->stream) == hipSuccess && hipMemsetAsync(nep, 0, slots * 4, c->stream) == hipSuccess &&
       hipMemsetAsync(d_err, 0, 4, c->stream) == hipSuccess;
  u32 err = 0;
  if (ok) {
    hipLaunchKernelGGL(k_wset_rehash, dim3(4096), dim3(256), 0, c->stream, (const u64*)c->h_wset.fp, (const u32*)c->h_wset.epoch, old_slots, nfp, nep, slots - 1, d_err);
    ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(&err, d_err, 4, hipMemcpyDeviceToHost, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess && err == 0;
  }
  if (d_err) (void)hipFree(d_err);
  if (!ok) {
    (void)hipGetLastError();
    if (nfp) (void)hipFree(nfp);
    if (nep) (void)hipFree(nep);
    return 1;
  }
  (void)hipFree(c->h_wset.fp);
  (void)hipFree(c->h_wset.epoch);
  c->h_wset.fp = nfp; c->h_wset.epoch = nep; c->h_wset.mask = slots - 1;
  if (hipMemcpy(c->d_wset, &c->h_wset, sizeof(WSet), hipMemcpyHostToDevice) != hipSuccess) return fail(VSRMC_E_HIP, "winner set: descriptor copy");
  return 0;
}
void wset_free(vsrmc_checker* c) {
  if (c->h_wset.fp) (void)hipFree(c->h_wset.fp);
  if (c->h_wset.epoch) (void)hipFree(c->h_wset.epoch);
  if (c->d_wset) (void)hipFree(c->d_wset);
  c->h_wset.fp = nullptr; c->h_wset.epoch = nullptr; c->h_wset.mask = 0; c->d_wset = nullptr; c->wepoch = 0;
}

void deep_free_scratch(vsrmc_checker* c);                       // (vsr_deep.hpp)

// Put the checker in its initial state (ModelChecker.doInit): empty seen-set, Init in frontier 0 and in the set.
int checker_seed(vsrmc_checker* c) {
  const Model& M = c->model.M;
  // the scratch plan of a deep search points into whichever record buffer was idle when it was made: a search that starts over starts without one
  deep_free_scratch(c);
  c->rebase_off = false;
  c->saw_violation = false;
  c->deep = 0;
  c->deep_lv.clear();
  c->deep_g = 2;
  c->deep_distinct = c->deep_generated = 0;
  c->deep_regen_done = false;
  c->hist_new[0] = c->hist_new[1] = 0;
  c->g_last = 16;
  c->cur_rec_w = 0;
  c->failed = 0;
  c->failed_code = 0;
  c->full_recoverable = false;
  HIPCHK(hipSetDevice(c->opt.device));
  if (c->claim_bits) { (void)hipFree(c->claim_bits); c->claim_bits = nullptr; c->claim_w = c->claim_parents = 0; }
  if (c->d_wset && c->wset_used) c->wset_dirty = true;           // (a fresh search: the next sharded deep pass empties the winner set before it uses it)
  c->wset_used = false;
  hipLaunchKernelGGL(k_table_init, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1);
  HIPCHK(hipGetLastError());
  std::vector<u64> wire, dev(512);
  init_record_wire(M, wire);
  int len = wire_to_device(M, wire.data(), dev.data());
  u64 H[6];
  hash_full_host(M, (const u64*)dev.data(), H);   // pure arithmetic on the constant Init record (same code as the kernels)
  for (int i = 0; i < M.np; i++) dev[M.h0 + i] = H[i];
  u64 zero = (u64)len, init_fp = 0;                            // ref of record 0: offset 0, length len
  u32 init_ak = 0;
  canonical_fp(M, dev[0], &dev[M.h0], &init_fp, &init_ak);
  // sharded: every rank starts with Init (replicated phase: the small early levels are explored by every rank on its own,
  // vsrmc_shard_local_step); vsrmc_shard_partition then leaves each state with its owner
  const bool mine = true;
  if (c->filter) HIPCHK(hipMemsetAsync(c->filter, 0, (c->fmask + 1) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(LevelCtl), c->stream));
  if (mine) {
    HIPCHK(hipMemcpyAsync(c->words[0], dev.data(), len * 8, hipMemcpyDefault, c->stream));   // the buffer may be pinned host memory
    HIPCHK(hipMemcpyAsync(c->off[0], &zero, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, c->stream, M, c->words[0], c->table, c->tmask, c->lvl_fp, c->ctl);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  c->cur = 0;
  c->level = 1;
  c->n_frontier = mine ? 1 : 0;
  c->n_valid = c->n_frontier;
  c->cur_w = (u64)len;
  c->distinct = mine ? 1 : 0;
  c->total_generated = 0;
  c->failed = 0;
  c->cur_max_bag = 0;
  c->bag_known = true;
  c->failed_code = 0;
  c->probe_fp = 0;
  c->probe_level = 0;
  c->probe_extra_fp = 0;
  c->probe_viol.clear();
  c->probe_viol_key.clear();
  c->probe_viol_level = 0;
  return 0;
}
}  // namespace

extern "C" {

void vsrmc_options_default(vsrmc_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->device = 0;
  o->table_log2 = 26;
  o->frontier_words = (uint64_t)1 << 27;
  o->frontier_states = (uint64_t)1 << 22;
  o->pending_entries = (uint64_t)1 << 23;
  o->keep_trace = 1;
  o->trace_entries = 0;
  o->rank = 0;
  o->world = 1;
}

// vsrmc_options with table_log2 == 0 and / or frontier_words == 0: sized from the free memory of the device.  Seen-set: the largest power
// of two of 16-byte slots within 30 % of what is free (1.7e9 states at load 0.4 on an empty MI355X); sharded runs on ONE device (tests)
// take their share.  Records: what is left after the seen-set, the index arrays (24 B per state index), the sent-filter and a reserve
// for the scratch buffers of the deep search (vsr_deep.hpp: 1/4 + 1/16 + .. of one record buffer) and the exchange buffers of a sharded
// run, in two equal buffers — the last two levels differ by the growth factor, but which of the two buffers holds the last one is not
// known in advance; pending list: only the exact scheme needs one worth the name.
static int autosize_options(vsrmc_options* o, const Model& M) {
  if (o->table_log2 != 0 && o->frontier_words != 0) {               // both pinned: only the two small sizes can still be "auto"
    if (o->pending_entries == 0) o->pending_entries = o->exact_ties ? (u64)1 << 24 : (u64)1 << 16;
    if (o->frontier_states == 0) o->frontier_states = std::max<u64>((u64)1 << 16, o->frontier_words / 24);
    return 0;
  }
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipSetDevice(o->device));
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const char* share_env = std::getenv("VSRMC_AUTOSIZE_SHARE");     // several checkers on one device (tests: ranks sharing a GPU): 1 / share each
  const double share = share_env ? std::max(1.0, std::atof(share_env)) : 1.0;
  double avail = ((double)free_b - 4.0e9) / share;                 // runtime, code objects, small allocations (a sharded run: make the
                                                                   // communicator BEFORE the checker, so that its buffers are not counted as free)
  if (avail < 256e6) return fail(VSRMC_E_HIP, "less than 256 MB of free device memory to size the checker from");
  if (o->table_log2 == 0) {
    int lg = 8;
    while (lg < 36 && (double)((u64)1 << (lg + 1)) * 16.0 <= 0.30 * avail) lg++;
    o->table_log2 = lg;
  }
  avail -= (double)((u64)1 << o->table_log2) * 16.0;
  if (o->world > 1 && !o->exact_ties) avail -= (double)((u64)1 << (o->filter_log2 > 0 ? o->filter_log2 : o->table_log2)) * 8.0;
  if (o->world > 1 && !o->exact_ties) avail -= (double)((u64)1 << o->table_log2) * 6.0;    // the winner set of the deep search: half the slots, 12 B each
  if (o->pending_entries == 0) o->pending_entries = o->exact_ties ? (u64)1 << 24 : (u64)1 << 16;
  avail -= (double)o->pending_entries * 24.0;
  if (o->frontier_words == 0) {
    if (o->world > 1) avail *= 0.80;                               // candidate / verdict / rebalancing buffers of the level loop
    // per record word: 8 B in each of two buffers, 1/24 state index (3 arrays of 8 B), a third of one buffer for the scratch buffers
    const double per_word = 2.0 * 8.0 + 8.0 / 3.0 + 24.0 / 24.0;
    const double words = avail / per_word;
    if (words < 4096.0 * (M.fixed + M.max_bag)) return fail(VSRMC_E_HIP, "not enough free device memory for the record buffers");
    o->frontier_words = (u64)words;
    o->frontier_words_b = 0;
    if (o->frontier_states == 0) o->frontier_states = std::max<u64>((u64)1 << 16, o->frontier_words / 24);
  }
  if (o->frontier_states == 0) o->frontier_states = std::max<u64>((u64)1 << 16, o->frontier_words / 24);
  return 0;
}

int32_t vsrmc_checker_create(const vsrmc_model* m, const vsrmc_options* o_in, vsrmc_checker** out) {
  if (!m || !o_in || !out) return fail(VSRMC_E_ARG, "NULL argument");
  vsrmc_options sized = *o_in;
  {
    const int rc0 = autosize_options(&sized, m->M);
    if (rc0) return rc0;
  }
  const vsrmc_options* o = &sized;
  if (o->table_log2 < 8 || o->table_log2 > 36 || o->frontier_states < 1 || o->frontier_words < 256 ||
      o->pending_entries < 4 * (uint64_t)VSR_CAND_CAP)
    return fail(VSRMC_E_ARG, "bad options");
  if (o->world < 1 || o->world > 8 || o->rank < 0 || o->rank >= o->world) return fail(VSRMC_E_ARG, "bad rank / world (1..8 ranks)");
  if (o->frontier_states > ((uint64_t)1 << 40)) return fail(VSRMC_E_ARG, "frontier_states > 2^40");   // origin word: parent index | ordinal << 40
  // a block reserves frontier indices in chunks of at least VSR_CAND_CAP (one tile's successors) and clears the unused tail of its
  // chunk: a frontier smaller than one chunk would be written past its end
  if (o->frontier_states < (uint64_t)VSR_CAND_CAP) return fail(VSRMC_E_ARG, "frontier_states must be at least 2048 (one index chunk)");
  int rc = check_device(o->device);
  if (rc) return rc;
  vsrmc_checker* c = new vsrmc_checker();
  c->model = *m;
  c->opt = *o;
  const Model& M = c->model.M;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, o->device));
  c->num_cus = prop.multiProcessorCount;
  c->lds_stride = (M.fixed + M.max_bag) | 1;
  HIPCHK(hipStreamCreate(&c->stream));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c->ev[i]));
  u64 slots = (u64)1 << o->table_log2;
  c->tmask = slots - 1;
  hipError_t e = table_alloc(&c->table, slots);
  c->host_frontier = o->host_frontier & 3;
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    // host_frontier: the records stay in pinned host memory and the kernels read / write them over PCIe (zero-copy); the
    // refs, fingerprints, trace log and the seen-set stay in HBM.  For state spaces whose frontier outgrows the 288 GB.
    if ((c->host_frontier >> b) & 1) e = hipHostMalloc((void**)&c->words[b], c->words_cap(b) * 8, hipHostMallocMapped | hipHostMallocPortable);
    else e = hipMalloc((void**)&c->words[b], c->words_cap(b) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->off[b], (o->frontier_states + 1) * 8);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&c->lvl_fp, o->frontier_states * 8);
  if (e == hipSuccess && o->world > 1 && !o->exact_ties) {
    const int fl = o->filter_log2 > 0 ? o->filter_log2 : o->table_log2;
    c->fmask = ((u64)1 << fl) - 1;
    e = hipMalloc((void**)&c->filter, (c->fmask + 1) * 8);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&c->pending, o->pending_entries * 24);   // (slot, key, parent index) entries of the exact scheme
  if (e == hipSuccess) e = hipMalloc((void**)&c->ctl, sizeof(LevelCtl));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_find, 8);
  if (e == hipSuccess) e = hipMalloc((void**)&c->redo_buf[0], (size_t)65536 * 8);
  if (e == hipSuccess) e = hipMalloc((void**)&c->redo_buf[1], (size_t)65536 * 8);
  if (e != hipSuccess) {
    vsrmc_checker_destroy(c);
    return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  c->fused_kernel = (void*)fused_kernel_for(M);
  c->modes_kernel = (void*)modes_kernel_for(M);
  c->plain_kernel = (void*)plain_kernel_for(M);
  c->plain5_kernel = (void*)plain5_kernel_for(M);
  if (!std::getenv("VSRMC_NO_MODE_KERNELS")) {                   // (A/B knob: the run-time-switched instantiation for every pass, as in rounds 3-4)
    c->regen_bits_kernel = (void*)one_mode_kernel_for(M, MODE_REGEN);
    c->insert_kernel = (void*)one_mode_kernel_for(M, MODE_INSERT);
    c->probe_kernel = (void*)probe_kernel_for(M);
  }
  rc = checker_seed(c);
  if (rc) { vsrmc_checker_destroy(c); return rc; }
  *out = c;
  return 0;
}

int32_t vsrmc_checker_options(const vsrmc_checker* c, vsrmc_options* out) {
  if (!c || !out) return fail(VSRMC_E_ARG, "NULL argument");
  *out = c->opt;
  return 0;
}

int32_t vsrmc_checker_reset(vsrmc_checker* c) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  return checker_seed(c);
}

}  // extern "C"

namespace {

// ---- the phases of one BFS level (shared by the single-GPU step and the sharded protocol) --------------------------
int level_error(vsrmc_checker* c, const LevelCtl& h, int new_level) {
  c->failed = 1;
  c->failed_code = (int)h.err;
  char buf[256];
  std::snprintf(buf, sizeof(buf), "device error %u at frontier index %llu ordinal %llu (level %d)", h.err,
                (unsigned long long)(h.err_info >> 16), (unsigned long long)(h.err_info & 0xFFFF), new_level);
  std::string msg = buf;
  if (h.err == ERR_EVAL_421) msg = "VSR.tla:421: record has no field 'commit' (ReceivePrepareMsg, ClientCount >= 2); " + msg;
  return fail(h.err < ERR_REP_RANGE ? VSRMC_E_EVAL : VSRMC_E_REP, msg);
}

// The tiles an ordinary single-pass launch wrote down because their enabled instances did not fit the LDS work list (k_expand: s_over; c->h.n_redo of
// them at c->h.redo_out as first record | size << 48): launched again — ONE launch over the list, every entry as two tiles of half its size, with the long
// work list — into the same destination: the level counters in the control block run on (only the tile cursor and the list start over), so the launches
// add up to what one launch with a longer list would have left.  Repeats while tiles are written down (a tile of 64 records is halved at most six
// times); a single record that does not fit is ERR_FRONTIER_FULL.
int redo_overflowed_tiles(vsrmc_checker* c, const void* kern, const u64* src_words, const u64* src_off, int level, int stride, u64* d_words, u64 d_wcap,
                          u64* d_off, u64 nx_cap, u64* d_fp, u32 ichunk, u32 wchunk, unsigned grid_cap /* the first launch's grid: its chunks fit the destination */,
                          int mode = MODE_NORMAL, u32 ccap_same = 0 /* the probe-only instantiation: the same short list — half a tile has half the instances */) {
  const Model& M = c->model.M;
  const u32 ccap = ccap_same ? ccap_same : M.R <= 3 ? (u32)VSR_CCAP64 : (u32)VSR_CAND_CAP;   // the long list (the short one is the five-block shape's)
  for (int round = 0; c->h.n_redo && !c->h.err && round < 8; round++) {   // (2^6 = a tile: the seventh round takes single records)
    const u64 n = c->h.n_redo;
    if (n > c->h.redo_cap) return fail(VSRMC_E_REP, "more tiles overflowed the work list of one launch than the list of them holds (65536)");
    const int in = (c->h.redo_out == (u64)(uintptr_t)c->redo_buf[0]) ? 0 : 1;      // what the last launch wrote is this one's input; it writes the other list
    LevelCtl patch = c->h;
    patch.n_redo = 0; patch.tile_cursor = 0; patch.redo_out = (u64)(uintptr_t)c->redo_buf[in ^ 1];
    HIPCHK(hipMemcpyAsync(&c->ctl->n_redo, &patch.n_redo, 3 * sizeof(u64), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(&c->ctl->tile_cursor, 0, 8, c->stream));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    const int tile = 64;
    const size_t lds = (size_t)tile * stride * 8 + 2 * (size_t)ccap * 4;
    const unsigned grid = (unsigned)std::max<u64>(1, std::min<u64>(std::min<u64>(2 * n, (u64)c->num_cus * 3), (u64)grid_cap));
    c->extra_launches++;
    hipLaunchKernelGGL((ExpandKernel)kern, dim3(grid), dim3(VSR_BLOCK), lds, c->stream, M, src_words, src_off, 2 * n * (u64)tile, level,
                       c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl, stride, 1, (u64*)nullptr, (u64)0, (u32)VSR_CAND_CAP,
                       d_words, d_wcap, d_off, nx_cap, d_fp, ichunk, wchunk, tile, ccap, (u64*)nullptr, (u64)0, (u64*)nullptr, 0u, mode | (int)MODE_REDO_LIST,
                       (u64)(uintptr_t)c->redo_buf[in], (const WSet*)nullptr, 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->expand_ms += ms;
  }
  if (!c->h.err && c->h.n_redo) { c->h.err = ERR_FRONTIER_FULL; c->h.err_info = 0; }
  return 0;
}

// phase 1: k_expand over the current frontier.  io == nullptr: unsharded.
int phase_expand(vsrmc_checker* c, const vsrmc_shard_io* io, int mode = MODE_NORMAL) {
  const Model& M = c->model.M;
  HIPCHK(hipSetDevice(c->opt.device));
  if (c->level + 1 >= 511) return fail(VSRMC_E_REP, "more than 510 BFS levels");
  c->t_level0 = now_s();
  c->expand_ms = c->materialize_ms = 0;
  std::memset(&c->h, 0, sizeof(c->h));
  c->h.viol_fp = ~(u64)0; c->h.redo_out = (u64)(uintptr_t)c->redo_buf[0]; c->h.redo_cap = 65536;
  HIPCHK(hipMemcpyAsync(c->ctl, &c->h, sizeof(c->h), hipMemcpyHostToDevice, c->stream));
  c->level_fused = !c->opt.exact_ties;
  const void* redo_kern = nullptr;                               // set for an ordinary (plain) launch: the instantiation that writes overflowed tiles down
  int redo_stride = 0;
  u32 redo_ichunk = 0, redo_wchunk = 0;
  unsigned redo_grid = 1;
  if (c->n_frontier > 0) {
    // 128 records per tile when the work list has room for them (about 4 successors per record at R <= 3), else 64
    const bool fused = !c->opt.exact_ties;                       // sharded (io != nullptr) or not
    // sharded: records arrive from other ranks (rebalancing), the local maximum says nothing -> the format's capacity
    // (a "sharded" run of ONE rank owns every fingerprint: nothing is ever remote, and the instantiation without the sharded branches does the level)
    const bool use_plain = fused && (!io || c->opt.world == 1) && mode == MODE_NORMAL && c->plain_kernel;
    const FusedShape fs = fused_shape(c, c->bag_known ? c->cur_max_bag : (u64)M.max_bag, use_plain);
    const int tile = fused ? fs.tile : (M.R <= 3 ? 128 : 64);
    const int stride = fused ? fs.stride : c->lds_stride;
    u64 ntiles = (c->n_frontier + tile - 1) / tile;
    // every block reserves pending-list room in chunks: the list must hold one chunk per block beyond the real entries
    const u32 pchunk = c->opt.pending_entries >= ((u64)1 << 24) ? 8192u : (u32)VSR_CAND_CAP;
    if (io && c->cand_idx_cap < (u64)c->opt.world * io->cand_cap) {   // per announced candidate: where it was written (fused) / its parent (exact)
      if (c->cand_idx) (void)hipFree(c->cand_idx);
      c->cand_idx = nullptr;
      c->cand_idx_cap = 0;
      HIPCHK(hipMalloc((void**)&c->cand_idx, (u64)c->opt.world * io->cand_cap * 8));
      c->cand_idx_cap = (u64)c->opt.world * io->cand_cap;
    }
    unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)c->num_cus * 3);
    if (!fused) grid = (unsigned)std::min<u64>(grid, std::max<u64>(1, c->opt.pending_entries / (4 * (u64)pchunk)));   // the pending list is only used by the two-kernel scheme
    const u32 ccap = fused ? fs.ccap : (tile == 128 ? 1536u : (u32)VSR_CAND_CAP);   // 128-record tiles: two blocks per CU in LDS
    size_t lds = (size_t)tile * stride * 8 + 2 * (size_t)ccap * 4;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    // fused single-pass mode (unsharded, not exact_ties): the lane that inserts a fingerprint writes the successor itself
    const int nxt = c->cur ^ 1;
    const u64 nx_cap = c->opt.frontier_states;
    u32 ichunk = 0, wchunk = 0, cchunk = 0;
    if (fused && io)   // candidate entries a block reserves per owner at a time: <= 1/4 of a bucket in total over all blocks
      cchunk = (u32)std::max<u64>(16, std::min<u64>(512, io->cand_cap / (4 * (u64)c->num_cus * 2)));
    if (fused) {
      // persistent blocks (2 resident per CU: 225 VGPRs, 79 KB LDS): every block leaves one partly used index chunk and
      // one word chunk behind per level, so fewer blocks = fewer unused slots in the next frontier
      grid = (unsigned)std::min<u64>((u64)ntiles, (u64)c->num_cus * fs.blocks_per_cu);
      grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, std::min<u64>(nx_cap / (4 * (u64)VSR_CAND_CAP), c->words_cap(nxt) / (4 * 16384))));
      // a tile's successors (at most ccap records of at most stride + 5 words each) must fit one word chunk, and every block
      // may leave one partly used chunk behind: fewer blocks if the buffer is too small for that
      // (a buffer too small even for one such chunk keeps going with what it has: the kernel refuses a tile that does not fit
      // its chunk with ERR_FRONTIER_FULL instead of writing past it)
      const u64 wmin = std::max<u64>(16384, (u64)ccap * (u64)(stride + 5));
      grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, c->words_cap(nxt) / (4 * wmin)));
      ichunk = (u32)std::max<u64>(VSR_CAND_CAP, std::min<u64>(8192, nx_cap / (4 * (u64)grid)));
      wchunk = (u32)std::max<u64>(std::min<u64>(wmin, c->words_cap(nxt) / 2), std::min<u64>(262144, c->words_cap(nxt) / (4 * (u64)grid)));
    }
    if (fused && use_plain) { redo_kern = fs.kernel; redo_stride = stride; redo_ichunk = ichunk; redo_wchunk = wchunk; redo_grid = grid; }
    if (fused)
      hipLaunchKernelGGL((ExpandKernel)(use_plain ? fs.kernel : c->fused_kernel), dim3(grid),
                         dim3(VSR_BLOCK), lds, c->stream, M, c->words[c->cur], c->off[c->cur],
                         c->n_frontier, c->level + 1, c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl,
                         stride, io ? c->opt.world : 1, io ? io->cand_send : nullptr, io ? io->cand_cap : 0, pchunk, c->words[nxt],
                         c->words_cap(nxt), c->off[nxt], nx_cap, c->lvl_fp, ichunk,
                         wchunk, tile, ccap, c->filter, c->fmask, c->cand_idx, cchunk,
                         mode | ((mode == MODE_PROBE && (c->saw_violation || c->probe_all_actions)) ? (int)MODE_NO_FOOTPRINT : 0), (u64)0, (const WSet*)nullptr, 0u);
    else
      hipLaunchKernelGGL(exact_kernel_for(M), dim3(grid), dim3(VSR_BLOCK), lds, c->stream, M, c->words[c->cur], c->off[c->cur],
                         c->n_frontier, c->level + 1, c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl,
                         c->lds_stride, io ? c->opt.world : 1, io ? io->cand_send : nullptr, io ? io->cand_cap : 0, pchunk, nullptr,
                         0, nullptr, 0, nullptr, 0, 0, tile, ccap, nullptr, 0, io ? c->cand_idx : nullptr, 0, 0, (u64)0, (const WSet*)nullptr, 0u);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
  }
  HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->n_frontier > 0) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->expand_ms = ms;
  }
  if (redo_kern && c->h.n_redo && !c->h.err) {                 // tiles that overflowed the (short) work list of an ordinary level: again, in halves
    const int nxt = c->cur ^ 1;
    const int rrc = redo_overflowed_tiles(c, redo_kern, c->words[c->cur], c->off[c->cur], c->level + 1, redo_stride, c->words[nxt], c->words_cap(nxt), c->off[nxt],
                                          c->opt.frontier_states, c->lvl_fp, redo_ichunk, redo_wchunk, redo_grid);
    if (rrc) return rrc;
  }
  c->nx_n = c->nx_w = 0;
  c->full_recoverable = false;
  if (!c->h.err && c->h.full) {                                // the record buffers ran out (LevelCtl::full): reported as before ..
    c->h.err = ERR_FRONTIER_FULL;
    c->h.err_info = c->h.full_info << 16;
    // .. but an unsharded ordinary level has lost nothing except the records: the automatic scheme goes on from the seen-set
    c->full_recoverable = !io && mode == MODE_NORMAL && !c->opt.exact_ties && c->opt.world == 1 && c->h.ties == 0;
    if (c->full_recoverable) { c->full_h = c->h; c->full_ms = c->expand_ms; c->full_t0 = c->t_level0; }
  }
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  if (!c->opt.exact_ties) {                                    // fused: the level is already materialised (sharded: speculatively)
    c->nx_n = c->h.n_new;
    c->nx_w = c->h.words_new;
    if (c->h.ties) {
      c->failed = 1;
      return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables "
                                 "(SURVEY F2); the single-pass scheme cannot arbitrate: create the checker with "
                                 "vsrmc_options.exact_ties = 1");
    }
  }
  return 0;
}

// phase 1 in SLICES (round 6: the overlapped exchange of vsr_shard_loop.hpp): k_expand over parents [first, first + n) of the current frontier of a sharded
// single-pass level, announcing into the caller's buckets (one of two sets) and APPENDING to the next frontier — the level counters in the control block
// run on from slice to slice (only the tile cursor and the bucket fills start over), so that the slices of a level add up to what one launch would have left.
// Asynchronous: the launch is queued on the checker's stream between two events and the call returns; expand_slice_wait() collects it.
// A block leaves a partly used index and word chunk behind per launch: chunks are smaller here (2048 indices, 64 K words) than for a whole level.
int expand_slice_launch(vsrmc_checker* c, const vsrmc_shard_io* io, u64* cand_idx, u64 first, u64 n, bool first_slice) {
  const Model& M = c->model.M;
  HIPCHK(hipSetDevice(c->opt.device));
  if (first_slice) {
    if (c->level + 1 >= 511) return fail(VSRMC_E_REP, "more than 510 BFS levels");
    c->t_level0 = now_s();
    c->expand_ms = c->materialize_ms = 0;
    std::memset(&c->h, 0, sizeof(c->h));
    c->h.viol_fp = ~(u64)0; c->h.redo_out = (u64)(uintptr_t)c->redo_buf[0]; c->h.redo_cap = 65536;
    HIPCHK(hipMemcpyAsync(c->ctl, &c->h, sizeof(c->h), hipMemcpyHostToDevice, c->stream));
    c->level_fused = true;
    c->nx_n = c->nx_w = 0;
    c->full_recoverable = false;
  } else {
    HIPCHK(hipMemsetAsync(&c->ctl->tile_cursor, 0, sizeof(u64), c->stream));
    HIPCHK(hipMemsetAsync(c->ctl->cand_cnt, 0, sizeof(c->ctl->cand_cnt), c->stream));
  }
  HIPCHK(hipEventRecord(c->ev[0], c->stream));
  if (n > 0) {
    const FusedShape fs = fused_shape(c, c->bag_known ? c->cur_max_bag : (u64)M.max_bag, false);
    const int nxt = c->cur ^ 1;
    const u64 nx_cap = c->opt.frontier_states;
    const u64 ntiles = (n + fs.tile - 1) / fs.tile;
    unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)c->num_cus * fs.blocks_per_cu);
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, std::min<u64>(nx_cap / (4 * (u64)VSR_CAND_CAP), c->words_cap(nxt) / (4 * 16384))));
    const u64 wmin = std::max<u64>(16384, (u64)fs.ccap * (u64)(fs.stride + 5));
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, c->words_cap(nxt) / (4 * wmin)));
    const u32 ichunk = (u32)std::max<u64>(VSR_CAND_CAP, std::min<u64>(2048, nx_cap / (4 * (u64)grid)));
    const u32 wchunk = (u32)std::max<u64>(std::min<u64>(wmin, c->words_cap(nxt) / 2), std::min<u64>(65536, c->words_cap(nxt) / (4 * (u64)grid)));
    const u32 cchunk = (u32)std::max<u64>(16, std::min<u64>(512, io->cand_cap / (4 * (u64)c->num_cus * 2)));
    hipLaunchKernelGGL((ExpandKernel)c->fused_kernel, dim3(grid), dim3(VSR_BLOCK), fs.lds, c->stream, M, c->words[c->cur], c->off[c->cur] + first,
                       n, c->level + 1, c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl,
                       fs.stride, c->opt.world, io->cand_send, io->cand_cap, (u32)VSR_CAND_CAP, c->words[nxt],
                       c->words_cap(nxt), c->off[nxt], nx_cap, c->lvl_fp, ichunk, wchunk, fs.tile, fs.ccap, c->filter, c->fmask, cand_idx, cchunk,
                       (int)MODE_NORMAL, (u64)0, (const WSet*)nullptr, 0u);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipEventRecord(c->ev[1], c->stream));
  HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
  return 0;
}
// ... the slice has run: its kernel time, its errors, how many candidates it left in each owner's bucket
int expand_slice_wait(vsrmc_checker* c, u64* cand_counts) {
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
  c->expand_ms += ms;
  for (int o = 0; o < c->opt.world; o++) cand_counts[o] = c->h.cand_cnt[o];
  if (!c->h.err && c->h.full) { c->h.err = ERR_FRONTIER_FULL; c->h.err_info = c->h.full_info << 16; }
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  return 0;
}

// phase 2: k_materialize over a list of (slot-or-fp, key) entries into one target (next frontier or a peer's bucket)
// (src_words / src_off: where the parents are read from — default: the current frontier; a slice of another buffer in the deep search)
int phase_materialize(vsrmc_checker* c, const u64* entries, u64 n, const uint8_t* verdict, u64* t_words, u64 t_words_cap,
                      u64* t_off, u64 t_cap, u64* t_fp, u64* cnt_n, u64* cnt_w, int entry_words, const u64* pidx_arr,
                      const u64* src_words = nullptr, const u64* src_off = nullptr) {
  if (n == 0) return 0;
  if (!src_words) { src_words = c->words[c->cur]; src_off = c->off[c->cur]; }
  const Model& M = c->model.M;
  // persistent waves: each keeps private output chunks, so the grid is sized to what is resident (LDS: 5 waves / CU)
  u64 grid64 = std::min<u64>((n + VSR_MAT_BLOCK - 1) / VSR_MAT_BLOCK, (u64)c->num_cus * 5);
  const u64 min_wchunk = (u64)VSR_MAT_BLOCK * c->lds_stride;
  grid64 = std::max<u64>(1, std::min<u64>(grid64, std::min<u64>(t_words_cap / (4 * min_wchunk), t_cap / (4 * 64))));
  const u32 ichunk = (u32)std::min<u64>(1024, std::max<u64>(64, t_cap / (4 * grid64)));
  const u32 wchunk = (u32)std::min<u64>(65536, std::max<u64>(min_wchunk, t_words_cap / (4 * grid64)));
  unsigned grid = (unsigned)grid64;
  size_t lds = (size_t)VSR_MAT_BLOCK * c->lds_stride * 8;
  HIPCHK(hipEventRecord(c->ev[2], c->stream));
  hipLaunchKernelGGL(materialize_kernel_for(M), dim3(grid), dim3(VSR_MAT_BLOCK), lds, c->stream, M, src_words, src_off, entries, n,
                     c->table, t_words, t_words_cap, t_off, t_cap, t_fp, c->ctl, verdict, cnt_n, cnt_w, c->lds_stride, ichunk, wchunk,
                     entry_words, pidx_arr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev[3], c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  c->materialize_ms += ms;
  return 0;
}

// materialise the local pending list straight into the next frontier (self bucket)
int phase_materialize_local(vsrmc_checker* c) {
  u64 n_pending = std::min<u64>(c->h.n_pending, c->opt.pending_entries);
  const u64 nx_cap = c->opt.frontier_states;
  const int nxt = c->cur ^ 1;
  int rc = phase_materialize(c, c->pending, n_pending, nullptr, c->words[nxt], c->words_cap(nxt), c->off[nxt], nx_cap,
                             c->lvl_fp, &c->ctl->n_new, &c->ctl->words_new, 3, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost));
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  c->nx_n = c->h.n_new;
  c->nx_w = c->h.words_new;
  return 0;
}

// phase 3: the level is complete: swap the frontiers, fill in the local statistics
int phase_commit(vsrmc_checker* c, vsrmc_level_info* info) {
  std::memset(info, 0, sizeof(*info));
  const LevelCtl& h = c->h;
  info->frontier = c->n_frontier;
  info->generated = h.generated;
  info->deadlocks = h.deadlocks;
  info->pending = h.n_pending;
  info->probes = h.probes;
  info->max_bag = h.max_bag;
  for (int a = 0; a < 16; a++) info->act_generated[a] = h.act_generated[a];
  for (int a = 0; a < 8; a++) info->phase_cycles[a] = h.phase_cycles[a];
  info->viol_fp = ~(u64)0;
  info->viol_index = ~(u64)0;
  info->expand_ms = c->expand_ms;
  info->materialize_ms = c->materialize_ms;
  // nx_n is an index RANGE: waves allocate indices in chunks and publish unused ones as invalid refs (0)
  u64 n_new = 0;
  if (c->nx_n > 0 && c->opt.world == 1 && !c->opt.exact_ties) {
    n_new = h.n_written;                                         // single-pass, unsharded: every record written is a new state
  } else if (c->nx_n > 0) {
    u64 zero = 0;
    HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_count_valid, dim3(1024), dim3(256), 0, c->stream, c->off[c->cur ^ 1], c->nx_n, c->d_find);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&n_new, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->total_generated += h.generated;
  info->n_new = n_new;
  info->words_new = c->nx_w;
  info->record_words = h.rec_words;
  if (n_new > 0 || c->opt.world > 1) {   // sharded: levels stay aligned across ranks even when this shard got nothing
    c->cur ^= 1;
    c->level += 1;
    c->distinct += n_new;
    c->n_frontier = c->nx_n;
    c->n_valid = n_new;
    c->cur_w = c->nx_w;
    c->cur_max_bag = h.max_bag;
    c->bag_known = c->opt.world == 1;
    c->hist_new[0] = c->hist_new[1];
    c->hist_new[1] = n_new;
    c->g_last = (h.generated + std::max<u64>(1, info->frontier) - 1) / std::max<u64>(1, info->frontier) + 1;
    c->cur_rec_w = h.rec_words;
  } else {
    c->n_frontier = 0;
    c->n_valid = 0;
  }
  if (h.viol_fp != ~(u64)0) {
    info->viol_fp = h.viol_fp;
    info->viol_mask = (int32_t)h.viol_mask;
    c->saw_violation = true;
  }
  info->level = c->level;
  info->distinct = c->distinct;
  info->total_generated = c->total_generated;
  info->seconds = now_s() - c->t_level0;
  return 0;
}

int find_fp_newest(vsrmc_checker* c, u64 fp, u64* idx) {
  *idx = ~(u64)0;
  if (c->n_frontier == 0) return 0;
  u64 big = ~(u64)0;
  HIPCHK(hipMemcpy(c->d_find, &big, 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_find_fp, dim3((unsigned)((c->n_frontier + 255) / 256)), dim3(256), 0, c->stream, c->lvl_fp, c->n_frontier, fp,
                     c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(idx, c->d_find, 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace

extern "C" {

static int32_t step_local(vsrmc_checker* c, vsrmc_level_info* info) {
  int rc = phase_expand(c, nullptr);
  if (!rc && c->opt.exact_ties) rc = phase_materialize_local(c);
  if (rc) {   // like a TLC evaluation error: the run aborts, the partial level is not committed
    std::memset(info, 0, sizeof(*info));
    info->level = c->level;
    info->distinct = c->distinct;
    info->error_code = (int32_t)c->h.err;
    return rc;
  }
  rc = phase_commit(c, info);
  if (rc) return rc;
  if (info->viol_mask) return find_fp_newest(c, info->viol_fp, &info->viol_index);
  return 0;
}

namespace {
// One single-pass launch over an arbitrary source (a slice of the newest level, or the partial next frontier a MODE_REGEN
// slice just wrote), unsharded.  Resets the level counters, returns them in c->h.  Destination = the next-frontier buffers.
// io != nullptr: a pass of a sharded run (vsr_deep.hpp) — successors owned by other ranks are announced into io's buckets
int expand_pass(vsrmc_checker* c, const u64* src_words, const u64* src_off, u64 n_parents, u64 p_offset, int level, int mode,
                u64 src_max_bag, const PassDst* dst = nullptr, const vsrmc_shard_io* io = nullptr,
                u32* claim_bits = nullptr, u64 claim_w = 0 /* unsharded MODE_INSERT / MODE_REGEN over the stored base: the claim bitmap (vsr_deep.hpp) */) {
  const Model& M = c->model.M;
  std::memset(&c->h, 0, sizeof(c->h));
  c->h.viol_fp = ~(u64)0; c->h.redo_out = (u64)(uintptr_t)c->redo_buf[0]; c->h.redo_cap = 65536;
  HIPCHK(hipMemcpyAsync(c->ctl, &c->h, sizeof(c->h), hipMemcpyHostToDevice, c->stream));
  c->extra_launches = 0;
  u32 probe_ccap = 256u;
  bool use_probe = false;                                        // the probe-only instantiation ran: its failing successors are (parent, ordinal) entries still to be resolved
  const void* redo_kern = nullptr;                               // (see phase_expand)
  int redo_stride = 0;
  u32 redo_ichunk = 0, redo_wchunk = 0;
  unsigned redo_grid = 1;
  u64 *redo_words = nullptr, *redo_off = nullptr, *redo_fp = nullptr, redo_wcap = 0, redo_cap = 0;
  if (n_parents > 0) {
    // an ordinary level into other buffers (the streamed level's sub-slices) runs the plain instantiation: the code of a stored level
    static const bool plain_normal = std::getenv("VSRMC_STREAM_MODES_KERNEL") == nullptr;
    const bool one_rank = !io || c->opt.world == 1;                // (world 1 through the level loop: nothing is remote — the unsharded instantiations)
    const bool use_plain = one_rank && mode == MODE_NORMAL && plain_normal && c->plain_kernel;
    FusedShape fs = fused_shape(c, src_max_bag, use_plain);
    // a probe pass (nothing seen to violate so far, no limit re-check): the probe-only instantiation at five blocks per CU when its tile — a work list of 256
    // entries: only the footprint's instances are listed — fits the LDS five times (same measured bound as the five-block ordinary level: fused_shape)
    if (one_rank && mode == MODE_PROBE && c->probe_kernel && !c->saw_violation && !c->probe_all_actions && fs.tile == 64) {
      hipFuncAttributes at;
      const u32 pcap = std::getenv("VSRMC_PROBE_CCAP") ? (u32)std::max(32, std::min(256, std::atoi(std::getenv("VSRMC_PROBE_CCAP")))) : 256u;   // (tests: a list that overflows; 32 = one bag entry per record and batch of the enumeration)
      const size_t dyn = (size_t)64 * fs.stride * 8 + 2 * (size_t)pcap * 4;
      static const size_t lds5 = std::getenv("VSRMC_LDS5") ? (size_t)std::atoll(std::getenv("VSRMC_LDS5")) : (size_t)31744;
      int nb5 = 0;
      if (hipFuncGetAttributes(&at, c->probe_kernel) == hipSuccess && dyn + at.sharedSizeBytes <= lds5 &&
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb5, c->probe_kernel, VSR_BLOCK, dyn) == hipSuccess && nb5 >= VSR_OCC + 1) {
        use_probe = true;
        probe_ccap = pcap;
        fs.kernel = c->probe_kernel; fs.ccap = pcap; fs.lds = dyn; fs.blocks_per_cu = (unsigned)(VSR_OCC + 1);
      }
    }
    u32 cchunk = 0;
    if (io) {
      if (c->cand_idx_cap < (u64)c->opt.world * io->cand_cap) {   // per announced candidate: where it was written / what regenerates it
        if (c->cand_idx) (void)hipFree(c->cand_idx);
        c->cand_idx = nullptr;
        c->cand_idx_cap = 0;
        HIPCHK(hipMalloc((void**)&c->cand_idx, (u64)c->opt.world * io->cand_cap * 8));
        c->cand_idx_cap = (u64)c->opt.world * io->cand_cap;
      }
      cchunk = (u32)std::max<u64>(16, std::min<u64>(512, io->cand_cap / (4 * (u64)c->num_cus * 2)));
    }
    const int tile = fs.tile;
    const u64 ntiles = (n_parents + tile - 1) / tile;
    const u32 ccap = fs.ccap;
    const size_t lds = fs.lds;
    const int nxt = c->cur ^ 1;
    u64* const d_words = dst ? dst->words : c->words[nxt];
    u64* const d_off = dst ? dst->off : c->off[nxt];
    u64* const d_fp = dst ? dst->fp : c->lvl_fp;
    const u64 d_wcap = dst ? dst->words_cap : c->words_cap(nxt);
    const u64 nx_cap = dst ? dst->cap : c->opt.frontier_states;
    unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)c->num_cus * fs.blocks_per_cu);
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, std::min<u64>(nx_cap / (4 * (u64)VSR_CAND_CAP), d_wcap / (4 * 16384))));
    const u64 wmin = std::max<u64>(16384, (u64)ccap * (u64)(fs.stride + 5));   // see phase_expand
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, d_wcap / (4 * wmin)));
    // index chunks: a block leaves the unused tail of its last chunk behind as invalid refs, and the NEXT pass stages those holes like records.
    // A whole level (2.6e8 states) loses 1-3 % to 8192-index chunks; a sub-slice of a streamed level (7e6 states from 1024 blocks) lost a third
    // of its index range, and the probe pass over it 16 % of its time (VSRMC_ICHUNK=8192: the old size, for A/B runs).
    static const u64 ichunk_small = std::getenv("VSRMC_ICHUNK") ? (u64)std::atoll(std::getenv("VSRMC_ICHUNK")) : 2048;
    const u64 ichunk_max = (dst || mode == MODE_REGEN) ? std::max<u64>(VSR_CAND_CAP, ichunk_small) : 8192;
    const u32 ichunk = (u32)std::max<u64>(VSR_CAND_CAP, std::min<u64>(ichunk_max, nx_cap / (4 * (u64)grid)));
    const u32 wchunk = (u32)std::max<u64>(std::min<u64>(wmin, d_wcap / 2), std::min<u64>(262144, d_wcap / (4 * (u64)grid)));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    const void* kern = !one_rank ? c->fused_kernel : (use_plain || use_probe) ? fs.kernel : (c->modes_kernel ? c->modes_kernel : c->fused_kernel);
    // the claim bitmap of the first seen-set-only level (vsr_deep.hpp): written by the pass that inserts it, read by every pass that regenerates it — both
    // expand the stored base (level c->level), slice by slice, p_offset = the slice's first parent
    if (!claim_bits && c->claim_bits && one_rank && level == c->level + 1 && (mode == MODE_INSERT || mode == MODE_REGEN) &&
        p_offset + n_parents <= c->claim_parents) { claim_bits = c->claim_bits; claim_w = c->claim_w; }
    if (one_rank && !use_plain) {                                // a pass that knows its mode: the instantiation with only that mode in it, when the configuration has one
      if (mode == MODE_REGEN && claim_bits && c->regen_bits_kernel && claim_w <= 2 * (u64)(VSR_BLOCK / fs.tile)) kern = c->regen_bits_kernel;   // (two bitmap words per thread)
      else if (mode == MODE_INSERT && c->insert_kernel) kern = c->insert_kernel;
    }
    if (claim_bits && kern != c->regen_bits_kernel && kern != c->insert_kernel) { claim_bits = nullptr; claim_w = 0; }   // (only those two know the bitmap)
    if (use_plain || use_probe) { redo_kern = kern; redo_stride = fs.stride; redo_ichunk = ichunk; redo_wchunk = wchunk; redo_words = d_words; redo_off = d_off; redo_fp = d_fp; redo_wcap = d_wcap; redo_cap = nx_cap; redo_grid = grid; }
    hipLaunchKernelGGL((ExpandKernel)kern, dim3(grid), dim3(VSR_BLOCK), lds, c->stream, M, src_words, src_off, n_parents, level, c->opt.rank,
                       c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl, fs.stride, io ? c->opt.world : 1,
                       io ? io->cand_send : nullptr, io ? io->cand_cap : (u64)0, (u32)VSR_CAND_CAP,
                       d_words, d_wcap, d_off, nx_cap, d_fp,
                       ichunk, wchunk, tile, ccap, !one_rank ? c->filter : (u64*)claim_bits, !one_rank ? c->fmask : claim_w, io ? c->cand_idx : nullptr, cchunk,
                       mode | ((mode == MODE_PROBE && (c->saw_violation || c->probe_all_actions)) ? (int)MODE_NO_FOOTPRINT : 0), p_offset,
                       (const WSet*)((io && c->opt.world > 1) ? c->d_wset : nullptr), c->wepoch);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
  }
  HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_parents > 0) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->expand_ms += ms;
  }
  if (redo_kern && !use_probe && c->h.n_redo && !c->h.err) {   // tiles that overflowed the work list of an ordinary pass: again, in halves
    const int rrc = redo_overflowed_tiles(c, redo_kern, src_words, src_off, level, redo_stride, redo_words, redo_wcap, redo_off, redo_cap, redo_fp, redo_ichunk, redo_wchunk, redo_grid);
    if (rrc) return rrc;
  }
  if (use_probe && c->h.n_redo > c->h.redo_cap && !c->h.err) c->h.err = ERR_FRONTIER_FULL;   // (more such tiles than the list holds: the general kernel, below)
  if (use_probe && c->h.n_redo && !c->h.err) {                 // tiles with more footprint instances than the short list holds: again, in halves, by the same kernel
    const int rrc = redo_overflowed_tiles(c, c->probe_kernel, src_words, src_off, level, redo_stride, redo_words, redo_wcap, redo_off, redo_cap, redo_fp, redo_ichunk, redo_wchunk,
                                          redo_grid, MODE_PROBE, probe_ccap);
    if (rrc) return rrc;
  }
  if (use_probe && c->h.err == ERR_FRONTIER_FULL) {            // a tile with more footprint instances than the short work list holds: the pass again with the general probe kernel
    void* const pk = c->probe_kernel;
    const u64 before = c->extra_launches + 1;
    c->probe_kernel = nullptr;
    const int rrc = expand_pass(c, src_words, src_off, n_parents, p_offset, level, mode, src_max_bag, dst, io, claim_bits, claim_w);
    c->probe_kernel = pk;
    c->extra_launches += before;
    return rrc;
  }
  if (use_probe && c->h.n_pending && !c->h.err) {              // the failing successors the probe-only pass wrote down: fingerprinted and looked up now (k_probe_resolve)
    const u64 n = c->h.n_pending;
    if (n > c->opt.pending_entries) return fail(VSRMC_E_REP, "more violating successors in the probed level than the pending list holds (pending_entries)");
    u64* d_out = nullptr;
    HIPCHK(hipMalloc((void**)&d_out, n * 16));
    struct FreeOut { u64* p; ~FreeOut() { (void)hipFree(p); } } free_out{d_out};
    u64 zero = 0, kept = 0;
    HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_probe_resolve<0>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, M, src_words, src_off, (const u64*)c->pending, n, (const Slot*)c->table,
                       c->tmask, level, d_out, n, (unsigned long long*)c->d_find, c->ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&kept, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (kept) HIPCHK(hipMemcpyAsync(c->pending, d_out, kept * 16, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(&c->ctl->n_pending, &kept, 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (!c->h.err && c->h.full) {                                // a pass into buffers that ran out (LevelCtl::full): an error here — the caller sized the slice
    c->h.err = ERR_FRONTIER_FULL;
    c->h.err_info = c->h.full_info << 16;
  }
  if (c->h.err) return level_error(c, c->h, level);
  if (c->h.ties) {
    c->failed = 1;
    return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2)");
  }
  return 0;
}

// one step of a trace walk through the seen-set (k_table_lookup): by_low_bits = 0: the slot of fingerprint `key`; 1: the slot of
// the level-`level` state whose fingerprint ends in the 45 bits `key` (what a child's meta word knows of its parent)
int table_lookup(vsrmc_checker* c, u64 key, int level, int by_low_bits, int* found, u64* fp, u64* meta) {   // *found = matching states
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 24));
  hipLaunchKernelGGL(k_table_lookup, dim3(1), dim3(64), 0, c->stream, c->table, c->tmask, key, level, by_low_bits, d);
  u64 h[3] = {0, 0, 0};
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return fail(VSRMC_E_HIP, "k_table_lookup failed");
  *found = (int32_t)std::min<u64>(h[0], 0x7FFFFFFF);          // by_low_bits: the number of matching states (> 1: ambiguous)
  *fp = h[1];
  *meta = h[2];
  return 0;
}
// TLCTrace.getTrace, backwards half: the fingerprints of the path Init -> the level-`level` state with fingerprint `fp`
int walk_trace(vsrmc_checker* c, u64 fp, int level, std::vector<u64>* fps) {
  if (level < 1) return fail(VSRMC_E_ARG, "no such level");
  fps->assign((size_t)level + 1, 0);
  u64* d_fps = nullptr;
  HIPCHK(hipMalloc((void**)&d_fps, ((u64)level + 1) * 8));
  bool ok = hipMemsetAsync(d_fps, 0, ((u64)level + 1) * 8, c->stream) == hipSuccess;
  hipLaunchKernelGGL(k_trace_walk, dim3(1), dim3(64), 0, c->stream, c->table, c->tmask, fp, level, d_fps);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
       hipMemcpy(fps->data(), d_fps, ((u64)level + 1) * 8, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d_fps);
  if (!ok) return fail(VSRMC_E_HIP, "k_trace_walk failed");
  const u64 status = fps->back();
  fps->pop_back();
  if ((status & 0xFF) == 2) {
    char buf[256];
    std::snprintf(buf, sizeof(buf), "ambiguous predecessor pointer: %llu states of level %llu share the 45 fingerprint bits a successor keeps of its "
                  "parent (expected about once in 2^45 / level size steps); the counter-example cannot be walked through the seen-set",
                  (unsigned long long)(status >> 16), (unsigned long long)((status >> 8) & 0xFF));
    return fail(VSRMC_E_STATE, buf);
  }
  if (status != 0 || (*fps)[0] == 0) return fail(VSRMC_E_STATE, "the seen-set holds no path from Init to this state at this level");
  return 0;
}

// smallest-fingerprint violator of the (fp, key) list a PROBE / INSERT pass left in c->pending; among equal fps the smallest key
int min_violator(vsrmc_checker* c, u64 fp_min, u64* key) {
  *key = ~(u64)0;
  const u64 n = std::min<u64>(c->h.n_pending, std::min<u64>(c->opt.pending_entries, (u64)1 << 20));
  std::vector<u64> list(2 * n);
  if (n) HIPCHK(hipMemcpy(list.data(), c->pending, 16 * n, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < n; i++)
    if (list[2 * i] == fp_min && list[2 * i + 1] < *key) *key = list[2 * i + 1];
  return 0;
}
}  // namespace

}  // extern "C"

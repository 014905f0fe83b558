// vsr_kernels.hpp — CDNA4 (gfx950, wave64) kernels of the BFS hot path.
//
// TLC roles replaced:
//   k_expand       Worker.run -> ModelChecker.doNext -> Tool.getNextStates + TLCState.fingerPrint + FPSet.put
//                  (rows a7, a10, a11): stage a tile of frontier records in LDS, enumerate the enabled
//                  (action, binding) instances, apply each one, hash incrementally, probe/insert the seen-set.
//   k_materialize  StateQueue.sEnqueue + TLCTrace.writeState + invariant check (rows a9, a12, a13): the winner of
//                  every newly claimed slot rebuilds its successor and appends it to the next frontier.
//   k_fpset_*      FPSet.putBlock / containsBlock (row a11) as a standalone batch API.
//   k_successors   Tool.getNextStates over all actions for a batch of states (boundary / parity-test entry).
//   k_replay       TLCTrace.getTrace: re-executes a path of ordinals from Init.
// No floating point, no MFMA: integer compare/shift/multiply, LDS staging, 64-bit atomics in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include "vsr_actions.hpp"
#include "vrst_actions.hpp"
#include "vras_actions.hpp"

namespace vsr {

// The model a kernel instantiation checks: 0 = VSR.tla (vsr_actions.hpp), 1 = analysis/03-state-transfer/VR_STATE_TRANSFER.tla
// (vrst_actions.hpp), 2 = analysis/04-application-state/VR_APP_STATE.tla (vras_actions.hpp).  Everything above the action table — staging, enumeration, sort, seen-set, frontier, trace — is shared.
template <int MODEL>
struct ModelOps;
template <>
struct ModelOps<0> {
  template <bool GUARD_ONLY, typename PTR>
  static VSR_HD bool gen_(const Model& M, PTR rec, int ord, Delta& D) { return gen<GUARD_ONLY>(M, rec, ord, D); }
  template <typename PTR>
  static VSR_HD u32 guard_pre(const Model& M, PTR rec, u64 hdr, const u64* Areg, int slot, int* kind0, int info) {
    return guard_slot_pre(M, rec, hdr, Areg, slot, kind0, info);
  }
  template <typename PTR>
  static VSR_HD u32 guard(const Model& M, PTR rec, int slot, int* kind0) { return guard_slot(M, rec, slot, kind0); }
  static VSR_HD int other_kind(int) { return A_SendGetState; }    // bits 1.. of a slot: SendGetState(r, rDest, m), one per destination
  template <typename PTR>
  static VSR_HD void hash_child_(const Model& M, PTR rec, const Delta& D, u64* Hc) { hash_child(M, rec, D, Hc); }
  template <typename PTR>
  static VSR_HD void hash_full_(const Model& M, PTR rec, u64* H) { hash_full(M, rec, H); }
  template <typename PTR>
  static VSR_HD int invariants(const Model& M, PTR rec, const Delta& D) { return check_invariants_child(M, rec, D); }
  // The invariants (VSR.tla:933-950) read rep_log and aux_client_acked only: a successor whose action writes neither has its parent's verdict, and
  // every expanded parent has passed.  These are the actions that write one of the two (or can raise a TLC evaluation error): the probe level runs
  // only them (tests/test_probe_footprint.py holds the claim against the CPU oracle, successor by successor).
  static VSR_HD u32 probe_actions() {
    return (1u << A_SendSV) | (1u << A_ExecuteOp) | (1u << A_ReceiveClientRequest) | (1u << A_SendGetState) | (1u << A_ReceiveSV) |
           (1u << A_ReceivePrepareMsg) | (1u << A_ReceiveGetState) | (1u << A_ReceiveNewState);
  }
};
template <>
struct ModelOps<1> {
  template <bool GUARD_ONLY, typename PTR>
  static VSR_HD bool gen_(const Model& M, PTR rec, int ord, Delta& D) { return vrst::gen<GUARD_ONLY>(M, rec, ord, D); }
  template <typename PTR>
  static VSR_HD u32 guard_pre(const Model& M, PTR rec, u64, const u64*, int slot, int* kind0, int) {
    return vrst::guard_slot(M, rec, slot, kind0);
  }
  template <typename PTR>
  static VSR_HD u32 guard(const Model& M, PTR rec, int slot, int* kind0) { return vrst::guard_slot(M, rec, slot, kind0); }
  static VSR_HD int other_kind(int kind0) { return kind0; }       // bits 1.. of a slot: the same action taken by another replica (AnyDest)
  template <typename PTR>
  static VSR_HD void hash_child_(const Model& M, PTR rec, const Delta& D, u64* Hc) { vrst::hash_child(M, rec, D, Hc); }
  template <typename PTR>
  static VSR_HD void hash_full_(const Model& M, PTR rec, u64* H) { vrst::hash_full(M, rec, H); }
  template <typename PTR>
  static VSR_HD int invariants(const Model& M, PTR rec, const Delta& D) { return vrst::check_invariants_child(M, rec, D); }
  static VSR_HD u32 probe_actions() { return ~0u; }               // not analysed: every action is run
};

template <>
struct ModelOps<2> {   // analysis/04-application-state/VR_APP_STATE.tla (vras_actions.hpp)
  template <bool GUARD_ONLY, typename PTR>
  static VSR_HD bool gen_(const Model& M, PTR rec, int ord, Delta& D) { return vras::gen<GUARD_ONLY>(M, rec, ord, D); }
  template <typename PTR>
  static VSR_HD u32 guard_pre(const Model& M, PTR rec, u64, const u64*, int slot, int* kind0, int) {
    return vras::guard_slot(M, rec, slot, kind0);
  }
  template <typename PTR>
  static VSR_HD u32 guard(const Model& M, PTR rec, int slot, int* kind0) { return vras::guard_slot(M, rec, slot, kind0); }
  static VSR_HD int other_kind(int kind0) { return kind0; }       // bits 1.. of a slot: the same action taken by another replica (AnyDest)
  template <typename PTR>
  static VSR_HD void hash_child_(const Model& M, PTR rec, const Delta& D, u64* Hc) { vras::hash_child(M, rec, D, Hc); }
  template <typename PTR>
  static VSR_HD void hash_full_(const Model& M, PTR rec, u64* H) { vras::hash_full(M, rec, H); }
  template <typename PTR>
  static VSR_HD int invariants(const Model& M, PTR rec, const Delta& D) { return vras::check_invariants_child(M, rec, D); }
  static VSR_HD u32 probe_actions() { return ~0u; }
};

struct Slot {       // one seen-set slot: 16 bytes, fp == 0 means empty
  u64 fp;
  u64 meta;
};

struct LevelCtl {   // device-resident counters of one BFS level
  u64 generated;    // enabled (action, binding) instances = successors generated (TLC "states generated")
  u64 deadlocks;    // frontier states without any successor
  u64 n_pending;    // candidates that hit a slot claimed during this level
  u64 n_new;        // successors appended to the next frontier = new distinct states
  u64 words_new;    // words used in the next frontier
  u64 viol_fp;      // smallest fingerprint among new states violating an invariant (~0 = none)
  u64 probes;       // seen-set slots inspected
  u64 max_bag;      // largest bag among the new states
  u32 viol_mask;
  u32 err;          // first ERR_* raised
  u64 err_info;     // (parent index << 16) | ordinal of the instance that raised it
  u64 act_generated[16];   // generated successors per action id
  u64 cand_cnt[8];         // sharded mode: candidates bucketed for each owner rank
  u64 rec_words;           // words of the records actually written to the next frontier (chunk slack excluded)
  u64 ties;                // fused mode: same-level candidates of one fingerprint with different auxkeys (must stay 0)
  // shader-clock breakdown (cheap, always on; printed by tools/run_bfs.py).  k_expand, wave 0 of every block: [0..4] stage,
  // enumerate, sort, apply, tail; fused apply loop of thread 0: [5..7] gen, hash, probe (+ act_generated[0] = successor write);
  // k_materialize (two-kernel levels) adds its lane-0 clocks to [5..7]: fetch + stage, gen + patch, allocate + write
  u64 phase_cycles[8];
  u64 tile_cursor;         // k_expand: the next frontier tile no block has taken yet
  u64 n_written;           // fused mode: records written to the next frontier (unsharded: = new states; sharded: incl. speculative ones)
  u64 fp_xor, fp_sum;      // MODE_INSERT (virtual level): xor / sum of the fingerprints this pass inserted = the level's checksums (no lvl_fp array exists)
  // single-pass levels: the next-frontier buffers ran out of index or word chunks.  NOT an `err`: every successor is still enumerated, hashed,
  // claimed, counted and checked — only the records (and refs / lvl_fp entries) written after that moment are garbage — so the seen-set holds the
  // complete level and the host can keep it as a seen-set-only level (host_search.hpp: adopt_overflowed_level) instead of failing.  `err` stays
  // free for the errors that DO lose successors (work list too small, a tile refused, a candidate bucket full) or abort the run.
  u32 full;
  u32 full_pad_;
  u64 full_info;
  // probe level with the footprint filter on: instances that were NOT applied although a record of their tile sits at a representation limit (a bag within
  // R - 1 entries of its capacity, a delivery count of 3): a successor of theirs could have raised ERR_REP_* unseen.  Not 0: the host runs the pass again
  // with every action applied (host_checker.hpp: expand_pass) — nothing goes unreported, and vsrmc_level_info.limit_rechecked says it happened.
  u64 limit_unchecked;
  // ordinary single-pass levels: the tiles whose enabled instances did not fit the LDS work list (k_expand: s_over) — n_redo of them written to the list at
  // redo_out (a device pointer the host puts here, redo_cap entries: first record | size << 48); the host launches them again in halves (redo_overflowed_tiles)
  u64 n_redo, redo_out, redo_cap;
};

// owner rank of a fingerprint: high bits, so that the table index (low bits) stays uniform inside a shard
__host__ __device__ __forceinline__ int owner_of(u64 fp, int world) { return (int)(((fp >> 40) & 0xFFFFFF) % (u64)world); }

enum { MODE_NORMAL = 0, MODE_PROBE = 1, MODE_INSERT = 2, MODE_REGEN = 3, MODE_NO_FOOTPRINT = 0x100 /* flag, or-ed to MODE_PROBE */,
       MODE_REDO_LIST = 0x200 /* flag of an ordinary launch: the tiles come from the list of overflowed tiles at p_offset (host_checker.hpp: redo_overflowed_tiles) */ };
#define VSR_TILE_MAX 128     // frontier records staged per block iteration: 64 or 128 (kernel parameter `tile`)
#define VSR_BLOCK 256
// per-phase shader clocks of k_expand (vsrmc_level_info.phase_cycles; tools/run_bfs.py prints the breakdown): every read is an
// s_memtime that waits for the wave's outstanding LDS / scalar loads.  -DVSR_PHASE_CLOCKS=0 builds without them.
#ifndef VSR_PHASE_CLOCKS    // round 6: off in the product build (-0.6 %: twelve s_memtime per tile and 16 SGPRs); tools/phase_split.py wants a -DVSR_PHASE_CLOCKS=1 build
#define VSR_PHASE_CLOCKS 0
#endif
#if VSR_PHASE_CLOCKS
#define VSR_CLK() __builtin_readcyclecounter()
#else
#define VSR_CLK() ((u64)0)
#endif
#define VSR_CAND_CAP 2048    // enabled instances per tile the LDS work list can hold
// two-stage enumeration of the enabled instances in k_expand (VSR.tla model): 0 = the full guard for every (record, slot) pair
// Block barriers of k_expand.  __syncthreads() is "s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier": every barrier also waits until the
// wave's outstanding GLOBAL stores have been acknowledged — after the apply phase that is the drain of ~28 scattered stores per
// new state.  Nothing in k_expand hands data from wave to wave through global memory inside a launch (the waves of a block talk
// through LDS, blocks through atomics), so the barriers only have to order LDS: wait for the wave's LDS operations, then s_barrier.
// The stores keep draining underneath the next tile's staging loads.
// -DVSR_WAVE_DIAG=1 (diagnostic build, tools/wave_diag.py): every wave of k_expand adds up the shader clocks it spends INSIDE the block barriers; the epilogue
// reports, per wave index 0..3, the barrier time in phase_cycles[0..3] and the wave's whole residency in phase_cycles[4..7] (instead of the phase split).
// =2: only the barrier that closes the apply loop (the wait for the block's slowest wave of the apply phase) is counted.
#ifndef VSR_WAVE_DIAG
#define VSR_WAVE_DIAG 0
#endif
#ifndef VSR_TAKE
#define VSR_TAKE 0
#endif
#ifndef VSR_SPEC_CAS
#define VSR_SPEC_CAS 0
#endif
#ifndef VSR_DIRECT_REFS     // every thread fetches the refs of the four records it stages itself (no barrier between the ref load and the record loads)
#define VSR_DIRECT_REFS 1   // (round 6, with the next switch: config 2 k_expand 130.7 -> 129.5 ms, README 1 091 -> 1 078 ms per step; 0: refs through LDS, as in rounds 1-5)
#endif
#ifndef VSR_NO_TAIL_SYNC    // no barrier at the bottom of the tile loop (what follows the apply-closing barrier touches LDS words of wave 0 only)
#define VSR_NO_TAIL_SYNC 1
#endif
#ifndef VSR_COPY8
#define VSR_COPY8 0
#endif
#ifndef VSR_TILE128         // EXPERIMENT: tiles of 128 records (three blocks per CU) for the ordinary levels of the specialised instantiations too: the fixed cost per tile over twice the records
#define VSR_TILE128 0
#endif
#ifndef VSR_TAKE_BATCH      // -DVSR_TAKE: records per draw from the cursor
#define VSR_TAKE_BATCH 512
#endif
#ifndef VSR_TILE_BATCH      // tiles a block draws from the cursor at a time (k_expand: s_tile_left); 1 = rounds 1-5
#define VSR_TILE_BATCH 4
#endif
#ifndef VSR_REDO            // a tile that overflows the work list is taken again in pieces (k_expand: s_redo_*); 0: ERR_FRONTIER_FULL as in rounds 1-5 (A/B; the experiments below need 0)
#define VSR_REDO 1
#endif
#ifndef VSR_REFS_AHEAD
#define VSR_REFS_AHEAD 0
#endif
#if (VSR_REFS_AHEAD || VSR_TAKE) && VSR_REDO
#undef VSR_REDO
#define VSR_REDO 0
#endif
#ifndef VSR_COOP_COPY       // wave-cooperative copy of the parent words of new states in the ordinary level's instantiation (see the apply loop); 0: the lane-serial copy everywhere
#define VSR_COOP_COPY 1
#endif

#ifndef VSR_ROUND_REV       // EXPERIMENT: the second round of the apply loop runs on the block's LAST waves (wave 0 carries the serial sections already)
#define VSR_ROUND_REV 0
#endif
// =3: wave 1's waits per barrier GROUP (0 top / bottom of the tile loop, 1 after the ref load, 2 after staging, 3 after the work-list fill + parent fingerprints,
// 4 inside the enumeration, 5 after it, 6 sort / reservations, 7 the barrier that closes the apply loop) in phase_cycles[0..7], its residency in act_generated[0].
#if VSR_WAVE_DIAG == 3
#define VSR_SYNC_G(g) do { const u64 b0_ = __builtin_readcyclecounter(); __syncthreads(); if (tid == 64) s_wd[g] += __builtin_readcyclecounter() - b0_; } while (0)
#elif VSR_WAVE_DIAG
#define VSR_SYNC_G(g) do { const u64 b0_ = __builtin_readcyclecounter(); __syncthreads(); if (VSR_WAVE_DIAG == 1 || (g) == 7) wd_wait += __builtin_readcyclecounter() - b0_; } while (0)
#else
#define VSR_SYNC_G(g) __syncthreads()
#endif
// seen-set probes read the home slot (16 B) first and the rest of its 64-byte line only when that slot holds another fingerprint
// successor write: parent words copied eight per trip (four LDS reads in flight) instead of two
// intra-tile duplicate filter in LDS ahead of the seen-set probes of k_expand (single-pass levels)
#ifndef VSR_OCC            // resident blocks per CU the specialised fused kernels are compiled for (4 = 128 VGPRs; 5 = 96: experiment)
#define VSR_OCC 4
#endif
#ifndef VSR_PROBE_FOOTPRINT   // probe level: only the actions that write what the invariants read are applied (Ops::probe_actions)
#define VSR_PROBE_FOOTPRINT 1
#endif
// frontier refs of a tile are loaded one tile ahead (tiles are drawn two ahead): the staging of a tile starts with its record loads

__device__ __forceinline__ void raise_error(LevelCtl* ctl, int code, u64 info) {
  if (atomicCAS(&ctl->err, 0u, (u32)code) == 0u) ctl->err_info = info;
}

__device__ __forceinline__ void raise_full(LevelCtl* ctl, u64 info) {   // called by one thread of a block; a plain flag, nothing is lost (see LevelCtl::full)
  if (ctl->full == 0u) {
    ctl->full_info = info;
    ctl->full = 1u;
  }
}

// ---- the generator-side winner set of a SHARDED deep search (round 5) -------------------------------------------------------------------------
// Records stay with the rank that generated them, so a level that exists in the seen-sets only has to be REGENERATED by its generators.  Rounds 3-4
// asked the OWNER every time: every candidate of every regenerating pass crossed the fabric with its key, the owner granted the one whose key is the
// slot's final meta word, a verdict byte came back (45.6 GB per rank and step on the README configuration at world 2).  But the generator has been
// told once already which of its candidates made a state: the verdict of the pass that INSERTED the level (first inserter wins, k_claim_batch_fused;
// the lane's own compare-and-swap for the states it owns).  It keeps those fingerprints here — an open-addressing set of its own, 12 B per slot —
// and a regenerating pass asks this set instead of the owner: no announcement, no exchange, no seen-set access.  A slot knows its state's LEVEL
// (a successor of a level-l state may be another level-l state of the set, which must not be rebuilt as a member of level l+1), and `epoch` makes
// the take exactly-once per descent (two instances of one rank can yield the same state): the first lane to raise a slot's epoch to the descent's
// rebuilds the state; nothing is cleared between descents.
struct WSet {
  u64* fp;          // 0 = empty
  u32* epoch;       // level(9) << 23 | the last descent that regenerated the state (23 bits)
  u64 mask;         // slots - 1
};
__device__ __forceinline__ u64 wset_home(u64 fp, u64 mask) { return (fp >> 13) & mask; }   // (other bits than the seen-set's index: fp & tmask)
__device__ __forceinline__ bool wset_insert(const WSet* w, u64 fp, int level) {            // false: the set is full.  (A state is inserted once in a
  const u64 mask = w->mask;                                                                // run — by the one candidate that made it — and never while
  u64 i = wset_home(fp, mask);                                                             // a regenerating pass reads the set: different launches.)
  for (u64 step = 0; step <= mask && step < 65536; step++, i = (i + 1) & mask) {
    const u64 cur = atomicCAS((unsigned long long*)&w->fp[i], 0ull, (unsigned long long)fp);
    if (cur == 0) { w->epoch[i] = (u32)level << 23; return true; }
    if (cur == fp) return true;
  }
  return false;
}
// is fp a level-`level` state of mine, and am I the first of this descent to ask?
__device__ __forceinline__ bool wset_take(const WSet* w, u64 fp, int level, u32 epoch) {
  const u64 mask = w->mask;
  u64 i = wset_home(fp, mask);
  for (u64 step = 0; step <= mask && step < 65536; step++, i = (i + 1) & mask) {
    const u64 cur = w->fp[i];
    if (cur == fp) {
      if ((w->epoch[i] >> 23) != (u32)level) return false;
      const u32 mine = ((u32)level << 23) | (epoch & 0x7FFFFFu);
      return atomicMax(&w->epoch[i], mine) < mine;
    }
    if (cur == 0) return false;
  }
  return false;
}

__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, l);
  u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}

// One atomic per wave: returns base + rank of this lane among the lanes that call it (all callers must pass the
// same counter).  Lanes call this from inside a divergent branch; the ballot only sees the active ones.
__device__ __forceinline__ u64 wave_alloc(u64* counter) {
  u64 active = __ballot(1);
  int lane = lane_id();
  int leader = __ffsll((long long)active) - 1;
  u64 base = 0;
  if (lane == leader) base = atomicAdd((unsigned long long*)counter, (unsigned long long)__popcll(active));
  base = __shfl(base, leader);
  return base + (u64)__popcll(active & (((u64)1 << lane) - 1));
}

// Find-or-insert of a fingerprint: linear probing from fp & mask, one 64-byte line (4 slots) per memory round trip.
// A loaded line may be stale with respect to concurrent inserts, which is harmless: a slot only ever goes empty -> fp and
// never changes afterwards, so "other key" and "this key" are final, and "empty" is re-checked by the atomicCAS.
// how a probe sequence is counted (vsrmc_level_info.probes, informational): in a register of the lane (small kernels), or — in k_expand, where
// every register of the apply loop counts — not at all for the home slot (the caller adds one per candidate) and with an LDS atomic per
// further slot (one candidate in ten gets that far)
struct CntReg {
  u32* p;
  __device__ __forceinline__ void home() const { (*p)++; }
  __device__ __forceinline__ void extra() const { (*p)++; }
};
struct CntLds {
  unsigned long long* p;
  __device__ __forceinline__ void home() const {}
  __device__ __forceinline__ void extra() const { atomicAdd(p, 1ull); }
};
struct Probe {
  u64 slot;         // index of the slot that holds fp
  u64 meta;         // its meta word as loaded with the line (valid unless `claimed` or `reload`)
  bool claimed;     // this lane inserted fp
  bool reload;      // fp was inserted by someone else between the load and our CAS: meta must be re-read
  bool full;
};
template <typename CNT>
__device__ __forceinline__ Probe probe_insert(Slot* table, u64 mask, u64 fp, CNT nprobe) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  Probe r;
  r.slot = 0; r.meta = META_EMPTY; r.claimed = false; r.reload = false; r.full = false;
  u64 i = fp & mask;
#if VSR_SPEC_CAS
  // EXPERIMENT: the compare-and-swap of the home slot is issued TOGETHER with its 16-byte load (a new state then costs one memory round trip, not two
  // dependent ones); an occupied slot makes the atomic a no-op that returns what the slot holds
  {
    const u64 old = atomicCAS((unsigned long long*)&table[i].fp, 0ull, (unsigned long long)fp);
    const u64x2 sk = *(const u64x2*)&table[i];
    nprobe.home();
    if (old == 0) { r.slot = i; r.claimed = true; return r; }
    if (old == fp) {
      r.slot = i;
      if (sk.x == fp) r.meta = sk.y; else r.reload = true;
      return r;
    }
    i = (i + 1) & mask;
  }
#else
  {
    const u64x2 sk = *(const u64x2*)&table[i];
    nprobe.home();
    u64 cur = sk.x;
    if (cur == 0) {
      cur = atomicCAS((unsigned long long*)&table[i].fp, 0ull, (unsigned long long)fp);
      if (cur == 0) { r.slot = i; r.claimed = true; return r; }
      if (cur == fp) { r.slot = i; r.reload = true; return r; }
    } else if (cur == fp) {
      r.slot = i;
      r.meta = sk.y;
      return r;
    }
    i = (i + 1) & mask;
  }
#endif
  for (u32 lines = 0; lines < 2048; lines++) {
    const u64 lb = i & ~(u64)3;
    const u64x2* lp = (const u64x2*)&table[lb];
    const u64x2 s0 = lp[0], s1 = lp[1], s2 = lp[2], s3 = lp[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (lb + k < i) continue;                                 // slots of the line before the probe start
      const u64x2 sk = k == 0 ? s0 : k == 1 ? s1 : k == 2 ? s2 : s3;
      nprobe.extra();
      u64 cur = sk.x;
      if (cur == 0) {
        cur = atomicCAS((unsigned long long*)&table[lb + k].fp, 0ull, (unsigned long long)fp);
        if (cur == 0) {
          r.slot = lb + k;
          r.claimed = true;
          return r;
        }
        if (cur == fp) {
          r.slot = lb + k;
          r.reload = true;
          return r;
        }
        continue;                                               // another fingerprint took the slot meanwhile
      }
      if (cur == fp) {
        r.slot = lb + k;
        r.meta = sk.y;
        return r;
      }
    }
    i = (lb + 4) & mask;
  }
  r.full = true;
  return r;
}

// Lookup without insertion (probe level, vsrmc_checker_probe): is fp in the table, and with which meta word?
template <typename CNT>
__device__ __forceinline__ bool probe_lookup(const Slot* table, u64 mask, u64 fp, u64* meta, CNT nprobe, u64* slot_out = nullptr) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  u64 i = fp & mask;
  {
    const u64x2 sk = *(const u64x2*)&table[i];
    nprobe.home();
    if (sk.x == fp) {
      *meta = sk.y;
      if (slot_out) *slot_out = i;
      return true;
    }
    if (sk.x == 0) return false;
    i = (i + 1) & mask;
  }
  for (u32 lines = 0; lines < 2048; lines++) {
    const u64 lb = i & ~(u64)3;
    const u64x2* lp = (const u64x2*)&table[lb];
    const u64x2 s0 = lp[0], s1 = lp[1], s2 = lp[2], s3 = lp[3];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      if (lb + k < i) continue;
      const u64x2 sk = k == 0 ? s0 : k == 1 ? s1 : k == 2 ? s2 : s3;
      nprobe.extra();
      if (sk.x == fp) {
        *meta = sk.y;
        if (slot_out) *slot_out = lb + k;
        return true;
      }
      if (sk.x == 0) return false;
    }
    i = (lb + 4) & mask;
  }
  return false;
}

// Seen-set claim of the two-kernel scheme.  Returns the slot index; *found_old = true when the candidate cannot win (the
// fingerprint belongs to an earlier level, or a smaller key of this level already holds the slot), otherwise the caller's key
// has been min-merged into the slot's meta word and the candidate goes to the pending list.
template <typename CNT>
__device__ __forceinline__ u64 table_claim(Slot* table, u64 mask, u64 fp, u64 key, int level, bool* found_old,
                                           CNT nprobe, bool* full) {
  const Probe p = probe_insert(table, mask, fp, nprobe);
  *full = p.full;
  *found_old = true;
  if (p.full) return 0;
  if (!p.claimed) {
    const u64 m = p.reload ? __hip_atomic_load(&table[p.slot].meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.meta;
    if (meta_level(m) < level || m < key) return p.slot;       // earlier level, or beaten already (meta only ever decreases)
  }
  const u64 prev = atomicMin((unsigned long long*)&table[p.slot].meta, (unsigned long long)key);
  *found_old = prev < key;                                     // lost the race after all
  return p.slot;
}

// Fused variant (single-pass BFS level): *claimed = this lane inserted the fingerprint (it writes the successor at once);
// *prev_meta = the meta word this candidate displaced / found (META_EMPTY if none): the caller compares auxkeys AFTER it
// has done its other work, so that the returning atomic's latency overlaps with the successor write.  A same-level
// duplicate with a different canonical auxkey is the VIEW collision of SURVEY F2 inside one level, which the single-pass
// scheme cannot arbitrate — the host then asks for the exact two-kernel scheme (never observed: `ties` is 0 everywhere).
template <typename CNT>
__device__ __forceinline__ void table_claim_fused(Slot* table, u64 mask, u64 fp, u64 key, int level, bool* claimed, u64* prev_meta,
                                                  CNT nprobe, bool* full) {
  const Probe p = probe_insert(table, mask, fp, nprobe);
  *full = p.full;
  *claimed = p.claimed;
  *prev_meta = META_EMPTY;
  if (p.full) return;
  if (!p.claimed) {
    const u64 m = p.reload ? __hip_atomic_load(&table[p.slot].meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : p.meta;
    if (meta_level(m) < level) return;                         // a state of an earlier level
    if (m != META_EMPTY && m < key) {                          // smaller key of this level already there
      *prev_meta = m;
      return;
    }
  }
  *prev_meta = atomicMin((unsigned long long*)&table[p.slot].meta, (unsigned long long)key);
}

// -----------------------------------------------------------------------------------------------------------------
// k_expand
// -----------------------------------------------------------------------------------------------------------------
// Overwrite the structural constants of a private copy of the model with the compile-time constants of SPEC (see k_expand).
template <int SPEC>
__device__ __forceinline__ void specialise(Model& M, const Model& Marg) {
  constexpr int MODEL = SPEC / 1000, S = SPEC % 1000;
  if constexpr (S != 0) {
    constexpr int SR = S / 100, SC = (S / 10) % 10, SN = S % 10;
    M.R = SR;
    M.C = SC;
    M.n = SN;
    if constexpr (MODEL == 0) {
      M.wpr = 1 + (SR + 2) / 2;
      M.m0 = 4 * SR + SR * SC * SN;
      if (Marg.np == 1) M.np = 1;                              // symmetry off: block-uniform, still cheap
      else M.np = SN == 1 ? 1 : SN == 2 ? 2 : 6;
    } else {                                                   // the second / third model: one / two words per replica, no clients, no symmetry
      M.wpr = MODEL == 1 ? 1 : 2;
      M.m0 = 4 * SR + SR * SN;
      M.np = 1;
    }
    M.h0 = 1 + SR * M.wpr;
    M.fixed = M.h0 + M.np + VSR_PAD_WORDS;
    // the permutation table in build_model's order (identity first): compile-time constants, so that permute_word's select
    // masks fold (np == 1 only ever looks at entry 0)
    M.pitab[0] = 0x24u;
    if (SN == 2) M.pitab[1] = 0x21u;
    if (SN == 3) { M.pitab[1] = 0x18u; M.pitab[2] = 0x21u; M.pitab[3] = 0x09u; M.pitab[4] = 0x12u; M.pitab[5] = 0x06u; }
  }
}

// SPEC = model * 1000 + R * 100 + C * 10 + |Values| of a configuration the kernel is specialised for (S = SPEC % 1000 = 0: generic
// in the constants, still specific to the model): the constants of the model
// become compile-time constants of this instantiation (every device function below is inlined), so loops over replicas,
// clients, values and permutations unroll without predicates and strides fold into addresses.
// BLK = threads per block = 256: four waves share a tile of 64 (or 128) records, block barriers between the phases.  (One wave per block
// with a 16-record tile of its own and 512-thread blocks were built and measured in round 3: 268 and 212 ms against 156 — DESIGN.md §5.)
// Resident blocks per CU an instantiation is compiled for (= its register budget: 4 -> 128 VGPRs, 5 -> 96).  Round 6: with MachineLICM off (build.py) the
// ordinary level's kernel of BASELINE configs[1] needs 107 registers and runs FIVE blocks per CU with 7 spilled ones at the top of the tile loop (k_expand
// 139 -> 134.6 ms; with the cooperative copy 127.9); the README configuration's (six permutations: 119 registers) loses at five (28 spilled: 232 -> 257 ms).
constexpr int expand_occ(bool fused, int spec, int plain) {
  return !fused ? 3 : spec == 0 ? 2 : (plain == 5 || plain == 6) ? VSR_OCC + 1 : VSR_OCC;
}
template <bool FUSED, int SPEC = 0, int PLAIN = 0, int BLK = VSR_BLOCK>
// (hipcc turns the second bound into waves per SIMD as blocks * max(1, threads / 256): 4 = 128 VGPRs)
__global__ void __launch_bounds__(BLK, expand_occ(FUSED, SPEC, PLAIN) / (BLK > VSR_BLOCK ? BLK / VSR_BLOCK : 1))
k_expand(Model Marg, const u64* __restrict__ fr_words, const u64* __restrict__ fr_off, u64 n_parents, int level, int rank,
         Slot* table, u64 tmask, u64* pending, u64 pending_cap, LevelCtl* ctl, int stride, int world_arg, u64* cand_send,
         u64 cand_cap, u32 pchunk /* pending entries a block reserves per global atomic, >= ccap */,
         // fused single-pass mode (nx_words != nullptr): the lane that inserts a fingerprint writes the successor at once
         u64* nx_words, u64 nx_words_cap, u64* nx_off, u64 nx_cap, u64* lvl_fp, u32 ichunk, u32 wchunk, int tile, u32 ccap /* work-list capacity per tile */,
         // fused + sharded (world > 1): successors owned by another rank pass the rank's sent-filter, are written to the local
         // next frontier SPECULATIVELY and announced to their owner; cand_idx remembers where, for k_apply_verdict
         u64* filter, u64 fmask, u64* cand_idx, u32 cchunk /* candidate entries a block reserves per owner and global atomic */,
         // fused, unsharded passes that do not materialise a level the normal way (vsrmc_checker_probe / _probe2):
         //   MODE_PROBE   nothing is inserted or written; every successor that is not a state of an EARLIER level gets its
         //                invariants checked — evaluated the cheap way round: only the actions inside the invariants' footprint are
         //                applied (Ops::probe_actions), their successors' invariants come first, and only a failing successor is
         //                fingerprinted and looked up
         //   MODE_INSERT  "virtual level": fingerprints are claimed (min-merged keys, as always) and the invariants of the
         //                new states checked, but no record, ref or trace key is written; ctl->n_new = exact number of new states
         //   MODE_REGEN   re-expansion of (a slice of) the same parents after a MODE_INSERT pass: the successor whose key IS
         //                the slot's final meta word — exactly one per new state — is written to the next frontier
         // violators of PROBE / INSERT go to the `pending` list as (fp, key) pairs (n_pending counts them).
         int mode_arg, u64 p_offset /* index of parent 0 of this launch in its level (slices) */,
         // sharded deep search (world > 1, a pass beyond the record buffers): the rank's winner set — MODE_INSERT / MODE_NORMAL record the states this
         // rank's own lanes insert, MODE_REGEN rebuilds exactly the states the set holds, once per descent (wepoch), and announces nothing
         const WSet* wset, u32 wepoch) {
  // PLAIN: the unsharded, ordinary level — the probe / virtual-level modes and the sharded branches are compiled out (11 % less
  // code: the specialised kernel then fits the 64-KB instruction cache with room to spare)
  // PLAIN == 2: unsharded with the modes (the probe / virtual / regenerated / streamed passes of vsrmc_checker_probe*): only the
  // sharded branches are compiled out — the mode-capable kernel of the README configuration then fits the 64-KB instruction cache
  // PLAIN == 3 / 4 (round 5): unsharded with ONE mode compiled in — 3 = MODE_REGEN by the claim bitmap (no guards and no seen-set code at all), 4 =
  // MODE_INSERT (a virtual level: no successor write; leaves the claim bitmap).  The mode-capable instantiation carries every mode behind a run-time switch
  // and pays for it in registers (its README build spills 27 VGPRs; one more branch took that to 93, and every pass 10 % with it): a pass that knows its mode
  // runs leaner code.  (A probe-only instantiation was built too and lost to the run-time-switched one — 88 spilled VGPRs: DESIGN.md §8.5.)
  constexpr bool IS_PLAIN = PLAIN == 1 || PLAIN == 5;          // the ordinary level's instantiation (5: the same compiled for five blocks per CU)
  // PLAIN == 6 (round 6): the PROBE pass of the deep search and nothing else, compiled for FIVE blocks per CU.  A probe pass stages and enumerates every
  // parent but applies 2 % of the instances (the actions inside the invariants' footprint) and fingerprints only a successor that FAILS an invariant — 8 of
  // 3.8e9 on the README configuration.  Here that rare successor is not fingerprinted at all: its (parent, ordinal) goes to the `pending` list and a small
  // kernel of its own (k_probe_resolve) hashes it and looks it up after the pass; and the instances outside the footprint are counted, not listed.  Without
  // the hash of six permutations and the seen-set code the kernel fits 96 registers, and with a work list of 256 entries its tile fits the LDS five times.
  const int mode = IS_PLAIN ? (int)MODE_NORMAL : PLAIN == 3 ? (int)MODE_REGEN : PLAIN == 4 ? (int)MODE_INSERT : PLAIN == 6 ? (int)MODE_PROBE : (mode_arg & 0xFF);
  // MODE_NO_FOOTPRINT: the probe pass applies every action — the caller has seen a violating state among the parents' levels (a search that
  // went on after a reported violation), and a violating parent hands its verdict to successors of actions outside the footprint
  const bool no_footprint = !IS_PLAIN && PLAIN != 6 && (mode_arg & MODE_NO_FOOTPRINT) != 0;
  const int world = PLAIN ? 1 : world_arg;
  Model M = Marg;
  specialise<SPEC>(M, Marg);
  typedef ModelOps<SPEC / 1000> Ops;
  extern __shared__ u64 smem[];
  u64* s_rec = smem;                                           // tile * stride words
  u32* s_cand = (u32*)(smem + tile * stride);                  // ccap entries: action << 18 | record << 11 | ordinal
  u32* s_cand2 = s_cand + ccap;                        // the same, sorted by action
  __shared__ u32 s_ncand, s_napply, s_dead, s_maxbag, s_maxbag_out, s_skip, s_nsurv, s_risky, s_ntotal;
  // (single-pass levels of a configuration with R <= 3 always run 64-record tiles: host_checker.hpp, fused_shape — 1.3 KB of LDS less, which five blocks per CU need)
  constexpr int TILE_MAX = BLK < VSR_BLOCK ? BLK / 2 : (!VSR_TILE128 && FUSED && SPEC % 1000 != 0 && (SPEC % 1000) / 100 <= 3) ? 64 : VSR_TILE_MAX;
  constexpr bool COOP = VSR_COOP_COPY && FUSED && IS_PLAIN;
  constexpr bool REDO_OK = VSR_REDO && FUSED && (IS_PLAIN || PLAIN == 6);   // a tile that overflows the work list goes to the host's list (LevelCtl::n_redo) instead of failing the launch
  __shared__ u32 s_alive[TILE_MAX];
  __shared__ u64 s_ref[TILE_MAX];
  __shared__ u64 s_pfp[TILE_MAX];                          // canonical fingerprint of every staged record (the parent part of its successors' keys)
  __shared__ u32 s_kcount[16], s_kbase[16];
  __shared__ int s_slotinfo[64];                               // decode of the replica-bound slots (m0 <= 50), see slot_info()
  // per-block accumulators (flushed once at the end: no hot global counters inside the tile loop)
  __shared__ unsigned long long s_fxs[2];                      // virtual level: xor / sum of the fingerprints this block inserted (flushed once, in the epilogue)
  __shared__ unsigned long long s_acc[32];                     // 0 generated, 1 deadlocks, 2 probes, 3..7 phase cycles, 16..31 per action
  // pending list: the block owns a chunk of pchunk entries at a time; unused entries are invalidated (key = ~0)
  __shared__ u64 s_chunk_base;
  __shared__ u32 s_chunk_used, s_tile_base, s_tile_cursor;
  // fused mode: the block's chunks of next-frontier indices and words
  __shared__ u64 s_ich_base, s_wch_base;
  __shared__ u32 s_ich_used, s_wch_used, s_tile_ibase, s_tile_wbase, s_tile_icur, s_tile_wcur, s_wneed;
  // fused + sharded: the block's chunk of each owner's candidate bucket, (chunk base << 24 | entries used) in ONE word so
  // that a lane's atomicAdd sees a consistent pair; "used == cchunk" = no room, the lane that draws it fetches a new chunk
  __shared__ unsigned long long s_cstate[8];
  constexpr bool fused = FUSED;
  constexpr u64 CS_NONE = ((u64)1 << 40) - 1;                   // chunk base of "no chunk yet"

  const int tid = threadIdx.x, lane = tid & 63;
#if VSR_WAVE_DIAG
  u64 wd_wait = 0;
  const u64 wd_t0 = __builtin_readcyclecounter();
  __shared__ unsigned long long s_wd[8];
  if (tid < 8) s_wd[tid] = 0;
#endif
  const u64 ntiles = (n_parents + tile - 1) / tile;
  const int tshift = 31 - __clz(tile);                           // tile is a power of two
  if (tid < 32) s_acc[tid] = 0;
  if (tid < 2) s_fxs[tid] = 0;
  if (tid == 0) s_maxbag_out = 0;
  if (tid < 64) s_slotinfo[tid] = (SPEC / 1000 == 0 && tid < M.m0) ? slot_info(M, tid) : 0;
  if (tid == 0) {                                              // "no chunk yet"
    s_chunk_base = 0; s_chunk_used = pchunk;
    s_ich_base = 0; s_ich_used = ichunk; s_wch_base = 0; s_wch_used = wchunk;
  }
  if (tid < 8) s_cstate[tid] = (CS_NONE << 24) | cchunk;
  VSR_SYNC_G(0);

  // blockIdx -> tiles: the persistent blocks draw tiles from an atomic counter (ctl->tile_cursor, zeroed by the host with the other
  // level counters).  Tile costs are uneven and correlated along the frontier (successors per record, bag sizes), so the static
  // mapping tile = blockIdx + k * gridDim left blocks idle at the end of every launch: 203.8 -> 177.6 ms per run of the benchmark
  // workload.  Thread 0 draws one tile ahead, so the atomic's latency hides behind the current tile.  The order in
  // which tiles are taken changes the order of the records in the next frontier, nothing else: counts, fingerprint sets and the
  // min-merged predecessor keys do not depend on it.
  __shared__ u64 s_tile_cur;
  u64 my_next = 0;
#if VSR_TAKE
  // EXPERIMENT (-DVSR_TAKE=<candidates>): the apply loop runs in rounds of BLK lanes and a tile of 64 records yields 320 (config 2) .. 416 (README)
  // instances: the second round is a quarter / two thirds full and three waves wait for the one that runs it.  Here a tile is as many RECORDS as are
  // expected to yield VSR_TAKE instances (smoothed instances per record of the block's own tiles); the cursor counts records, not tiles.
  __shared__ u32 s_tile_n;
  u32 my_take = (u32)tile, my_n = 0, avg_q8 = 0;                // thread 0: records to draw next time / drawn with my_next / instances per record x 256
  // (the cursor is drawn VSR_TAKE_BATCH records at a time — see VSR_TILE_BATCH below — and the block cuts its batch into tiles)
  __shared__ u64 s_take_end;                                     // end of the drawn batch (thread 0's)
  if (tid == 0) {
    my_next = atomicAdd((unsigned long long*)&ctl->tile_cursor, (unsigned long long)VSR_TAKE_BATCH);
    s_take_end = my_next + VSR_TAKE_BATCH;
    my_n = my_take;
  }
#elif VSR_REFS_AHEAD
  // EXPERIMENT: the refs of a tile are fetched while the tile BEFORE it is staged (tiles are drawn two ahead), so staging starts with the record loads —
  // one HBM round trip per tile instead of two dependent ones
  __shared__ u64 s_tile_nxt;
  __shared__ u64 s_ref2[TILE_MAX];
  u64 my_next2 = 0;
  if (tid == 0) {
    my_next = atomicAdd((unsigned long long*)&ctl->tile_cursor, 1ull);
    my_next2 = atomicAdd((unsigned long long*)&ctl->tile_cursor, 1ull);
    s_tile_nxt = my_next;
  }
  VSR_SYNC_G(0);
  if (tid < tile) {
    const u64 i0 = s_tile_nxt * (u64)tile + (u64)tid;
    s_ref2[tid] = (s_tile_nxt < ntiles && i0 < n_parents) ? fr_off[i0] : 0;
  }
  VSR_SYNC_G(0);                                                // (thread 0 rewrites s_tile_nxt at the top of the loop)
#else
  // Round 6: VSR_TILE_BATCH consecutive tiles per draw.  Atomics on ONE address are served one after the other — 13 - 15 ns each on this part (the staging
  // micro-benchmark: 65 - 75 tiles per microsecond whatever the layout, the occupancy or the prefetch depth, tools/bench_layout.py) — and a pass that only
  // stages and enumerates (the probe pass: 65 tiles per microsecond) ran AT that rate: the cursor, not the HBM, was its bound.
  __shared__ u32 s_tile_left;                                    // tiles left of the drawn batch after my_next (thread 0's)
  const u32 tbatch = ntiles >= (u64)gridDim.x * 16 ? (u32)VSR_TILE_BATCH : 1u;   // (the same in every block of the launch; a small level keeps single tiles: every block gets work)
  if (tid == 0) { my_next = atomicAdd((unsigned long long*)&ctl->tile_cursor, 1ull) * (u64)tbatch; s_tile_left = tbatch - 1; }
#endif
  // A tile whose enabled instances do not fit the work list (ccap entries: 12 .. 24 per record, the mean is 5 - 7) is not an error for the ordinary level's
  // instantiations since round 6: nothing of it has been applied when the counting sort finds out, so the block writes the tile down (its first record and
  // its size, in the `pending` list an ordinary single-pass level has no other use for) and goes on; the host launches the listed tiles again in halves
  // (host_checker.hpp: redo_overflowed_tiles).  Only a SINGLE record with more instances than the list holds is ERR_FRONTIER_FULL.  This is what lets the
  // launch shape shorten the list until five blocks fit a CU (fused_shape) without betting correctness on it.
  __shared__ u32 s_over;
  if (tid == 0) s_over = 0;
  for (;;) {
    if (tid == 0) {
      s_tile_cur = my_next;
#if VSR_REFS_AHEAD
      s_tile_nxt = my_next2;
      my_next = my_next2;
      if (my_next2 < ntiles) my_next2 = atomicAdd((unsigned long long*)&ctl->tile_cursor, 1ull);
    }
    if (0) {
#endif
#if VSR_TAKE
      s_tile_n = my_n;
      if (my_next < n_parents) {
        const u64 end = s_take_end;
        my_next += my_n;
        if (my_next >= end) { my_next = atomicAdd((unsigned long long*)&ctl->tile_cursor, (unsigned long long)VSR_TAKE_BATCH); s_take_end = my_next + VSR_TAKE_BATCH; my_n = my_take; }
        else { const u64 room = end - my_next; my_n = (room < (u64)my_take + 8 && room <= (u64)tile) ? (u32)room : my_take; }   // (a remainder of fewer than 8 records rides with the last tile)
      }
#else
      if (my_next < ntiles) {
        const u32 left = s_tile_left;
        if (left) { my_next++; s_tile_left = left - 1; }
        else { my_next = atomicAdd((unsigned long long*)&ctl->tile_cursor, 1ull) * (u64)tbatch; s_tile_left = tbatch - 1; }
      }
#endif
    }
    const u64 t_0 = VSR_CLK();
    if (tid == 0) { s_ncand = 0; s_dead = 0; s_maxbag = 0; s_wneed = 0; s_skip = 0; s_nsurv = 0; s_over = 0; if (!IS_PLAIN) s_risky = 0; }
    if (tid < tile) s_alive[tid] = 0;
    if (tid < 16) s_kcount[tid] = 0;
    VSR_SYNC_G(0);
    u64 tile_i = s_tile_cur;
    tile_i = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(tile_i >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)tile_i);   // (block-uniform: scalar)
#if VSR_TAKE
    if (tile_i >= n_parents) break;
    const u64 p_base = tile_i;
    const int np_tile = (int)((n_parents - p_base) < (u64)s_tile_n ? (n_parents - p_base) : (u64)s_tile_n);
#else
    if (tile_i >= ntiles) break;
    u64 p_base = tile_i * (u64)tile;
    int np_tile = (int)((n_parents - p_base) < (u64)tile ? (n_parents - p_base) : (u64)tile);
    if constexpr (REDO_OK) {
      // a launch over the LIST of tiles an earlier launch could not fit into its work list (MODE_REDO_LIST: p_offset = the list, n_parents = 2 x entries x
      // tile): tile 2 k / 2 k + 1 = the two halves of entry k
      if (mode_arg & MODE_REDO_LIST) {
        const u64 e = ((const u64*)p_offset)[tile_i >> 1];
        const int cnt = (int)(e >> 48);
        int half = 1;
        while (half * 2 < cnt) half *= 2;
        p_base = (e & (((u64)1 << 48) - 1)) + ((tile_i & 1) ? (u64)half : 0);
        np_tile = (tile_i & 1) ? cnt - half : half;
      }
      // (block-uniform values that came out of LDS / memory: into scalar registers, not three VGPRs live across the whole tile)
      p_base = ((u64)(u32)__builtin_amdgcn_readfirstlane((int)(u32)(p_base >> 32)) << 32) | (u32)__builtin_amdgcn_readfirstlane((int)(u32)p_base);
      np_tile = __builtin_amdgcn_readfirstlane(np_tile);
    }
#endif
    // PLAIN == 3 (regeneration by the claim bitmap): the bitmap IS the list of enabled instances that matter — the (parent, ordinal) pairs whose lane
    // made a state when the level was inserted.  The words of this thread's record (thread g of the record's G threads takes words g and g + G; the host
    // offers the bitmap only when 2 G words cover a parent) are fetched now, so that their latency hides behind the staging loads.
    u32 cbits[2] = {0u, 0u};
    if constexpr (PLAIN == 3) {
      const int pm = tid & (tile - 1), gg = tid >> tshift, GG = BLK >> tshift;
      if (pm < np_tile) {
        const u32* row = (const u32*)filter + (p_offset + p_base + (u64)pm) * fmask;
#pragma unroll
        for (int q = 0; q < 2; q++)
          if ((u64)(gg + q * GG) < fmask) cbits[q] = row[gg + q * GG];
      }
    }

    // ---- stage the tile.  Frontier refs are (word offset << 8 | length): one coalesced load of 64 refs, then 16 lanes per
    // record / 16 records per pass, all 16 loads of a thread issued before the first LDS store (one HBM latency per tile)
#if VSR_REFS_AHEAD
    u64 ref_pre = 0;                                            // wave 0: the refs of the NEXT tile, on their way while this one is staged
    if (tid < tile) {
      const u64 i1 = s_tile_nxt * (u64)tile + (u64)tid;
      if (s_tile_nxt < ntiles && i1 < n_parents) ref_pre = fr_off[i1];
    }
#endif
    if (tid < np_tile) {
#if VSR_REFS_AHEAD
      const u64 ref = s_ref2[tid];
#else
      const u64 ref = fr_off[p_base + tid];
#endif
      s_ref[tid] = ref;
      if (ref) atomicMax(&s_maxbag, (u32)((int)(ref & 255) - M.fixed));
      if ((int)(ref & 255) > stride) raise_error(ctl, ERR_INTERNAL, (p_base + (u64)tid) << 16);   // LDS slots sized for shorter records
    }
#if !VSR_DIRECT_REFS
    VSR_SYNC_G(1);
#endif
    constexpr int SG = BLK / 16;                                 // records staged per pass (16 lanes each)
    for (int half = 0; half < tile; half += 4 * SG)
      for (int wbase = 0; wbase < stride; wbase += 64) {      // records longer than 64 words (R >= 4): a second window
        u64 v[4][4];
#if VSR_DIRECT_REFS
        u64 refq[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int p = half + (tid >> 4) + SG * q;
          refq[q] = p < np_tile ? fr_off[p_base + (u64)p] : 0;
        }
#endif
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int p = half + (tid >> 4) + SG * q;
#if VSR_DIRECT_REFS
          const u64 ref = refq[q];
#else
          const u64 ref = p < np_tile ? s_ref[p] : 0;
#endif
          const u64 off = ref >> 8;
          const int len = (int)(ref & 255) < stride ? (int)(ref & 255) : stride;
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int k = wbase + (tid & 15) + 16 * j;
            v[q][j] = k < len ? fr_words[off + k] : 0;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int p = half + (tid >> 4) + SG * q;
#if VSR_DIRECT_REFS
          const int len = (int)(refq[q] & 255) < stride ? (int)(refq[q] & 255) : stride;
#else
          const int len = p < np_tile ? ((int)(s_ref[p] & 255) < stride ? (int)(s_ref[p] & 255) : stride) : 0;
#endif
#pragma unroll
          for (int j = 0; j < 4; j++) {
            const int k = wbase + (tid & 15) + 16 * j;
            if (k < len) s_rec[p * stride + k] = v[q][j];
          }
        }
      }
#if VSR_REFS_AHEAD
    if (tid < tile) s_ref2[tid] = ref_pre;                      // (read again at the top of the next tile, two barriers from here)
#endif
    VSR_SYNC_G(2);
    if (tid < np_tile && s_ref[tid] != 0) {                     // the parent's own fingerprint, from the view hashes it carries
      const u64* r0 = s_rec + tid * stride;
      u64 pf;
      u32 pa;
      canonical_fp(M, r0[0], r0 + M.h0, &pf, &pa);
      s_pfp[tid] = pf;
    }

    const u64 t_1 = VSR_CLK();
    // ---- enumerate enabled instances, slot-major: item = slot * 64 + record, so the 64 lanes of a wave evaluate the same
    // slot kind for 64 different records (LDS columns are conflict-free: odd record stride).  Work list entry =
    // action id << 17 | record << 11 | ordinal; per-action counters feed the counting sort below.
    {
      const int nslots = M.m0 + (int)s_maxbag;
      const int nitems = nslots << tshift;
      // A thread always looks at the same record (p = tid mod tile; the block size is a multiple of the tile size): its
      // header and replica A words are read from LDS once, every slot's guard then costs one LDS read (the bag word).
      const int p_mine = tid & (tile - 1);
      const bool mine_valid = p_mine < np_tile && s_ref[p_mine] != 0;
      const u64* rec_mine = s_rec + p_mine * stride;
      u64 Areg[6] = {0, 0, 0, 0, 0, 0};
      u64 hdr_mine = 0;
      if (mine_valid) {
        hdr_mine = rec_mine[0];
#pragma unroll
        for (int r = 1; r <= 5; r++)
          if (r <= M.R) Areg[r] = rec_mine[1 + (r - 1) * M.wpr];
      }
      // Work-list layout during enumeration: thread t owns entries [t*PRIV, t*PRIV + PRIV) — it appends there with a cursor
      // in a register, no atomics, no cross-lane traffic inside the slot loop; the entries beyond 256*PRIV are a shared
      // overflow area (atomic cursor) for the rare thread that finds more.  Unused entries hold ~0; the counting sort below
      // skips them.
      const u32 PRIV = PLAIN == 6 ? 0u : (ccap - (u32)BLK) / BLK;   // 5 at ccap 1536, 7 at 2048; the probe-only instantiation lists 2 % of the instances: one shared region
      const u32 shared0 = PRIV * BLK;
      for (u32 k = tid; k < ccap; k += BLK) s_cand[k] = ~0u;
      VSR_SYNC_G(3);
      u32 nmine = 0;
      bool alive = false;
      if constexpr (SPEC / 1000 == 0) {
        // Two-stage enumeration.  Evaluating the full guard for every (record, slot) pair was more than half of the kernel's
        // instructions although 1 pair in 20 is enabled.  Stage 1, per bag entry: delivery count > 0 and one bit of a 64-bit table
        // per replica (prefilter_lut: the guard's (type, view) conjuncts) — the survivors, (record, entry) pairs, are compacted
        // into a list with one ballot and one LDS atomic per wave.  Stage 2: the exact guard on the list, dense.  The replica-bound
        // instances of a record are evaluated as one bit mask per replica by the record's threads (thread g takes replicas g+1, g+1+G, ..).
        auto emit = [&](int kind, int p, int ord) {
          atomicAdd(&s_kcount[kind], 1u);
          if constexpr (PLAIN == 6) {                               // the probe-only instantiation lists what it will apply; the rest is counted
            if (!((Ops::probe_actions() >> kind) & 1u)) { s_alive[p] = 1; return; }
          }
          u32 idx;
          if (nmine < PRIV) idx = (u32)tid * PRIV + nmine;
          else idx = shared0 + atomicAdd(&s_ncand, 1u);
          nmine++;
          if (idx < ccap) s_cand[idx] = ((u32)kind << 18) | ((u32)p << 11) | (u32)ord;
          else s_ncand = 0x40000000u;                             // overflow marker (work list too small)
          s_alive[p] = 1;
        };
        const int G = BLK >> tshift, g = tid >> tshift;     // threads per record, this thread's rank among them
        if constexpr (PLAIN == 3) {
          // no guards, no bag scans: every set bit is an instance that was enabled (and made a state) when the level was inserted; only its action id has
          // to be found again — by the ordinal's range, or, for a message-bound one, by the one slot's own guard (which also names the action)
          if (mine_valid) {
#pragma unroll
            for (int q = 0; q < 2; q++) {
              u32 bits = cbits[q];
              while (bits) {
                const int b = __ffs((int)bits) - 1;
                bits &= bits - 1;
                const int ord = 32 * (g + q * G) + b;
                int kind;
                if (ord < M.m0) {
                  kind = ord < M.R ? A_TimerSendSVC : ord < 2 * M.R ? A_SendDVC : ord < 3 * M.R ? A_SendSV : ord < 4 * M.R ? A_ExecuteOp : A_ReceiveClientRequest;
                } else {
                  const int q2 = ord - M.m0, j = q2 / (M.R + 1), k2 = q2 - j * (M.R + 1);
                  int kind0 = 0;
                  (void)Ops::guard(M, rec_mine, M.m0 + j, &kind0);
                  kind = k2 == 0 ? kind0 : Ops::other_kind(kind0);
                }
                emit(kind, p_mine, ord);
              }
            }
          }
        } else {
        u64 lut[6] = {0, 0, 0, 0, 0, 0};
        if (mine_valid) {
#pragma unroll
          for (int r = 1; r <= 5; r++)
            if (r <= M.R) lut[r] = prefilter_lut(M, Areg[r], r);
          for (int r = g + 1; r <= M.R; r += G) {
            u32 m = rep_slots_mask(M, rec_mine, hdr_mine, areg_of(Areg, r), r);
            while (m) {
              const int b = __ffs((int)m) - 1;
              m &= m - 1;
              const int kind = b == 0 ? A_TimerSendSVC : b == 1 ? A_SendDVC : b == 2 ? A_SendSV : b == 3 ? A_ExecuteOp : A_ReceiveClientRequest;
              emit(kind, p_mine, b < 4 ? b * M.R + (r - 1) : 4 * M.R + (r - 1) * M.C * M.n + (b - 4));
            }
          }
        }
        u16* s_surv = (u16*)s_cand2;                              // 2 * ccap entries of (record << 8 | bag index); s_cand2 is free until the sort
        const int nmsg_mine = mine_valid ? hdr_nmsg(hdr_mine) : 0;
        const int maxbag = (int)s_maxbag;
        const int jbatch = (int)((2u * ccap) >> tshift);          // bag entries per batch: even if every pair survives the list holds them
        for (int j0 = 0; j0 < maxbag; j0 += jbatch) {
          const int j1 = j0 + jbatch < maxbag ? j0 + jbatch : maxbag;
          for (int jj = j0; jj < j1; jj += G) {                   // block-uniform trip count (the ballot below wants every lane)
            const int j = jj + g;
            bool pass = false;
            if (j < j1 && j < nmsg_mine) {
              const u64 w = rec_mine[M.fixed + j];
              const int r = m_dest(w);
              const u64 l = r == 1 ? lut[1] : r == 2 ? lut[2] : r == 3 ? lut[3] : r == 4 ? lut[4] : lut[5];
              pass = m_count(w) != 0 && ((l >> (w & 63)) & 1);
              if (!IS_PLAIN && PLAIN != 3 && PLAIN != 4 && mode == MODE_PROBE && m_count(w) == 3) s_risky = 1;   // one more Send of this key would not fit the count field
            }
            const u64 bal = __ballot(pass);
            if (bal) {
              u32 base = 0;
              if (lane == 0) base = atomicAdd(&s_nsurv, (u32)__popcll(bal));
              base = (u32)__builtin_amdgcn_readfirstlane((int)base);
              if (pass) s_surv[base + (u32)__popcll(bal & (((u64)1 << lane) - 1))] = (u16)((p_mine << 8) | j);
            }
          }
          VSR_SYNC_G(4);
          const u32 nsurv = s_nsurv;
          for (u32 i = tid; i < nsurv; i += BLK) {
            const int p = s_surv[i] >> 8, j = s_surv[i] & 255;
            int kind0 = 0;
            u32 mask = Ops::guard(M, s_rec + p * stride, M.m0 + j, &kind0);
            const int ordbase = M.m0 + j * (M.R + 1);
            while (mask) {
              const int k = __ffs((int)mask) - 1;
              mask &= mask - 1;
              emit(k == 0 ? kind0 : Ops::other_kind(kind0), p, ordbase + k);
            }
          }
          if (j1 < maxbag) {                                      // another batch: the list is reused
            VSR_SYNC_G(4);
            if (tid == 0) s_nsurv = 0;
            VSR_SYNC_G(4);
          }
        }
        }   // (PLAIN != 3)
      } else
      {
      // four independent guard evaluations per trip
      for (int item0 = tid; item0 < nitems; item0 += 4 * BLK) {
        u32 masks[4];
        int kinds[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int item = item0 + u * BLK;
          const int slot = item >> tshift;
          masks[u] = 0;
          kinds[u] = 0;
          if (item < nitems && mine_valid) masks[u] = Ops::guard_pre(M, rec_mine, hdr_mine, Areg, slot, &kinds[u], s_slotinfo[slot & 63]);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
          u32 mask = masks[u];
          if (!mask) continue;
          const int slot = (item0 + u * BLK) >> tshift;
          const int ordbase = slot < M.m0 ? slot : M.m0 + (slot - M.m0) * (M.R + 1);
          alive = true;
          while (mask) {
            const int k = __ffs((int)mask) - 1;
            mask &= mask - 1;
            const int kind = k == 0 ? kinds[u] : Ops::other_kind(kinds[u]);
            atomicAdd(&s_kcount[kind], 1u);
            u32 idx;
            if (nmine < PRIV) idx = (u32)tid * PRIV + nmine;
            else idx = shared0 + atomicAdd(&s_ncand, 1u);
            nmine++;
            if (idx < ccap) s_cand[idx] = ((u32)kind << 18) | ((u32)p_mine << 11) | (u32)(ordbase + k);
            else s_ncand = 0x40000000u;                           // overflow marker (work list too small)
          }
        }
      }
      if (alive) s_alive[p_mine] = 1;
      }
    }
    VSR_SYNC_G(5);
    if (tid < np_tile && s_alive[tid] == 0 && s_ref[tid] != 0) atomicAdd(&s_dead, 1u);   // ref 0 = unused index (see k_materialize)
    // ---- counting sort of the work list by action id: the lanes of a wave then run the same action (no divergence between
    // the 15 action bodies, only inside one)
    if (tid == 0) {
      if (s_ncand >= 0x40000000u) {
        bool listed = false;
        if constexpr (REDO_OK) {
          if (np_tile > 1) {                                     // the work list is too short for this tile: written down for the host, which launches it again in halves
            const u64 k = atomicAdd((unsigned long long*)&ctl->n_redo, 1ull);
            if (k < ctl->redo_cap) { ((u64*)ctl->redo_out)[k] = p_base | ((u64)np_tile << 48); listed = true; s_over = 1; }
          }
        }
        if (!listed) raise_error(ctl, ERR_FRONTIER_FULL, p_base);
      }
      u32 acc = 0;
      // probe level: an action outside the invariants' footprint cannot turn a passing parent into a violating successor (Ops::probe_actions):
      // its instances are counted and sorted behind the others, and the apply loop stops in front of them.  (Compiled into the mode-capable
      // kernels only: the plain kernels sit on a register-allocation cliff — one more LDS word here cost the README configuration's 20
      // stored levels 22 ms.)
      if constexpr (!IS_PLAIN && FUSED && VSR_PROBE_FOOTPRINT) {
        const u32 keep = (mode == MODE_PROBE && !no_footprint) ? Ops::probe_actions() : ~0u;
        for (int a = 0; a < 16; a++)
          if ((keep >> a) & 1u) {
            s_kbase[a] = acc;
            acc += s_kcount[a];
          }
        s_napply = acc > ccap ? ccap : acc;
        if (keep != ~0u) {
          for (int a = 0; a < 16; a++)
            if (!((keep >> a) & 1u)) {
              s_kbase[a] = acc;
              acc += s_kcount[a];
            }
          // the instances behind s_napply are not applied: is a record of this tile so close to a representation limit that one of them could have hit it?
          if (!s_over && (s_risky || (int)s_maxbag + M.R - 1 > M.max_bag)) s_acc[15] += acc - s_napply;   // (an overflowed tile is counted when it is taken again)
        }
      } else {
        for (int a = 0; a < 16; a++) {
          s_kbase[a] = acc;
          acc += s_kcount[a];
        }
      }
      s_ncand = acc > ccap ? ccap : acc;
      if constexpr (PLAIN == 6) { s_ntotal = acc; s_ncand = s_napply; }   // (listed = the footprint's instances; every instance counts as generated)
#if VSR_TAKE
      {
        const u32 q8 = (acc << 8) / (u32)np_tile;
        avg_q8 = avg_q8 ? (3u * avg_q8 + q8) >> 2 : q8;
        const u32 want = avg_q8 ? ((u32)VSR_TAKE << 8) / avg_q8 : (u32)tile;
        my_take = want < 8u ? 8u : want > (u32)tile ? (u32)tile : want;
      }
#endif
      if (fused) s_wneed = s_ncand * (u32)(M.fixed + (int)s_maxbag + 5);   // upper bound of the successors' total length
    }
    const u64 t_2 = VSR_CLK();
    VSR_SYNC_G(6);
    const u32 ncand = s_ncand;
    if constexpr (REDO_OK) {
      if (s_over) {                                             // (block-uniform) the tile goes to the host's list: nothing of it has been applied or counted
        VSR_SYNC_G(6);                                          // every wave has read the flag before thread 0 clears it at the top of the loop
        continue;
      }
    }
    for (u32 c = tid; c < ccap; c += BLK) {
      const u32 code = s_cand[c];
      if (code == ~0u) continue;
      const u32 pos = atomicAdd(&s_kbase[code >> 18], 1u);
      if (pos < ccap) s_cand2[pos] = code;
    }
    VSR_SYNC_G(6);

    const u64 t_3 = VSR_CLK();
    if (fused && (mode == MODE_NORMAL || mode == MODE_REGEN)) {
      // ---- reserve room for this tile's successors (upper bounds: ncand states, s_wneed words) in the block's chunks
      if (s_ich_used + ncand > ichunk) {                        // block-uniform
        const u32 used = s_ich_used;
        const u64 base = s_ich_base;
        for (u32 k = used + tid; k < ichunk; k += BLK) {  // unused indices of the old chunk: invalid refs
          nx_off[base + k] = 0;
          lvl_fp[base + k] = 0;
        }
        VSR_SYNC_G(6);
        if (tid == 0) {
          u64 nb = atomicAdd((unsigned long long*)&ctl->n_new, (unsigned long long)ichunk);
          if (nb + ichunk > nx_cap) { raise_full(ctl, nb); nb = 0; }     // keep writing inside the buffer; the records are discarded, the claims stand
          s_ich_base = nb;
          s_ich_used = 0;
        }
        VSR_SYNC_G(6);
      }
      if (s_wch_used + s_wneed > wchunk) {                      // records do not straddle chunks: the remainder is skipped
        VSR_SYNC_G(6);
        if (tid == 0) {
          u64 nb = atomicAdd((unsigned long long*)&ctl->words_new, (unsigned long long)wchunk);
          if (nb + wchunk > nx_words_cap) { raise_full(ctl, nb); nb = 0; }
          // a tile whose successors cannot fit even a fresh chunk would run into the next block's chunk: the host sizes
          // wchunk >= ccap * (stride + 5) so that this cannot happen; refuse instead of corrupting records if it ever does
          if (s_wneed > wchunk) { raise_error(ctl, ERR_FRONTIER_FULL, p_base); s_skip = 1; }
          s_wch_base = nb;
          s_wch_used = 0;
        }
        VSR_SYNC_G(6);
      }
      if (tid == 0) { s_tile_ibase = s_ich_used; s_tile_wbase = s_wch_used; s_tile_icur = 0; s_tile_wcur = 0; }
    } else
    // ---- reserve room for this tile's pending entries (at most ncand) in the block's chunk
    if (!fused && s_chunk_used + ncand > pchunk) {          // block-uniform
      const u32 used = s_chunk_used;
      const u64 base = s_chunk_base;
      for (u32 k = used + tid; k < pchunk && used < pchunk; k += BLK) pending[3 * (base + k) + 1] = ~(u64)0;
      VSR_SYNC_G(6);
      if (tid == 0) {
        u64 nb = atomicAdd((unsigned long long*)&ctl->n_pending, (unsigned long long)pchunk);
        if (nb + pchunk > pending_cap) {
          raise_error(ctl, ERR_FRONTIER_FULL, nb);
          nb = 0;                                               // keep writing inside the buffer; the level is discarded
        }
        s_chunk_base = nb;
        s_chunk_used = 0;
      }
      VSR_SYNC_G(6);
    }
    const u32 ncand_apply = s_skip ? 0u : ((!IS_PLAIN && FUSED && VSR_PROBE_FOOTPRINT) ? s_napply : ncand);                // s_skip: the tile was refused (see the word-chunk reservation)
    constexpr bool dd_on = false;
    if (tid == 0) { s_tile_base = s_chunk_used; s_tile_cursor = 0; }
    VSR_SYNC_G(6);
    // ---- apply + fingerprint + seen-set claim: one lane per enabled instance
    // (no per-lane statistics: a lane runs this body once per tile, so every loop-carried register is a register of the body's peak — words
    // written come from the tile's word cursor, the largest bag / the virtual level's checksums go to LDS when a state is new, probes are counted
    // as one per candidate plus an LDS atomic per slot beyond the home slot)
    const CntLds my_probes{&s_acc[2]};
#if VSR_ROUND_REV
    for (u32 c0 = 0; c0 < ncand_apply; c0 += BLK) {
      const u32 c = c0 + (((c0 / BLK) & 1u) ? (u32)(BLK - 1 - tid) : (u32)tid);
      if (c >= ncand_apply) continue;
#else
    for (u32 c = tid; c < ncand_apply; c += BLK) {
#endif
      const u32 code = s_cand2[c];
      const int p = (int)((code >> 11) & 127), ord = (int)(code & 2047);
      const u64* rec = s_rec + p * stride;
      Delta D;
      const u64 a_0 = VSR_CLK();
      if (!Ops::template gen_<false>(M, rec, ord, D) || D.action != (int)(code >> 18)) {
        raise_error(ctl, ERR_INTERNAL, ((p_base + (u64)p) << 16) | (u64)ord);
        continue;
      }
      if (D.err) {
        raise_error(ctl, D.err, ((p_base + (u64)p) << 16) | (u64)ord);
        continue;
      }
      if constexpr (PLAIN == 6) {                                 // probe-only: a successor that fails an invariant is written down for k_probe_resolve, nothing else happens here
        if (Ops::invariants(M, rec, D) != 0) {
          const u64 i = atomicAdd((unsigned long long*)&ctl->n_pending, 1ull);
          if (i < pending_cap) { pending[2 * i] = origin_make(p_base + (u64)p, ord); pending[2 * i + 1] = 0; }   // (p_offset is 0 for this instantiation, or the tile list)
        }
        continue;
      }
      const u64 a_1 = VSR_CLK();
      // Probe level: nothing is inserted, so the fingerprint and the seen-set matter only for a successor that VIOLATES an invariant — it is
      // reported unless it is a state of an earlier level.  Invariants first, then (for the handful that fail) hash and lookup: the README
      // configuration's level 24 has 3.8e9 successors of which 8 violate; hashing them under six permutations and fetching a 128-byte line
      // of the seen-set for each was a third of that run.
      if (!IS_PLAIN && fused && mode == MODE_PROBE && Ops::invariants(M, rec, D) == 0) continue;
      u64 Hc[6];
      Ops::hash_child_(M, rec, D, Hc);
      u64 fp;
      u32 ak;
      canonical_fp(M, D.hdr, Hc, &fp, &ak);
      const u64 key = meta_make(level, ak, s_pfp[p]);
      const u64 a_2 = VSR_CLK();
      if (tid == 0) { s_acc[10] += a_1 - a_0; s_acc[11] += a_2 - a_1; }
      if (!fused && world > 1) {                                // sharded seen-set, exact scheme: route to the owner of fp
        const int owner = owner_of(fp, world);
        if (owner != rank) {
          for (int o = 0; o < world; o++)
            if (owner == o) {
              u64 i = wave_alloc(&ctl->cand_cnt[o]);
              if (i < cand_cap) {
                cand_send[2 * ((u64)o * cand_cap + i)] = fp;
                cand_send[2 * ((u64)o * cand_cap + i) + 1] = key;
                cand_idx[(u64)o * cand_cap + i] = origin_make(p_offset + p_base + (u64)p, ord);   // where it comes from, for k_materialize on this rank
              } else {
                raise_error(ctl, ERR_FRONTIER_FULL, i);
              }
            }
          continue;
        }
      }
      if (fused) {
        bool do_write = false, remote = false, check = false, claimed_now = false;
        u64 prev_meta = META_EMPTY;
        const int owner = world > 1 ? owner_of(fp, world) : rank;
        if (mode == MODE_PROBE) {                               // probe level: looked up, not inserted; unseen successors are checked
          u64 m = META_EMPTY;
          check = !(probe_lookup(table, tmask, fp, &m, my_probes) && meta_level(m) < level);
          if (!check) continue;
        } else if (owner != rank) {
          // sent-filter: a direct-mapped, lossy set of (fingerprint, auxkey) tags this rank has announced before (any level).
          // A hit = an exact repeat, dropped; a miss (or an evicted tag) only costs a redundant announcement.  Plain 8-byte
          // loads / stores: a lost update has the same effect as an eviction.  (MODE_REGEN asks the owner about EVERY candidate: which
          // of the copies of a state carries the slot's final key is not a question a repeat filter can answer.)
          if (mode != MODE_REGEN) {
            u64 tag = fp ^ ((u64)(ak + 1) * 0x9E3779B97F4A7C15ull);
            if (tag == 0) tag = 1;
            u64* fs = filter + ((fp >> 6) & fmask);
            if (__hip_atomic_load(fs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == tag) continue;
            __hip_atomic_store(fs, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            remote = true;
            do_write = mode == MODE_NORMAL;                     // a virtual level: announced, not written (vsr_deep.hpp, sharded)
          } else if (wset) {
            do_write = wset_take(wset, fp, level, wepoch);      // regenerated level: this rank rebuilds what ITS candidates inserted — nobody is asked
          } else {
            remote = true;                                      // (no winner set: the owner grants the regeneration, rounds 3-4)
          }
        } else if (mode == MODE_REGEN && PLAIN == 0 && wset) {
          do_write = wset_take(wset, fp, level, wepoch);        // the same rule for the states this rank owns itself: no seen-set access in a sharded regeneration
        } else if (PLAIN == 3) {
          // The first seen-set-only level, unsharded (round 5): the pass that INSERTED it left one bit per (parent, ordinal) whose lane made a state
          // (`filter` / `fmask` carry the bitmap and its words per parent here — an unsharded pass has no sent-filter).  That instance rebuilds the
          // state: no seen-set access at all, exactly once by construction (a state has one inserting lane).
          do_write = (((const u32*)filter)[(p_offset + p_base + (u64)p) * fmask + ((u32)ord >> 5)] >> ((u32)ord & 31u)) & 1u;
        } else if (mode == MODE_REGEN) {
          u64 m = META_EMPTY;
          u64 slot_i = 0;
          do_write = probe_lookup(table, tmask, fp, &m, my_probes, &slot_i) && m == key &&
                     atomicCAS((unsigned long long*)&table[slot_i].meta, (unsigned long long)key, (unsigned long long)(key | META_TAKEN)) == key;
        } else {
          bool claimed, full;
          table_claim_fused(table, tmask, fp, key, level, &claimed, &prev_meta, my_probes, &full);
          if (full) {
            raise_error(ctl, ERR_TABLE_FULL, fp);
            continue;
          }
          do_write = claimed;
          claimed_now = claimed;
          if (PLAIN == 0 && wset && claimed && !wset_insert(wset, fp, level)) raise_error(ctl, ERR_TABLE_FULL, fp);   // a state this rank's own lane made
          if (PLAIN == 4 && filter && claimed)   // unsharded virtual level: remember WHICH instance made the state (see PLAIN == 3 above)
            atomicOr(&((u32*)filter)[(p_offset + p_base + (u64)p) * fmask + ((u32)ord >> 5)], 1u << ((u32)ord & 31u));
        }
        // the ONE evaluation of the invariants (three inlined copies pushed the mode-capable kernels out of the instruction cache)
        int bad = (check || do_write || (remote && mode == MODE_INSERT)) ? Ops::invariants(M, rec, D) : 0;
#ifdef VSRMC_TEST_HOOKS                                       // test hook (Model::test_bad_fp): only in the library built with -DVSRMC_TEST_HOOKS (libvsrmc_hooks.so)
        if constexpr (PLAIN == 0)
          if (M.test_bad_mask && fp == M.test_bad_fp && (check || do_write || remote)) bad |= (int)M.test_bad_mask;
#endif
        if (mode == MODE_PROBE) {
          if (bad) {
            const u64 i = atomicAdd((unsigned long long*)&ctl->n_pending, 1ull);
            if (i < pending_cap) {
              pending[2 * i] = fp;
              pending[2 * i + 1] = key;
            }
            atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
            atomicOr(&ctl->viol_mask, (u32)bad);
          }
          continue;
        }
        if (do_write && mode == MODE_INSERT) {                  // virtual level: the state is counted and checked, not stored
          if (bad) {
            const u64 i = atomicAdd((unsigned long long*)&ctl->n_pending, 1ull);
            if (i < pending_cap) {
              pending[2 * i] = fp;
              pending[2 * i + 1] = key;
            }
            atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
            atomicOr(&ctl->viol_mask, (u32)bad);
          }
          if ((u32)hdr_nmsg(D.hdr) > s_maxbag_out) atomicMax(&s_maxbag_out, (u32)hdr_nmsg(D.hdr));
          atomicAdd(&s_acc[9], 1ull);                           // counts states in this mode
          atomicXor(&s_fxs[0], (unsigned long long)fp);         // into LDS, not into the control block (flushed once per block, in the epilogue)
          atomicAdd(&s_fxs[1], (unsigned long long)fp);
          do_write = false;
        }
        const u64 a_3 = VSR_CLK();
        if (tid == 0) s_acc[12] += a_3 - a_2;
        u64 idx = 0;
        // Since round 6 the ordinary level's instantiation (PLAIN == 1) copies the parent words of the successors a wave writes in a round with the WHOLE wave —
        // lane k moves word k of one record per instruction: one contiguous 350-byte store instead of twenty-two 16-byte stores of the claiming lane alone, and
        // the LDS reads of different records are independent of each other (the lane-serial copy waits for its own LDS read in every trip).  The claiming lane
        // then writes its patches on top (same wave, program order).  Lanes that left the body earlier (errors) take no part: the words are shared out over the
        // lanes that are here.  (Config 2: k_expand -2.5 %, with five blocks per CU -8 %; the regenerating instantiation LOSES 13 % with it — every lane writes
        // there — and keeps the lane-serial copy)
        u64 dst = 0;
        int plen = 0, clen = 0;
        if (do_write) {                                         // new state (or: possibly new, the owner decides): room in the tile's reservation
          plen = (int)(s_ref[p] & 255);
          clen = M.fixed + hdr_nmsg(D.hdr);
          const u32 io = atomicAdd(&s_tile_icur, 1u);
          const u32 wo = atomicAdd(&s_tile_wcur, (u32)clen);
          idx = s_ich_base + s_tile_ibase + io;
          dst = s_wch_base + s_tile_wbase + wo;
        }
        if constexpr (COOP) {
          const u64 act = __ballot(1);
          u64 wm = __ballot(do_write);
          const int nact = __popcll(act), rnk = __popcll(act & (((u64)1 << lane) - 1));
          while (wm) {
            const int l0 = __ffsll((long long)wm) - 1;
            wm &= wm - 1;
            const int l1 = wm ? __ffsll((long long)wm) - 1 : l0;   // two records per trip: their LDS reads are in flight together
            wm &= wm - 1;
            const int n0 = __builtin_amdgcn_readlane(plen, l0), n1 = l1 != l0 ? __builtin_amdgcn_readlane(plen, l1) : 0;
            const u64* r0 = s_rec + __builtin_amdgcn_readlane(p, l0) * stride;
            const u64* r1 = s_rec + __builtin_amdgcn_readlane(p, l1) * stride;
            u64* o0 = nx_words + readlane64(dst, l0);
            u64* o1 = nx_words + readlane64(dst, l1);
            for (int k = rnk; k < n0 || k < n1; k += nact) {
              const u64 v0 = k < n0 ? r0[k] : 0, v1 = k < n1 ? r1[k] : 0;
              if (k < n0) o0[k] = v0;
              if (k < n1) o1[k] = v1;
            }
          }
        }
        if (do_write) {
          u64* out = nx_words + dst;
          if constexpr (!COOP) {
          // parent from LDS, 16 bytes per store (records are 8-byte aligned), then the patches on top (same lane: ordered)
            typedef u64 u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
            int k = 0;
#if VSR_COPY8
            for (; k + 7 < plen; k += 8) {                        // EXPERIMENT: eight words per trip, four LDS reads in flight (round 3 lost registers to this; MachineLICM off leaves room)
              u64x2_a8 a0, a1, a2, a3;
              a0.x = rec[k]; a0.y = rec[k + 1]; a1.x = rec[k + 2]; a1.y = rec[k + 3];
              a2.x = rec[k + 4]; a2.y = rec[k + 5]; a3.x = rec[k + 6]; a3.y = rec[k + 7];
              *(u64x2_a8*)(out + k) = a0;
              *(u64x2_a8*)(out + k + 2) = a1;
              *(u64x2_a8*)(out + k + 4) = a2;
              *(u64x2_a8*)(out + k + 6) = a3;
            }
#endif
            if (plen - k >= 2) {
              u64x2_a8 cur;
              cur.x = rec[k];
              cur.y = rec[k + 1];
              for (; k + 3 < plen; k += 2) {
                u64x2_a8 nxt;
                nxt.x = rec[k + 2];
                nxt.y = rec[k + 3];
                *(u64x2_a8*)(out + k) = cur;
                cur = nxt;
              }
              *(u64x2_a8*)(out + k) = cur;
              k += 2;
            }
            for (; k + 1 < plen; k += 2) {
              u64x2_a8 v2;
              v2.x = rec[k];
              v2.y = rec[k + 1];
              *(u64x2_a8*)(out + k) = v2;
            }
            if (k < plen) out[k] = rec[k];
          }
          out[0] = D.hdr;
          u64* ob = out + 1 + (D.r - 1) * M.wpr;
          ob[0] = D.rep[0];
          if (M.wpr > 1) ob[1] = D.rep[1];
          if (M.wpr > 2) ob[2] = D.rep[2];
          if (M.wpr > 3) ob[3] = D.rep[3];
#pragma unroll
          for (int k = 0; k < 6; k++)
            if (k < M.np) out[M.h0 + k] = Hc[k];
          int a = 0;
#pragma unroll
          for (int k = 0; k < VSR_NSLOT; k++)
            if ((D.used >> k) & 1) {
              if (D.pj(k) >= 0) out[M.fixed + D.pj(k)] = D.pnew[k];
              else out[plen + (a++)] = D.pnew[k];
            }
          nx_off[idx] = (dst << 8) | (u64)clen;
          lvl_fp[idx] = fp;
          if (bad && !remote) {
            atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
            atomicOr(&ctl->viol_mask, (u32)bad);
          }
          if ((u32)hdr_nmsg(D.hdr) > s_maxbag_out) atomicMax(&s_maxbag_out, (u32)hdr_nmsg(D.hdr));
        }
        if (PLAIN == 0 && remote) {
          // announce (fp, key) to the owner: entry i of the block's chunk of that owner's bucket
          u64 i = 0;
          for (;;) {
            const unsigned long long old = atomicAdd(&s_cstate[owner], 1ull);
            const u32 pos = (u32)(old & 0xFFFFFFull);
            if (pos == cchunk) {                              // drew the "chunk is full" ticket: fetch the next chunk
              u64 nb = atomicAdd((unsigned long long*)&ctl->cand_cnt[owner], (unsigned long long)cchunk);
              if (nb + cchunk > cand_cap) { raise_error(ctl, ERR_FRONTIER_FULL, nb); nb = 0; }
              const u64 ob_ = (u64)(old >> 24);
              (void)ob_;                                      // the old chunk was used up completely: nothing to invalidate
              i = nb;
              atomicExch(&s_cstate[owner], ((unsigned long long)nb << 24) | 1ull);
              break;
            }
            if (pos < cchunk) {
              i = (u64)(old >> 24) + pos;
              break;
            }
          }                                                   // pos > cchunk: another lane is fetching the chunk; draw again
          const u64 e = (u64)owner * cand_cap + i;
          cand_send[2 * e] = fp;
          cand_send[2 * e + 1] = key;
          // what the generator keeps beside the candidate: where it wrote the record speculatively (ordinary level), the size of the state's bag
          // (virtual level: nothing is written), or the instance that regenerates it (parent index in this launch's source, ordinal)
          cand_idx[e] = mode == MODE_REGEN ? origin_make(p_base + (u64)p, ord)
                                           : cand_pack(mode == MODE_INSERT ? (u64)hdr_nmsg(D.hdr) : idx, (u32)bad);
        }
        // same-level duplicate with a different canonical auxkey = the tie the single-pass scheme cannot arbitrate
        if (prev_meta != META_EMPTY && meta_level(prev_meta) == level && meta_auxkey(prev_meta) != meta_auxkey(key)) atomicAdd(&s_acc[8], 1ull);
        if (tid == 0) s_acc[13] += VSR_CLK() - a_3;
        continue;
      }
      bool found_old, full;
      u64 slot = table_claim(table, tmask, fp, key, level, &found_old, my_probes, &full);
      if (full) {
        raise_error(ctl, ERR_TABLE_FULL, fp);
        continue;
      }
      if (!found_old) {                                         // one LDS atomic per wave, no global counter
        const u64 active = __ballot(1);
        const int leader = __ffsll((long long)active) - 1;
        u32 b = 0;
        if (lane == leader) b = atomicAdd(&s_tile_cursor, (u32)__popcll(active));
        b = __shfl(b, leader);
        const u64 i = s_chunk_base + s_tile_base + b + (u32)__popcll(active & (((u64)1 << lane) - 1));
        pending[3 * i] = slot;
        pending[3 * i + 1] = key;
        pending[3 * i + 2] = origin_make(p_offset + p_base + (u64)p, ord);
      }
    }
    const u64 t_4 = VSR_CLK();
    VSR_SYNC_G(7);
    if (tid == 0) {
      if (fused) {
        s_ich_used += s_tile_icur;
        s_wch_used += s_tile_wcur;
        s_acc[14] += s_tile_icur;
        if (mode == MODE_NORMAL || mode == MODE_REGEN) s_acc[9] += s_tile_wcur;   // words of the records this tile wrote
      } else {
        s_chunk_used += s_tile_cursor;
      }
      s_acc[0] += PLAIN == 6 ? s_ntotal : s_ncand;
      s_acc[1] += s_dead;
      if (PLAIN != 6) s_acc[2] += ncand_apply;                   // one home-slot probe per applied candidate
      const u64 t_5 = VSR_CLK();
      s_acc[3] += t_1 - t_0;
      s_acc[4] += t_2 - t_1;
      s_acc[5] += t_3 - t_2;
      s_acc[6] += t_4 - t_3;
      s_acc[7] += t_5 - t_4;
    }
    if (tid < 16) s_acc[16 + tid] += s_kcount[tid];
#if !VSR_NO_TAIL_SYNC
    VSR_SYNC_G(0);
#endif
  }
  // ---- block epilogue: invalidate the unused tail of the chunk, flush the accumulators
  {
    if (fused) {
      const u32 used = s_ich_used;
      const u64 base = s_ich_base;
      for (u32 k = used + tid; k < ichunk; k += BLK) {
        nx_off[base + k] = 0;
        lvl_fp[base + k] = 0;
      }
      if (world > 1) {
        VSR_SYNC_G(0);
        for (int o = 0; o < world; o++) {
          const u64 st = s_cstate[o];
          const u64 base = st >> 24;
          const u32 used = (u32)(st & 0xFFFFFFull) < cchunk ? (u32)(st & 0xFFFFFFull) : cchunk;
          if (base == CS_NONE) continue;
          for (u32 k = used + tid; k < cchunk; k += BLK) {
            cand_send[2 * ((u64)o * cand_cap + base + k)] = 0;  // fingerprint 0 = no candidate
            cand_send[2 * ((u64)o * cand_cap + base + k) + 1] = ~(u64)0;
          }
        }
      }
      if (tid == 0) {
        if (s_acc[8]) atomicAdd((unsigned long long*)&ctl->ties, s_acc[8]);
        if (!IS_PLAIN && s_acc[15]) atomicAdd((unsigned long long*)&ctl->limit_unchecked, s_acc[15]);
        if (s_acc[14]) atomicAdd((unsigned long long*)&ctl->n_written, s_acc[14]);
        if (s_acc[9]) atomicAdd((unsigned long long*)(mode == MODE_INSERT ? &ctl->n_new : &ctl->rec_words), s_acc[9]);
        if (s_maxbag_out) atomicMax((unsigned long long*)&ctl->max_bag, (unsigned long long)s_maxbag_out);
        if (!IS_PLAIN && (s_fxs[0] | s_fxs[1])) {
          atomicXor((unsigned long long*)&ctl->fp_xor, s_fxs[0]);
          atomicAdd((unsigned long long*)&ctl->fp_sum, s_fxs[1]);
        }
      }
    } else {
      const u32 used = s_chunk_used;
      const u64 base = s_chunk_base;
      for (u32 k = used + tid; k < pchunk; k += BLK) pending[3 * (base + k) + 1] = ~(u64)0;
    }
#if VSR_NO_FLUSH     // EXPERIMENT (timing only: the level's reported figures are wrong): how much of a launch is the drain of the per-block statistics?
    if (tid >= 0) return;
#endif
    if (tid == 0) {
      if (s_acc[0]) atomicAdd((unsigned long long*)&ctl->generated, s_acc[0]);
      if (s_acc[1]) atomicAdd((unsigned long long*)&ctl->deadlocks, s_acc[1]);
      if (s_acc[2]) atomicAdd((unsigned long long*)&ctl->probes, s_acc[2]);
    }
#if VSR_WAVE_DIAG == 3
    if (tid == 64) {
      for (int g = 0; g < 8; g++) atomicAdd((unsigned long long*)&ctl->phase_cycles[g], s_wd[g]);
      atomicAdd((unsigned long long*)&ctl->act_generated[0], (unsigned long long)(__builtin_readcyclecounter() - wd_t0));
    }
    (void)wd_wait;
#elif VSR_WAVE_DIAG
    if (lane == 0 && (tid >> 6) < 4) {
      atomicAdd((unsigned long long*)&ctl->phase_cycles[tid >> 6], (unsigned long long)wd_wait);
      atomicAdd((unsigned long long*)&ctl->phase_cycles[4 + (tid >> 6)], (unsigned long long)(__builtin_readcyclecounter() - wd_t0));
    }
#else
    if (tid >= 3 && tid < 8 && s_acc[tid]) atomicAdd((unsigned long long*)&ctl->phase_cycles[tid - 3], s_acc[tid]);
    // wave 0's clock inside the apply loop: gen, hash, probe (into phase_cycles[5..7]), successor write (act_generated[0])
    if (tid >= 10 && tid < 13 && s_acc[tid]) atomicAdd((unsigned long long*)&ctl->phase_cycles[tid - 5], s_acc[tid]);
    if (tid == 13 && s_acc[13]) atomicAdd((unsigned long long*)&ctl->act_generated[0], s_acc[13]);
#endif
    if (tid >= 16 && tid < 32 && s_acc[tid]) atomicAdd((unsigned long long*)&ctl->act_generated[tid - 16], s_acc[tid]);
  }
}

// Write the child record: copy of the parent with the Delta applied.  Single-lane version (k_successors, k_replay).
__device__ __forceinline__ void write_child_serial(const Model& M, const u64* rec, const Delta& D, const u64* Hc, u64* dst) {
  int nmsg = hdr_nmsg(rec[0]);
  int len = M.fixed + nmsg;
  for (int k = 0; k < len; k++) dst[k] = rec[k];
  dst[0] = D.hdr;
  for (int k = 0; k < M.wpr; k++) dst[1 + (D.r - 1) * M.wpr + k] = D.rep[k];
  for (int i = 0; i < M.np; i++) dst[M.h0 + i] = Hc[i];
  int a = 0;
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if ((D.used >> k) & 1) {
      if (D.pj(k) >= 0) dst[M.fixed + D.pj(k)] = D.pnew[k];
      else dst[M.fixed + nmsg + (a++)] = D.pnew[k];
    }
}

// -----------------------------------------------------------------------------------------------------------------
// k_materialize: one wave per block, one lane per pending entry.  Winners (slot meta == own key, or the owner's verdict)
// rebuild their successor: the parent records of the wave's winners are staged in LDS (cooperative, coalesced: lane k
// moves word k), each winner lane re-runs its action against its LDS slot and patches the slot in place into the child,
// and the children are written out cooperatively — every HBM word of the next frontier is written exactly once.
// -----------------------------------------------------------------------------------------------------------------
#define VSR_MAT_BLOCK 64
#define VSR_MAT_GROUP 16

__device__ __forceinline__ void lds_wave_sync() {   // orders this wave's LDS writes before its later LDS reads (other lanes)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int SPEC>
__global__ void __launch_bounds__(VSR_MAT_BLOCK)
k_materialize(Model Marg, const u64* __restrict__ fr_words, const u64* __restrict__ fr_off, const u64* __restrict__ pending,
              u64 n_pending, Slot* table, u64* nx_words, u64 nx_words_cap, u64* nx_off, u64 nx_cap,
              u64* lvl_fp, LevelCtl* ctl, const uint8_t* __restrict__ verdict, u64* cnt_n, u64* cnt_w, int stride,
              u32 ichunk /* state indices per reservation, >= 64 */, u32 wchunk /* words per reservation, >= 64 * stride */,
              // entry i = pending[es * i ..): (slot or fingerprint, key[, origin]); origin = parent index | ordinal << 40 comes from
              // the entry's third word (local pending list, es = 3) or from pidx_arr[i] (candidates sent to a remote owner, es = 2)
              int es, const u64* __restrict__ pidx_arr) {
  Model M = Marg;
  specialise<SPEC>(M, Marg);
  typedef ModelOps<SPEC / 1000> Ops;
  extern __shared__ u64 s_slot[];                              // 64 slots of `stride` words
  const int lane = threadIdx.x;
  // This wave's private chunks of the output (wave-uniform values): state indices [idx_base, idx_base + idx_left) and
  // words [w_base, w_base + w_left).  One global atomic per chunk instead of two per 64 entries; the unused tail of the
  // last index chunk is published as invalid refs (0), which every consumer skips.
  u64 idx_base = 0, w_base = 0;
  u32 idx_left = 0, w_left = 0;
  u64 cyc_fetch = 0, cyc_stage = 0, cyc_gen = 0, cyc_write = 0, rec_words = 0;
  const u64 nthreads = (u64)gridDim.x * VSR_MAT_BLOCK;
  const u64 rounds = (n_pending + nthreads - 1) / nthreads;
  for (u64 it = 0; it < rounds; it++) {
    const u64 i = it * nthreads + (u64)blockIdx.x * VSR_MAT_BLOCK + lane;
    const u64 c_0 = __builtin_readcyclecounter();
    bool win = false;
    u64 key = 0, src = 0, origin = 0;
    int plen = 0;
    if (i < n_pending) {
      key = pending[(u64)es * i + 1];
      if (key != ~(u64)0) {                                     // ~0 = unused entry of a block's chunk
        // winner test: the owner's verdict (sharded, remote owner) or the slot's final meta word (local owner) — taken with a
        // compare-and-swap key -> key | taken, so that a state is materialised exactly once; the parent's ref is fetched
        // alongside, not after (one HBM latency, not two)
        origin = pidx_arr ? pidx_arr[i] : pending[(u64)es * i + 2];
        const u64 ref = fr_off[origin_pidx(origin)];
        if (verdict) win = verdict[i] != 0;
        else win = atomicCAS((unsigned long long*)&table[pending[(u64)es * i]].meta, (unsigned long long)key,
                             (unsigned long long)(key | META_TAKEN)) == key;
        src = ref >> 8;
        plen = (int)(ref & 255);
        if (plen > stride) plen = stride;
      }
    }
    if (!win) plen = 0;
    const u64 wmask = __ballot(win);
    const u64 c_1 = __builtin_readcyclecounter();
    cyc_fetch += c_1 - c_0;
    if (wmask == 0) continue;
    // ---- (a) stage the winners' parents: slot w <- record of lane w, by LDS-DMA (global_load_lds_dword: lane l moves dword l
    // of the record straight into LDS, no VGPR round trip), so all ~2 x 64 loads of the wave are in flight at once
    for (int w = 0; w < 64; w++) {
      const int n2 = 2 * __builtin_amdgcn_readlane(plen, w);    // dwords; 0 for lanes without a winner
      if (n2 == 0) continue;
      const u32* g = (const u32*)(fr_words + readlane64(src, w));
      __attribute__((address_space(3))) void* l = (__attribute__((address_space(3))) void*)(s_slot + w * stride);
      if (lane < n2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane), l, 4, 0, 0);
      // the instruction offset applies to the global address AND the LDS address: same pointers, +256 bytes on both sides
      if (lane + 64 < n2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane), l, 4, 256, 0);
      if (lane + 128 < n2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane), l, 4, 512, 0);
      if (lane + 192 < n2) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + lane), l, 4, 768, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_wave_sync();
    const u64 c_2 = __builtin_readcyclecounter();
    cyc_stage += c_2 - c_1;
    // ---- (b) winners re-run their action on the staged parent and patch the slot into the child
    u64 fp = 0;
    int clen = 0, bad = 0, nbag = 0;
    if (win) {
      u64* rec = s_slot + lane * stride;
      Delta D;
      Ops::template gen_<false>(M, (const u64*)rec, origin_ord(origin), D);
      u64 Hc[6];
      Ops::hash_child_(M, (const u64*)rec, D, Hc);
      u32 ak;
      canonical_fp(M, D.hdr, Hc, &fp, &ak);
      bad = Ops::invariants(M, (const u64*)rec, D);
#ifdef VSRMC_TEST_HOOKS
      if (M.test_bad_mask && fp == M.test_bad_fp) bad |= (int)M.test_bad_mask;   // test hook (Model::test_bad_fp)
#endif
      nbag = hdr_nmsg(D.hdr);
      clen = M.fixed + nbag;
      if (clen > stride) clen = stride;                        // cannot happen: gen() raised ERR_REP_BAG in k_expand
      rec[0] = D.hdr;
      u64* pb = rec + 1 + (D.r - 1) * M.wpr;
      pb[0] = D.rep[0];
      if (M.wpr > 1) pb[1] = D.rep[1];
      if (M.wpr > 2) pb[2] = D.rep[2];
      if (M.wpr > 3) pb[3] = D.rep[3];
#pragma unroll
      for (int k = 0; k < 6; k++)
        if (k < M.np) rec[M.h0 + k] = Hc[k];
      int a = 0;
#pragma unroll
      for (int k = 0; k < VSR_NSLOT; k++)
        if ((D.used >> k) & 1) {
          if (D.pj(k) >= 0) rec[M.fixed + D.pj(k)] = D.pnew[k];
          else if (plen + a < stride) rec[plen + (a++)] = D.pnew[k];
        }
    }
    lds_wave_sync();
    const u64 c_3 = __builtin_readcyclecounter();
    cyc_gen += c_3 - c_2;
    // ---- wave-wide allocation out of the wave's chunks
    const int nwin = __popcll(wmask);
    int incl = clen;                                           // inclusive scan of child lengths over the wave
    for (int o = 1; o < 64; o <<= 1) {
      int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const u32 total = (u32)__builtin_amdgcn_readlane(incl, 63);
    const u32 rank = (u32)__popcll(wmask & (((u64)1 << lane) - 1));
    u64 idx;
    bool overflow = false;
    if (idx_left < (u32)nwin) {                                // finish the old index chunk, continue in a new one
      u64 nb = 0;
      if (lane == 0) nb = atomicAdd((unsigned long long*)cnt_n, (unsigned long long)ichunk);
      nb = readlane64(nb, 0);
      if (nb + ichunk > nx_cap) overflow = true;
      idx = rank < idx_left ? idx_base + rank : nb + (rank - idx_left);
      idx_base = nb + ((u32)nwin - idx_left);
      idx_left = ichunk - ((u32)nwin - idx_left);
    } else {
      idx = idx_base + rank;
      idx_base += (u32)nwin;
      idx_left -= (u32)nwin;
    }
    if (w_left < total) {                                      // records do not straddle word chunks: the remainder is skipped
      u64 nb = 0;
      if (lane == 0) nb = atomicAdd((unsigned long long*)cnt_w, (unsigned long long)wchunk);
      nb = readlane64(nb, 0);
      if (nb + wchunk > nx_words_cap) overflow = true;
      w_base = nb;
      w_left = wchunk;
    }
    const u64 dst = w_base + (u64)(incl - clen);
    rec_words += total;
    w_base += total;
    w_left -= total;
    if (overflow) {
      if (lane == 0) raise_error(ctl, ERR_FRONTIER_FULL, idx_base);
      idx_left = 0;                                            // nothing of the refused chunks is used
      w_left = 0;
      continue;
    }
    // ---- (c) write the children out: lane k moves word k of one record per instruction
    for (int g = 0; g < 64; g += VSR_MAT_GROUP) {
      if (((wmask >> g) & ((1u << VSR_MAT_GROUP) - 1)) == 0) continue;
#pragma unroll
      for (int q = 0; q < VSR_MAT_GROUP; q++) {
        const u64 d = readlane64(dst, g + q);
        const int n = __builtin_amdgcn_readlane(clen, g + q);
        if (lane < n) nx_words[d + lane] = s_slot[(g + q) * stride + lane];
        if (lane + 64 < n) nx_words[d + lane + 64] = s_slot[(g + q) * stride + lane + 64];   // records longer than 64 words (R >= 4)
      }
    }
    if (win) {
      nx_off[idx] = (dst << 8) | (u64)clen;
      lvl_fp[idx] = fp;
      if (bad) {
        atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
        atomicOr(&ctl->viol_mask, (u32)bad);
      }
    }
    // one max_bag update per wave
    int mb = nbag;
    for (int o = 32; o > 0; o >>= 1) {
      int t = __shfl_down(mb, o);
      mb = t > mb ? t : mb;
    }
    if (lane == 0 && (u64)mb > __hip_atomic_load(&ctl->max_bag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax((unsigned long long*)&ctl->max_bag, (unsigned long long)mb);
    lds_wave_sync();                                           // slots are reused by the next round
    cyc_write += __builtin_readcyclecounter() - c_3;
  }
  if (lane == 0) {
    atomicAdd((unsigned long long*)&ctl->phase_cycles[5], (unsigned long long)(cyc_fetch + cyc_stage));
    atomicAdd((unsigned long long*)&ctl->phase_cycles[6], (unsigned long long)cyc_gen);
    atomicAdd((unsigned long long*)&ctl->phase_cycles[7], (unsigned long long)cyc_write);
    atomicAdd((unsigned long long*)&ctl->rec_words, (unsigned long long)rec_words);
  }
  // the unused tail of the wave's last index chunk: invalid refs
  for (u32 k = lane; k < idx_left; k += 64) {
    nx_off[idx_base + k] = 0;
    lvl_fp[idx_base + k] = 0;
  }
}

// number of valid (non-zero) refs in a frontier index range
__global__ void k_count_valid(const u64* __restrict__ refs, u64 n, u64* out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u32 c = 0;
  for (; i < n; i += stride) c += refs[i] != 0 ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd((unsigned long long*)out, (unsigned long long)c);
}

// xor / sum (mod 2^64) / count of the fingerprints of a level's index range (0 = unused index): out[0..2]
__global__ void k_level_checksum(const u64* __restrict__ fps, u64 n, u64* out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 x = 0, s = 0, c = 0;
  for (; i < n; i += stride) {
    const u64 f = fps[i];
    x ^= f;
    s += f;
    c += f != 0 ? 1 : 0;
  }
  for (int o = 32; o > 0; o >>= 1) {
    x ^= __shfl_down(x, o);
    s += __shfl_down(s, o);
    c += __shfl_down(c, o);
  }
  if ((threadIdx.x & 63) == 0) {
    if (x) atomicXor((unsigned long long*)&out[0], (unsigned long long)x);
    if (s) atomicAdd((unsigned long long*)&out[1], (unsigned long long)s);
    if (c) atomicAdd((unsigned long long*)&out[2], (unsigned long long)c);
  }
}

// xor / sum / count of the fingerprints of the states of ONE level, read from the seen-set itself (a level whose lvl_fp array does not exist or
// is not to be trusted: a stored level that overflowed its record buffer and is kept as a seen-set-only level): out[0..2]
__global__ void k_table_level_checksum(const Slot* __restrict__ table, u64 n_slots, int level, u64* out) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  u64 x = 0, s = 0, c = 0;
  for (; i < n_slots; i += stride) {
    const u64 f = table[i].fp;
    if (f != 0 && meta_level(table[i].meta) == level) { x ^= f; s += f; c++; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    x ^= __shfl_down(x, o);
    s += __shfl_down(s, o);
    c += __shfl_down(c, o);
  }
  if ((threadIdx.x & 63) == 0 && c) {
    atomicXor((unsigned long long*)&out[0], (unsigned long long)x);
    atomicAdd((unsigned long long*)&out[1], (unsigned long long)s);
    atomicAdd((unsigned long long*)&out[2], (unsigned long long)c);
  }
}

// re-insertion of every occupied slot into a table of another size (growth of the seen-set between two levels: host_search.hpp, table_grow)
__global__ void k_table_rehash(const Slot* __restrict__ old_table, u64 n_old, Slot* table, u64 tmask, u32* err) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n_old; i += stride) {
    const Slot s = old_table[i];
    if (s.fp == 0) continue;
    u32 np = 0;
    const Probe p = probe_insert(table, tmask, s.fp, CntReg{&np});
    if (p.full) { atomicExch(err, (u32)ERR_TABLE_FULL); continue; }
    table[p.slot].meta = s.meta;                                 // every fingerprint occurs once in the old table: nobody else writes this slot
  }
}

// the same for the generator-side winner set of a sharded deep search (host_checker.hpp: wset_grow): every fingerprint occurs once, its epoch word travels with it
__global__ void k_wset_rehash(const u64* __restrict__ old_fp, const u32* __restrict__ old_epoch, u64 n_old, u64* fp, u32* epoch, u64 mask, u32* err) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n_old; i += stride) {
    const u64 f = old_fp[i];
    if (f == 0) continue;
    u64 j = wset_home(f, mask);
    bool done = false;
    for (u64 step = 0; step <= mask && step < 65536 && !done; step++, j = (j + 1) & mask)
      if (atomicCAS((unsigned long long*)&fp[j], 0ull, (unsigned long long)f) == 0) { epoch[j] = old_epoch[i]; done = true; }
    if (!done) atomicExch(err, (u32)ERR_TABLE_FULL);
  }
}

// clears the taken bit of every state of level >= min_level (vsr_deep.hpp: before a descent regenerates those levels again)
__global__ void k_table_untake(Slot* table, u64 n_slots, int min_level) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  const u64 stride = (u64)gridDim.x * blockDim.x;
  for (; i < n_slots; i += stride) {
    const u64 m = table[i].meta;
    if (table[i].fp != 0 && (m & META_TAKEN) && m != META_EMPTY && meta_level(m) >= min_level) table[i].meta = m & ~META_TAKEN;
  }
}

// empty seen-set: fp = 0, meta = all ones
__global__ void k_table_init(Slot* table, u64 slots) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < slots; i += (u64)gridDim.x * blockDim.x) {
    table[i].fp = 0;
    table[i].meta = META_EMPTY;
  }
}

// ---- sharded seen-set: the owner's side of one level -------------------------------------------------------------
// k_claim_batch: claim the received (fp, key) candidates in this rank's shard; rslot[i] = slot, or ~0 for a duplicate
// of an earlier level.
__global__ void k_claim_batch(Slot* table, u64 tmask, const u64* __restrict__ entries, u64 n, int level, u64* rslot,
                              LevelCtl* ctl) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  bool found_old, full;
  u32 np = 0;
  u64 slot = table_claim(table, tmask, entries[2 * i], entries[2 * i + 1], level, &found_old, CntReg{&np}, &full);
  if (full) raise_error(ctl, ERR_TABLE_FULL, entries[2 * i]);
  rslot[i] = found_old ? ~(u64)0 : slot;
  atomicAdd((unsigned long long*)&ctl->probes, (unsigned long long)np);
}
// k_verdict: after every claim of the level has landed: did candidate i win its slot?
__global__ void k_verdict(Slot* table, const u64* __restrict__ entries, const u64* __restrict__ rslot, u64 n, uint8_t* verdict) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u64 s = rslot[i];
  const u64 key = entries[2 * i + 1];
  verdict[i] = (s != ~(u64)0 && atomicCAS((unsigned long long*)&table[s].meta, (unsigned long long)key,
                                          (unsigned long long)(key | META_TAKEN)) == key) ? 1 : 0;
}
// Single-pass (fused) flavour of the owner's side: the candidate that INSERTS the fingerprint wins, every other candidate
// of the same fingerprint loses — the same rule k_expand<true> applies to the successors this rank owns itself, which are
// already in the table when the received candidates are claimed.  verdict[i] = 1 for winners; entries with fp == 0 are the
// unused tails of the senders' chunks.
__global__ void k_claim_batch_fused(Slot* table, u64 tmask, const u64* __restrict__ entries, u64 n, int level, uint8_t* verdict,
                                    LevelCtl* ctl) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 fp = entries[2 * i], key = entries[2 * i + 1];
  if (fp == 0) {
    verdict[i] = 0;
    return;
  }
  bool claimed, full;
  u64 prev_meta;
  u32 np = 0;
  table_claim_fused(table, tmask, fp, key, level, &claimed, &prev_meta, CntReg{&np}, &full);
  if (full) raise_error(ctl, ERR_TABLE_FULL, fp);
  verdict[i] = claimed ? 1 : 0;
  if (prev_meta != META_EMPTY && meta_level(prev_meta) == level && meta_auxkey(prev_meta) != meta_auxkey(key))
    atomicAdd((unsigned long long*)&ctl->ties, 1ull);
  for (int o = 32; o > 0; o >>= 1) np += __shfl_down(np, o);
  if ((threadIdx.x & 63) == 0 && np) atomicAdd((unsigned long long*)&ctl->probes, (unsigned long long)np);
}
// k_apply_verdict: the generator's side of the single-pass sharded level — candidate i of the bucket sent to one owner was
// written speculatively at state index cand_idx[i] (bits 56..63: violated-invariant mask); losers are withdrawn (invalid
// ref, exactly like an unused index), winners that violate an invariant are reported now.
__global__ void k_apply_verdict(const u64* __restrict__ entries, const u64* __restrict__ cand_idx, const uint8_t* __restrict__ verdict,
                                u64 n, u64* nx_off, u64* lvl_fp, LevelCtl* ctl, const WSet* wset /* deep passes: the winners are remembered */, int level) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 fp = entries[2 * i];
  if (fp == 0) return;
  const u64 e = cand_idx[i];
  const u64 idx = cand_index(e);
  if (verdict[i]) {
    if (wset && !wset_insert(wset, fp, level)) raise_error(ctl, ERR_TABLE_FULL, fp);
    const u32 bad = cand_bad(e);
    if (bad) {
      atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
      atomicOr(&ctl->viol_mask, bad);
    }
  } else {
    nx_off[idx] = 0;
    lvl_fp[idx] = 0;
  }
}
__device__ __forceinline__ u64 find_exact(const Slot* table, u64 tmask, u64 fp);
// The generator's side of the levels of a sharded run that exist in the seen-sets only (vsr_deep.hpp).  (Rounds 3-4 had an owner side for the
// regenerated levels too — k_regen_verdict granted the candidate whose key is the slot's final meta word; since round 5 a rank regenerates what
// its winner set holds and asks nobody: WSet, above.)
// k_count_verdict, generator side of a virtual level: nothing was written, so the winners among the announced successors are only
// counted — new states, checksums of their fingerprints, largest bag (cand_idx: bag size | violated-invariant mask << 56), violators.
__global__ void k_count_verdict(const u64* __restrict__ entries, const u64* __restrict__ cand_idx, const uint8_t* __restrict__ verdict, u64 n,
                                u64* pending, u64 pending_cap, LevelCtl* ctl, const WSet* wset, int level) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 cnt = 0, fx = 0, fs = 0, bag = 0;
  if (i < n && entries[2 * i] != 0 && verdict[i]) {
    const u64 fp = entries[2 * i], e = cand_idx[i];
    if (wset && !wset_insert(wset, fp, level)) raise_error(ctl, ERR_TABLE_FULL, fp);
    cnt = 1; fx = fp; fs = fp; bag = cand_index(e);
    const u32 bad = cand_bad(e);
    if (bad) {
      const u64 k = atomicAdd((unsigned long long*)&ctl->n_pending, 1ull);
      if (k < pending_cap) { pending[2 * k] = fp; pending[2 * k + 1] = entries[2 * i + 1]; }
      atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
      atomicOr(&ctl->viol_mask, bad);
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_down(cnt, o);
    fx ^= __shfl_down(fx, o);
    fs += __shfl_down(fs, o);
    const u64 t = __shfl_down(bag, o);
    bag = t > bag ? t : bag;
  }
  if ((threadIdx.x & 63) == 0 && cnt) {
    atomicAdd((unsigned long long*)&ctl->n_new, (unsigned long long)cnt);
    atomicXor((unsigned long long*)&ctl->fp_xor, (unsigned long long)fx);
    atomicAdd((unsigned long long*)&ctl->fp_sum, (unsigned long long)fs);
    atomicMax((unsigned long long*)&ctl->max_bag, (unsigned long long)bag);
  }
}
// k_partition: end of the replicated phase of a sharded run (every rank explored the small early levels by itself): keep
// the states of the current frontier this rank owns, withdraw the others.  counter[0] = states kept.
__global__ void k_partition(u64* off, u64* lvl_fp, u64 n, int rank, int world, u64* counter) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 keep = 0;
  if (i < n && off[i] != 0) {
    if (owner_of(lvl_fp[i], world) == rank) keep = 1;
    else {
      off[i] = 0;
      lvl_fp[i] = 0;
    }
  }
  for (int o = 32; o > 0; o >>= 1) keep += __shfl_down(keep, o);
  if ((threadIdx.x & 63) == 0 && keep) atomicAdd((unsigned long long*)counter, (unsigned long long)keep);
}
// k_append_fixup: records received from a peer were copied to nx_words[base_words ..); publish their offsets and
// fingerprints at state indices n0 .. (the predecessor pointers live in the owners' table slots and do not move)
__global__ void k_append_fixup(u64* nx_off, u64* lvl_fp, const u64* __restrict__ rel_off, const u64* __restrict__ fps, u64 n,
                               u64 base_words) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  nx_off[i] = rel_off[i] ? rel_off[i] + (base_words << 8) : 0;   // refs are (word offset << 8 | length); 0 stays invalid
  lvl_fp[i] = fps[i];
}

// k_export: rebalancing between ranks — move the valid records of the index window [first, first + n) of the (new) frontier
// into contiguous send streams and invalidate them here (ref = 0, fp = 0).  One wave per block, lane per index; records
// are copied cooperatively (lane k moves word k).  counters[0] = records exported, counters[1] = words exported.
__global__ void __launch_bounds__(64)
k_export(const u64* __restrict__ nx_words, u64* nx_off, u64* lvl_fp, u64 first, u64 n,
         u64* out_words, u64 out_words_cap, u64* out_off, u64* out_fp, u64 out_cap, u64* counters, u32* err) {
  const int lane = threadIdx.x;
  const u64 i = (u64)blockIdx.x * 64 + lane;
  u64 ref = 0;
  if (i < n) ref = nx_off[first + i];
  const bool valid = ref != 0;
  const int len = valid ? (int)(ref & 255) : 0;
  const u64 wmask = __ballot(valid);
  if (wmask == 0) return;
  int incl = len;
  for (int o = 1; o < 64; o <<= 1) {
    int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  const int total = __builtin_amdgcn_readlane(incl, 63);
  const int nval = __popcll(wmask);
  u64 kb = 0, wb = 0;
  if (lane == 0) {
    kb = atomicAdd((unsigned long long*)&counters[0], (unsigned long long)nval);
    wb = atomicAdd((unsigned long long*)&counters[1], (unsigned long long)total);
  }
  kb = readlane64(kb, 0);
  wb = readlane64(wb, 0);
  if (kb + (u64)nval > out_cap || wb + (u64)total > out_words_cap) {
    if (lane == 0) atomicExch(err, (u32)ERR_FRONTIER_FULL);
    return;
  }
  const u64 dst = wb + (u64)(incl - len);
  for (int w = 0; w < 64; w++) {
    const int ln = __builtin_amdgcn_readlane(len, w);
    if (ln == 0) continue;
    const u64 s = readlane64(ref, w) >> 8, d = readlane64(dst, w);
    for (int k = lane; k < ln; k += 64) out_words[d + k] = nx_words[s + k];
  }
  if (valid) {
    const u64 k = kb + (u64)__popcll(wmask & (((u64)1 << lane) - 1));
    out_off[k] = (dst << 8) | (u64)len;
    out_fp[k] = lvl_fp[first + i];
    nx_off[first + i] = 0;
    lvl_fp[first + i] = 0;
  }
}

// ---- checkpoint / recover (TLC's FPSet.beginChkpt / recover): the seen-set travels as its occupied slots only -----------
// k_table_export: copy the occupied slots of [first, first + n) to out (any order), counter[0] = how many.
__global__ void k_table_export(const Slot* __restrict__ table, u64 first, u64 n, Slot* out, u64 out_cap, u64* counter) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  Slot s;
  s.fp = 0;
  s.meta = 0;
  if (i < n) s = table[first + i];
  if (s.fp != 0) {
    const u64 k = wave_alloc(counter);
    if (k < out_cap) out[k] = s;
  }
}
// k_table_import: insert (fp, meta) pairs into a table of any size (the probe sequence depends on the size, the content does not)
__global__ void k_table_import(Slot* table, u64 tmask, const Slot* __restrict__ in, u64 n, u32* err) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 np = 0;
  const Probe p = probe_insert(table, tmask, in[i].fp, CntReg{&np});
  if (p.full) {
    atomicExch(err, (u32)ERR_TABLE_FULL);
    return;
  }
  table[p.slot].meta = in[i].meta;
}

// The failing successors a probe-only pass (k_expand<.., 6>) wrote down as (parent index, ordinal): fingerprinted and looked up here, after the pass — the ones
// that are states of an earlier level are dropped, the rest become the (fingerprint, key) pairs the probe passes of rounds 3-5 produced inside k_expand, in
// `out` (out_n counts them; viol_fp / viol_mask of the control block as before).  A handful of entries per pass (README: 8 of 3.8e9 successors): one lane each,
// records read from HBM, the model's constants at run time.
template <int MODEL>
__global__ void k_probe_resolve(Model M, const u64* __restrict__ src_words, const u64* __restrict__ src_off, const u64* __restrict__ list, u64 n, const Slot* table,
                                u64 tmask, int level, u64* out, u64 out_cap, unsigned long long* out_n, LevelCtl* ctl) {
  typedef ModelOps<MODEL> Ops;
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 origin = list[2 * i];
  const u64 ref = src_off[origin_pidx(origin)];
  const u64* rec = src_words + (ref >> 8);
  Delta D;
  if (!Ops::template gen_<false>(M, rec, origin_ord(origin), D) || D.err) { raise_error(ctl, D.err ? D.err : (int)ERR_INTERNAL, origin); return; }
  const int bad = Ops::invariants(M, rec, D);
  if (!bad) return;
  u64 Hc[6];
  Ops::hash_child_(M, rec, D, Hc);
  u64 fp, pfp;
  u32 ak, pak;
  canonical_fp(M, D.hdr, Hc, &fp, &ak);
  canonical_fp(M, rec[0], rec + M.h0, &pfp, &pak);
  u64 m = META_EMPTY;
  u32 np = 0;
  const bool old_state = probe_lookup(table, tmask, fp, &m, CntReg{&np}) && meta_level(m) < level;
  if (np) atomicAdd((unsigned long long*)&ctl->probes, (unsigned long long)np);             // (the few lookups a probe level makes: vsrmc_level_info.probes)
  if (old_state) return;                                        // a state of an earlier level: TLC drops it as seen
  const u64 k = atomicAdd(out_n, 1ull);
  if (k < out_cap) { out[2 * k] = fp; out[2 * k + 1] = meta_make(level, ak, pfp); }
  atomicMin((unsigned long long*)&ctl->viol_fp, (unsigned long long)fp);
  atomicOr(&ctl->viol_mask, (u32)bad);
}

// the winner set of a sharded deep search as (fingerprint, level) pairs (checkpoint of a sharded search beyond its record buffers: host_checkpoint.hpp) ...
__global__ void k_wset_export(const u64* __restrict__ fp, const u32* __restrict__ epoch, u64 first, u64 n, Slot* out, u64 out_cap, u64* counter) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  Slot s;
  s.fp = 0;
  s.meta = 0;
  if (i < n) { s.fp = fp[first + i]; s.meta = (u64)(epoch[first + i] >> 23); }
  if (s.fp != 0) {
    const u64 k = wave_alloc(counter);
    if (k < out_cap) out[k] = s;
  }
}
// ... and back into an empty set (any size): the descent counter of every entry starts at 0 again, like the recovered checker's
__global__ void k_wset_import(const WSet* w, const Slot* __restrict__ in, u64 n, u32* err) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (!wset_insert(w, in[i].fp, (int)in[i].meta)) atomicExch(err, (u32)ERR_TABLE_FULL);
}

// Seed the search with the initial state (ModelChecker.doInit): record already in frontier slot 0.
__global__ void k_seed(Model M, const u64* rec, Slot* table, u64 tmask, u64* lvl_fp, LevelCtl* ctl) {
  if (threadIdx.x || blockIdx.x) return;
  u64 fp;
  u32 ak;
  canonical_fp(M, rec[0], rec + M.h0, &fp, &ak);
  u64 key = meta_make(1, ak, 0);
  bool found_old, full;
  u32 np = 0;
  table_claim(table, tmask, fp, key, 1, &found_old, CntReg{&np}, &full);
  lvl_fp[0] = fp;
  ctl->n_new = 1;
  ctl->words_new = (u64)(M.fixed + hdr_nmsg(rec[0]));
}

// Slot of the state whose fingerprint has the low bits `pfp` and whose level is `level` (the parent a meta word names): linear
// probing stores a fingerprint at or after its home slot fp & mask with no empty slot in between, and the home slot only needs
// the bits the meta word keeps (table_log2 <= 36 < 45).  ~0 = no such state.
// *n_match (optional) = how many states of that level carry those 45 bits: the whole probe run (up to the first empty slot) is
// scanned, and a second match means the predecessor pointer is AMBIGUOUS (about level size / 2^45 per step: 2e-5 for the 7.9e8
// states of the README configuration's level 23) — the caller reports it instead of walking on through the first match.
__device__ __forceinline__ u64 find_by_low_bits(const Slot* table, u64 tmask, u64 pfp, int level, u32* n_match = nullptr) {
  u64 i = pfp & tmask, first = ~(u64)0;
  u32 n = 0;
  for (u64 step = 0; step <= tmask; step++, i = (i + 1) & tmask) {
    const u64 f = table[i].fp;
    if (f == 0) break;
    if ((f & PFP_MASK) == pfp && meta_level(table[i].meta) == level) {
      if (n++ == 0) first = i;
      if (!n_match) break;
    }
  }
  if (n_match) *n_match = n;
  return first;
}
__device__ __forceinline__ u64 find_exact(const Slot* table, u64 tmask, u64 fp) {
  u64 i = fp & tmask;
  for (u64 step = 0; step <= tmask; step++, i = (i + 1) & tmask) {
    const u64 f = table[i].fp;
    if (f == 0) return ~(u64)0;
    if (f == fp) return i;
  }
  return ~(u64)0;
}
// TLCTrace.getTrace, backwards half: follow the predecessor pointers in the seen-set from the state with fingerprint `fp` of
// level `level` back to Init.  fps[l-1] = fingerprint of the path's level-l state; fps[level] = status: 0 ok, 1 broken chain (cannot
// happen on a table the search itself filled), 2 | l << 8 | matches << 16 = the pointer of the path's level-(l+1) state is ambiguous.
__global__ void k_trace_walk(const Slot* table, u64 tmask, u64 fp, int level, u64* fps) {
  if (threadIdx.x || blockIdx.x) return;
  u64 slot = find_exact(table, tmask, fp);
  fps[level] = 0;
  for (int l = level; l >= 1; l--) {
    if (slot == ~(u64)0 || meta_level(table[slot].meta) != l) {
      fps[0] = 0;
      fps[level] = 1;
      return;
    }
    fps[l - 1] = table[slot].fp;
    if (l > 1) {
      u32 n = 0;
      slot = find_by_low_bits(table, tmask, meta_pfp(table[slot].meta), l - 1, &n);
      if (n > 1) {
        fps[level] = 2 | ((u64)(l - 1) << 8) | ((u64)n << 16);
        return;
      }
    }
  }
}
// one step of the same walk, for walks that cross ranks (sharded runs): mode 0: exact fingerprint -> (fp, meta); mode 1: the 45
// low fingerprint bits a child keeps of its parent + the parent's level -> (fp, meta).  out[0] = found (0 / 1), out[1] = fingerprint, out[2] = meta.
__global__ void k_table_lookup(const Slot* table, u64 tmask, u64 key, int level, int mode, u64* out) {
  if (threadIdx.x || blockIdx.x) return;
  u32 n = 1;
  const u64 slot = mode == 0 ? find_exact(table, tmask, key) : find_by_low_bits(table, tmask, key & PFP_MASK, level, &n);
  out[0] = slot != ~(u64)0 ? (mode == 0 ? 1 : (u64)n) : 0;    // by low bits: the number of matching states of this shard (> 1: ambiguous)
  out[1] = slot != ~(u64)0 ? table[slot].fp : 0;
  out[2] = slot != ~(u64)0 ? table[slot].meta : 0;
}

// flags[i] = 1 if fingerprint fps[i] is a state of a level below `level` (a streamed probe collects violating successors while the
// level above them is still being inserted: the ones that turn out to be states of that level — same VIEW fingerprint, i.e. the
// same state for the search, whatever the invariant says about this copy's aux variables — are dropped afterwards)
__global__ void k_table_seen(const Slot* table, u64 tmask, const u64* __restrict__ fps, u64 n, int level, u64* flags) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 slot = find_exact(table, tmask, fps[i]);
  flags[i] = (slot != ~(u64)0 && meta_level(table[slot].meta) < level) ? 1 : 0;
}

// k_select: indices of the frontier records in which at least one instance of an action of `action_mask` (bit a = action id a)
// is enabled — TLC's per-action coverage, as a filter.  One lane per record, guards only (guard_slot: the same statement of the
// guards k_expand enumerates with).  counters[0] = matching records (all of them), out_idx holds the first out_cap that arrived.
template <int MODEL>
__global__ void k_select(Model M, const u64* __restrict__ fr_words, const u64* __restrict__ fr_off, u64 n, u32 action_mask, u64* out_idx,
                         u64 out_cap, u64* counters) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 ref = fr_off[i];
  if (!ref) return;
  const u64* rec = fr_words + (ref >> 8);
  const int nslots = M.m0 + hdr_nmsg(rec[0]);
  bool hit = false;
  for (int slot = 0; slot < nslots && !hit; slot++) {
    int kind0 = 0;
    const u32 mask = ModelOps<MODEL>::guard(M, rec, slot, &kind0);
    if ((mask & 1u) && ((action_mask >> kind0) & 1u)) hit = true;
    if ((mask & ~1u) && ((action_mask >> ModelOps<MODEL>::other_kind(kind0)) & 1u)) hit = true;
  }
  if (hit) {
    const u64 k = wave_alloc(&counters[0]);
    if (k < out_cap) out_idx[k] = i;
  }
}

// index of the record of the newest level whose fingerprint is `fp`
__global__ void k_find_fp(const u64* lvl_fp, u64 n, u64 fp, u64* out_idx) {
  u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && lvl_fp[i] == fp) atomicMin((unsigned long long*)out_idx, (unsigned long long)i);
}

// -----------------------------------------------------------------------------------------------------------------
// Standalone FPSet batch operations (tlc2.tool.fp.FPSet.putBlock / containsBlock)
// -----------------------------------------------------------------------------------------------------------------
__global__ void k_fpset_put(Slot* table, u64 tmask, const u64* fps, u64 n, uint8_t* was_present, u64* size, u32* err) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u64 fp = fps[t] ? fps[t] : 1;
  u64 i = fp & tmask;
  for (u64 step = 0; step <= tmask; step++, i = (i + 1) & tmask) {
    u64 cur = __hip_atomic_load(&table[i].fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (cur == 0) {
      cur = atomicCAS((unsigned long long*)&table[i].fp, 0ull, (unsigned long long)fp);
      if (cur == 0) {
        // first claimer of a batch wins: later equal keys in the same batch report "present"
        table[i].meta = 0;
        was_present[t] = 0;
        atomicAdd((unsigned long long*)size, 1ull);
        return;
      }
    }
    if (cur == fp) {
      was_present[t] = 1;
      return;
    }
  }
  was_present[t] = 1;
  atomicExch(err, (u32)ERR_TABLE_FULL);
}

__global__ void k_fpset_contains(const Slot* table, u64 tmask, const u64* fps, u64 n, uint8_t* present) {
  u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n) return;
  u64 fp = fps[t] ? fps[t] : 1;
  u64 i = fp & tmask;
  for (u64 step = 0; step <= tmask; step++, i = (i + 1) & tmask) {
    u64 cur = table[i].fp;
    if (cur == fp) { present[t] = 1; return; }
    if (cur == 0) break;
  }
  present[t] = 0;
}

// -----------------------------------------------------------------------------------------------------------------
// k_successors: all successors of a batch of records in ordinal order (Tool.getNextStates over every action).
// One lane per parent; out_meta has 8 words per successor:
//   [parent index, ordinal, action id, fingerprint, auxkey, violated-invariant mask, error code, word offset]
// -----------------------------------------------------------------------------------------------------------------
template <int MODEL>
__global__ void k_successors(Model M, const u64* words, const u64* off, u64 n, u64* out_words, u64 out_words_cap,
                             u64* out_meta, u64 out_cap, u64* counters /* [0] successors, [1] words, [2] overflow */) {
  u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const u64* rec = words + off[p];
  int nords = ord_count(M, hdr_nmsg(rec[0]));
  for (int ord = 0; ord < nords; ord++) {
    Delta D;
    if (!ModelOps<MODEL>::template gen_<true>(M, rec, ord, D)) continue;
    ModelOps<MODEL>::template gen_<false>(M, rec, ord, D);
    u64 Hc[6];
    ModelOps<MODEL>::hash_child_(M, rec, D, Hc);
    u64 fp;
    u32 ak;
    canonical_fp(M, D.hdr, Hc, &fp, &ak);
    int clen = M.fixed + hdr_nmsg(D.hdr);
    u64 k = atomicAdd((unsigned long long*)&counters[0], 1ull);
    u64 w = atomicAdd((unsigned long long*)&counters[1], (unsigned long long)clen);
    if (k >= out_cap || w + clen > out_words_cap) {
      counters[2] = 1;
      continue;
    }
    if (!D.err) write_child_serial(M, rec, D, Hc, out_words + w);
    else for (int q = 0; q < clen; q++) out_words[w + q] = 0;
    u64* m = out_meta + 8 * k;
    m[0] = p;
    m[1] = (u64)ord;
    m[2] = (u64)D.action;
    m[3] = fp;
    m[4] = ak;
    m[5] = D.err ? 0 : (u64)ModelOps<MODEL>::invariants(M, rec, D);
    m[6] = (u64)D.err;
    m[7] = w;
  }
}

// -----------------------------------------------------------------------------------------------------------------
// k_simulate — TLC's simulation mode (`-simulate`, README:22 of the reference: "has a good chance of hitting it sooner"):
// every lane is one random walker.  A walk starts at Init, takes up to max_depth steps, each chosen uniformly among the
// enabled (action, binding) instances of the current state, checks the invariants after every step and restarts at the
// depth limit or in a terminal state.  No seen-set, no fingerprints.  The first walker that violates an invariant
// publishes its ordinal sequence (replayed into a TLC-style trace by k_replay).
// -----------------------------------------------------------------------------------------------------------------
struct SimCtl {
  u32 found;          // 0 = searching, 1 = a violating walk was published
  u32 viol_mask;
  u32 viol_depth;     // number of steps of the violating walk
  u32 pad;
  u64 steps;          // steps taken by all walkers
  u64 walks;          // walks started
  u32 ords[512];      // the violating walk
};

__device__ __forceinline__ u64 sim_rng(u64* s) {   // xorshift64*
  u64 x = *s;
  x ^= x >> 12;
  x ^= x << 25;
  x ^= x >> 27;
  *s = x;
  return x * 0x2545F4914F6CDD1DULL;
}

template <int SPEC>
__global__ void __launch_bounds__(64)
k_simulate(Model Marg, const u64* __restrict__ init_rec, int init_len, u64* walker_words, int stride, u32* walker_depth,
           u16* walker_ords, u64* walker_rng, u32 n_walkers, int max_depth, int steps_per_launch, SimCtl* ctl) {
  Model M = Marg;
  specialise<SPEC>(M, Marg);
  typedef ModelOps<SPEC / 1000> Ops;
  const u32 t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_walkers) return;
  u64* w = walker_words + (u64)t * stride;
  u16* my_ords = walker_ords + (u64)t * max_depth;
  u64 rng = walker_rng[t];
  u32 depth = walker_depth[t];
  u64 steps = 0, walks = 0;
  for (int it = 0; it < steps_per_launch; it++) {
    if (__hip_atomic_load(&ctl->found, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (depth == 0xFFFFFFFFu || depth >= (u32)max_depth) {     // (re)start at Init
      for (int k = 0; k < init_len; k++) w[k] = init_rec[k];
      depth = 0;
      walks++;
    }
    // count the enabled instances, then draw one
    const int nslots = M.m0 + hdr_nmsg(w[0]);
    int total = 0;
    for (int slot = 0; slot < nslots; slot++) {
      int kind0;
      total += __builtin_popcount(Ops::guard(M, (const u64*)w, slot, &kind0));
    }
    if (total == 0) {                                           // terminal state: the walk ends (TLC -deadlock)
      depth = 0xFFFFFFFFu;
      continue;
    }
    int pick = (int)(sim_rng(&rng) % (u64)total), ord = -1;
    for (int slot = 0; slot < nslots && ord < 0; slot++) {
      int kind0;
      u32 mask = Ops::guard(M, (const u64*)w, slot, &kind0);
      const int c = __builtin_popcount(mask);
      if (pick >= c) { pick -= c; continue; }
      while (pick--) mask &= mask - 1;
      const int k = __ffs((int)mask) - 1;
      ord = slot < M.m0 ? slot : M.m0 + (slot - M.m0) * (M.R + 1) + k;
    }
    Delta D;
    if (ord < 0 || !Ops::template gen_<false>(M, (const u64*)w, ord, D)) {
      atomicExch(&ctl->viol_mask, 0x80000000u | (u32)ERR_INTERNAL);
      atomicExch(&ctl->found, 2u);
      break;
    }
    if (D.err) {                                                // evaluation / representation error: report like a violation
      if (atomicCAS(&ctl->found, 0u, 2u) == 0u) {
        ctl->viol_mask = 0x80000000u | (u32)D.err;
        ctl->viol_depth = depth;
        for (u32 k = 0; k < depth && k < 512; k++) ctl->ords[k] = my_ords[k];
      }
      break;
    }
    const int bad = Ops::invariants(M, (const u64*)w, D);
    // apply the step in place
    const int plen = M.fixed + hdr_nmsg(w[0]);
    w[0] = D.hdr;
    u64* pb = w + 1 + (D.r - 1) * M.wpr;
    pb[0] = D.rep[0];
    if (M.wpr > 1) pb[1] = D.rep[1];
    if (M.wpr > 2) pb[2] = D.rep[2];
    if (M.wpr > 3) pb[3] = D.rep[3];
    int a = 0;
#pragma unroll
    for (int k = 0; k < VSR_NSLOT; k++)
      if ((D.used >> k) & 1) {
        if (D.pj(k) >= 0) w[M.fixed + D.pj(k)] = D.pnew[k];
        else w[plen + (a++)] = D.pnew[k];
      }
    my_ords[depth] = (u16)ord;
    depth++;
    steps++;
    if (bad) {
      if (atomicCAS(&ctl->found, 0u, 1u) == 0u) {
        ctl->viol_mask = (u32)bad;
        ctl->viol_depth = depth;
        for (u32 k = 0; k < depth && k < 512; k++) ctl->ords[k] = my_ords[k];
      }
      break;
    }
  }
  walker_rng[t] = rng;
  walker_depth[t] = depth;
  atomicAdd((unsigned long long*)&ctl->steps, (unsigned long long)steps);
  atomicAdd((unsigned long long*)&ctl->walks, (unsigned long long)walks);
}

// k_hash_records: fill in the H words of device-layout records (used when records enter through the C ABI)
template <int MODEL>
__global__ void k_hash_records(Model M, u64* words, const u64* off, u64 n) {
  u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  u64* rec = words + off[p];
  u64 H[6];
  ModelOps<MODEL>::hash_full_(M, (const u64*)rec, H);
  for (int i = 0; i < M.np; i++) rec[M.h0 + i] = H[i];
}

// k_replay: re-execute a path starting from the record at out_words[0..).  Record t+1 = the successor of record t named by
// ords[t] (fps == nullptr: a path of ordinals — simulation walks), or the successor whose fingerprint is fps[t + 1] (a path of
// fingerprints — what a trace walk through the seen-set yields; of several instances with that successor the smallest ordinal is
// taken, the state is the same).  out_meta per step: [action id, fingerprint, invariant mask, error]; ords_out[t] (may be null) =
// the ordinal taken.
template <int MODEL>
__global__ void k_replay(Model M, u64* out_words, u64* out_off, const u32* ords, int nsteps, u64* out_meta, const u64* fps, u32* ords_out) {
  if (threadIdx.x || blockIdx.x) return;
  u64 pos = 0;
  out_off[0] = 0;
  for (int t = 0; t < nsteps; t++) {
    const u64* rec = out_words + pos;
    u64 next = pos + (u64)(M.fixed + hdr_nmsg(rec[0]));
    Delta D;
    u64 Hc[6];
    u64 fp = 0;
    u32 ak = 0;
    bool en = false;
    int ord = 0;
    if (fps) {
      const int nords = ord_count(M, hdr_nmsg(rec[0]));
      for (ord = 0; ord < nords && !en; ord++) {
        if (!ModelOps<MODEL>::template gen_<true>(M, rec, ord, D)) continue;
        ModelOps<MODEL>::template gen_<false>(M, rec, ord, D);
        if (D.err) continue;
        ModelOps<MODEL>::hash_child_(M, rec, D, Hc);
        canonical_fp(M, D.hdr, Hc, &fp, &ak);
        if (fp == fps[t + 1]) {
          en = true;
          break;
        }
      }
    } else {
      ord = (int)ords[t];
      en = ModelOps<MODEL>::template gen_<true>(M, rec, ord, D);
      if (en) {
        ModelOps<MODEL>::template gen_<false>(M, rec, ord, D);
        ModelOps<MODEL>::hash_child_(M, rec, D, Hc);
        canonical_fp(M, D.hdr, Hc, &fp, &ak);
      }
    }
    if (!en) {
      out_meta[4 * t + 3] = 0xFFFF;
      out_off[t + 1] = next;
      for (int k = t + 1; k < nsteps; k++) out_off[k + 1] = next;
      return;
    }
    if (ords_out) ords_out[t] = (u32)ord;
    write_child_serial(M, rec, D, Hc, out_words + next);
    out_meta[4 * t + 0] = (u64)D.action;
    out_meta[4 * t + 1] = fp;
    out_meta[4 * t + 2] = (u64)ModelOps<MODEL>::invariants(M, rec, D);
    out_meta[4 * t + 3] = (u64)D.err;
    out_off[t + 1] = next;
    pos = next;
  }
}

}  // namespace vsr

// vsr_deep.hpp — the search beyond the last level whose records fit the HBM (included by vsrmc.hip after the level phases).
//
// TLC role replaced: DiskStateQueue + DiskFPSet — what lets TLC go on when the frontier outgrows memory.  Here nothing leaves the HBM:
// a level beyond the last materialised one ("base", level L) exists only as seen-set entries (16 B per state instead of ~350 B of
// record), and its records are REGENERATED from the base level whenever they are needed, slice by slice, never all at once:
//
//   pass 1            level L expanded, MODE_INSERT                  -> level L+1 in the seen-set (exact count, final min-merged keys)
//   pass j (j >= 2)   descend from the base:
//                       for each slice of L:        MODE_REGEN       = that slice's part of level L+1 (the successor whose key IS the
//                         for each slice of L+1:    MODE_REGEN         slot's meta word takes it with a compare-and-swap: exactly once)
//                           ...                                      = ... of level L+j-1
//                             for each sub-slice:   MODE_NORMAL      = new level-(L+j) states: inserted, counted, checked, checksummed,
//                                                                      written to a scratch buffer and dropped after
//                                                   MODE_PROBE       = invariants of THEIR successors (level L+j+1), nothing inserted
//   Level L+j-1 is complete in the seen-set before the first level-(L+j) state is inserted (else a state of L+j-1 met first as a
//   successor of L+j-1 would be filed one level too deep) — that is what the separate passes are for.  Level L+j is NOT complete
//   while level L+j+1 is probed, and that matters: the invariants read aux variables outside the VIEW (VSR.tla:102-104), so a
//   successor with the fingerprint of a level-(L+j) state that is not inserted yet can look violating although the search never
//   visits it (TLC drops it as seen).  The probe passes therefore only COLLECT violating successors (fingerprint, key); when the
//   level is complete the ones that are states of a level < L+j+1 are dropped (k_table_seen) and the smallest remaining fingerprint
//   is the violation.
// Every pass inserts one more level and probes the one after it, so the search ROLLS ON past the memory horizon: the cost of level
// L+j is the re-expansion of levels L .. L+j-1, about 1 + 1/g + 1/g^2 .. = 2.2 x its own expansion at the growth g = 1.8 of this
// model — until the seen-set is full (16 B per state: 1e10 states on one MI355X), a level comes back empty (the search is
// exhausted) or an invariant fails.  A regenerated state takes its slot with key -> key | taken; the taken bits of the levels
// beyond the base are cleared by one pass over the table before every descent (k_table_untake).
//
// vsrmc_checker_probe2 / _probe3 (rounds 2-3: virtual + probed, virtual + streamed + probed) are two and three steps of this.
#pragma once

namespace {

typedef DeepLevelRec DeepLevel;   // a level beyond the base that is complete in the seen-set (vsrmc_checker::deep_lv)

// how large the next slice of a source may be so that what it yields fits the target buffer: the first one for the worst case
// (every generated successor new, of the largest size), later ones by the largest yield per source index seen so far (x 3 headroom;
// a target loses up to a quarter to the blocks' unfinished chunks)
struct Slicer {
  u64 cap_n, cap_w, g1, stride;
  u64 cap_c = ~(u64)0;          // sharded: parents whose successors fit one owner's candidate bucket (half of it: the blocks' chunks)
  double yield_n = 0, yield_w = 0;
  u64 worst() const {
    const u64 sz = std::min<u64>(std::min<u64>(cap_n / (4 * g1), cap_w / (4 * g1 * stride)), cap_c);
    return std::max<u64>(128, sz & ~(u64)127);
  }
  u64 next() const {
    if (yield_n <= 0) return worst();
    const double sz = std::min((double)cap_n / (3.0 * std::max(yield_n, 1e-3)), (double)cap_w / (3.0 * std::max(yield_w, 1e-3)));
    return std::max<u64>(worst(), std::min<u64>(cap_c, std::max<u64>(128, (u64)std::min(sz, 1e15) & ~(u64)127)));
  }
  void expect(double per_n, double per_w) { yield_n = std::max(yield_n, per_n); yield_w = std::max(yield_w, per_w); }
  void observe(u64 n_src, u64 n_out, u64 w_out) {
    if (!n_src) return;
    yield_n = std::max(yield_n, (double)n_out / (double)n_src);
    yield_w = std::max(yield_w, (double)w_out / (double)n_src);
  }
};

struct DeepRun;
// A sharded run (vsr_shard_loop.hpp) supplies these: every pass is then a collective — successors owned by other ranks are announced to
// their owners, claimed / answered there, and the verdicts applied here — and every loop runs as long as ANY rank has work left.
struct DeepIo {
  void* ctx = nullptr;
  int world = 1;
  u64 cand_cap = 0;              // candidates per owner the exchange buffers hold
  int (*pass)(void* ctx, const u64* sw, const u64* so, u64 n, u64 p_off, int level, int mode, u64 bag, const PassDst* dst) = nullptr;
  int (*any)(void* ctx, u64* flag) = nullptr;                   // *flag = its maximum over the ranks
  int (*reduce)(void* ctx, DeepRun* R) = nullptr;               // R->ins / R->prb: this rank's figures -> the level's
  int (*resolve)(void* ctx, DeepRun* R) = nullptr;              // the probe's violating successors shown to their owners
};

struct DeepRun {
  vsrmc_checker* c;
  const DeepIo* io = nullptr;
  int base;                    // level L: the newest materialised level
  int last_regen;              // levels base+1 .. last_regen are regenerated (they are complete in the seen-set)
  bool insert;                 // the level last_regen + 1 is inserted (MODE_NORMAL into scratch); false: it is only probed (probe2)
  vsrmc_level_info ins, prb;   // figures of the inserted and of the probed level
  u64 viol_ins = ~(u64)0;
  u32 mask_ins = 0, mask_prb = 0;
  std::vector<u64> bad;        // (fingerprint, key) of the violating successors the probe passes saw
  u64 launches = 0, n_slices = 0, n_subs = 0;
  u64* d_sum = nullptr;
  bool rebase = false;         // deep_rebase: the regenerated slices of level last_regen are not probed but moved, packed, into the idle record buffer
  u64* d_cnt = nullptr;        //   [0] records, [1] words moved so far (k_export's running cursors)
  u32* d_err = nullptr;
  int err = 0;                 // sharded: a failure of this rank's LOCAL work between two collectives (a probe pass, a checksum) — carried to the next
                               // "any slice left?" exchange, where every rank learns of it and all of them leave together (nobody waits in a collective)
  int probed_level() const { return last_regen + (insert ? 2 : 1); }
};

// The scratch buffers of a descent: buffer k takes what nesting level k yields (k = 0: the slices of level base + 1, ..).  Every state of every
// regenerated level passes through its buffer once per descent, so the launches of a descent number about sum_k (level size / buffer k's capacity):
// EQUAL buffers minimise that (a geometric series — round 3's rule — starves the deepest levels of a long descent: the analysis model's run to
// depth 33 took 570 s that way).  Rounds 3-4: buffer 0 = the WHOLE idle record buffer, buffers 1 .. count = equal shares of what is free on the device
// (the reserve autosize_options leaves: a third of a record buffer) — at twelve nested levels 59 GB for buffer 0 and 1.4 GB for each of the others.
// Round 5: the idle record buffer and the reserve are ONE pool, cut into count + 1 equal buffers (the first ones pieces of the idle record buffer,
// the rest — and every piece's index arrays — allocated from the reserve) whenever that lowers the sum above.  Planned at the start of a pass,
// when the buffers are empty; re-planned (freed and allocated anew) when a pass needs more levels than the plan has, or the other kind of plan:
// a re-basing descent (deep_rebase) writes the new base level INTO the front of the idle record buffer (`front` words, known exactly), uses no
// buffer 0, and cuts its buffers 1 .. count out of what lies behind that front and the reserve.
void deep_free_scratch(vsrmc_checker* c) {
  for (PassDst& B : c->scratch) {
    if (B.words && B.own_words) (void)hipFree(B.words);
    if (B.off && B.own_index) (void)hipFree(B.off);
    if (B.fp && B.own_index) (void)hipFree(B.fp);
  }
  c->scratch.clear();
  c->scratch0 = PassDst();
  c->scratch_carved = false;
  c->scratch_buf = -1;
  c->scratch_front = 0;
}

int deep_plan_scratch(vsrmc_checker* c, int count, bool carve = true, u64 front = 0) {
  if (count < 0) count = 0;
  if (count > 400) return fail(VSRMC_E_REP, "deep search: more than 400 nested levels");
  const int nxt = c->cur ^ 1;
  if ((c->host_frontier >> nxt) & 1) carve = false;              // (records in pinned host memory: scratch stays on the device)
  front = (front + 63) & ~(u64)63;
  // (a carved plan holds raw pointers into the record buffer that was idle when it was made: after a swap — a re-basing, a reset, a manual deepen at
  // another level — those pieces alias the frontier the descent reads, so such a plan is never reused)
  const bool same_buf = !c->scratch_carved || c->scratch_buf == nxt;
  if ((size_t)count <= c->scratch.size() && c->scratch_front == front && same_buf && (c->scratch_carved == carve || (count == 0 && !carve))) return 0;
  deep_free_scratch(c);
  if (count == 0) return 0;                                       // one level below the base: buffer 0 (the idle record buffer) is all a pass needs
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const char* share_env = std::getenv("VSRMC_AUTOSIZE_SHARE");   // several checkers on one device (tests: ranks sharing a GPU)
  const double share = share_env ? std::max(1.0, std::atof(share_env)) : 1.0;
  const double reserve = std::max(0.0, (double)free_b - 1.0e9) / share;   // bytes
  if (front >= c->words_cap(nxt)) carve = false;
  const double idle_w = (double)(c->words_cap(nxt) - (carve ? front : 0));   // what of the idle record buffer the plan may cut up
  const int with0 = front ? 0 : 1;                               // a plan with a front has no buffer 0 (the front IS the pass's destination)
  // per buffer from the reserve: W words of records + W / 12 state indices (refs + fingerprints: 16 B each) = 9.34 B per word; a piece of the idle
  // record buffer only needs its index arrays there: 1.34 B per word
  const u64 floor_w = (u64)1 << 21;                              // 16 MB of records (tiny spaces; a device without that much left fails below)
  const double w_old = std::max((double)floor_w, std::min(idle_w / 4.0, reserve / (double)count / 9.34));   // rounds 3-4: buffers 1 .. count
  const double cost_old = (with0 ? 1.0 / idle_w : 0.0) + (double)count / w_old;
  int best_n = 0;
  double best_s = 0;
  if (carve)
    for (int n_idle = 1; n_idle <= count + with0; n_idle++) {
      const double from_res = (double)(count + with0 - n_idle) * 9.34 + (double)(n_idle - with0) * 1.34;
      const double s = std::min(idle_w / (double)n_idle, from_res > 0 ? reserve / from_res : 1e30);
      if (s > best_s) { best_s = s; best_n = n_idle; }
    }
  const bool carved = carve && best_n >= 1 + with0 && best_s >= (double)floor_w && (double)(count + with0) / best_s < cost_old;
  const u64 wcap = carved ? (u64)best_s & ~(u64)63 : (u64)w_old;
  auto alloc_index = [&](PassDst& B) -> bool {
    B.cap = std::max<u64>((u64)1 << 15, B.words_cap / 12);
    return hipMalloc((void**)&B.off, (B.cap + 1) * 8) == hipSuccess && hipMalloc((void**)&B.fp, B.cap * 8) == hipSuccess;
  };
  bool ok = true;
  if (carved && with0) {
    c->scratch0.words = c->words[nxt]; c->scratch0.words_cap = wcap; c->scratch0.own_words = false;
    c->scratch0.off = c->off[nxt]; c->scratch0.fp = c->lvl_fp; c->scratch0.own_index = false;
    c->scratch0.cap = std::min<u64>(c->opt.frontier_states, std::max<u64>((u64)1 << 15, wcap / 12));
  }
  for (int k = 1; k <= count && ok; k++) {
    PassDst B;
    B.words_cap = wcap;
    if (carved && k - 1 + with0 < best_n) { B.words = c->words[nxt] + front + (u64)(k - 1 + with0) * wcap; B.own_words = false; }   // piece k - 1 + with0 behind the front
    else ok = hipMalloc((void**)&B.words, B.words_cap * 8) == hipSuccess;
    ok = ok && alloc_index(B);
    c->scratch.push_back(B);                                     // (pushed even when an allocation failed: deep_free_scratch releases what it holds)
  }
  if (!ok) {
    (void)hipGetLastError();
    deep_free_scratch(c);
    return fail(VSRMC_E_HIP, "hipMalloc of a deep-search scratch buffer failed");
  }
  c->scratch_carved = carve;
  c->scratch_buf = nxt;
  c->scratch_front = front;
  return 0;
}

int deep_buffer(vsrmc_checker* c, int k, PassDst* out) {
  const int nxt = c->cur ^ 1;
  if (k == 0) {
    if (c->scratch0.words) { *out = c->scratch0; return 0; }     // a piece of the idle record buffer (the plan carved it)
    out->words = c->words[nxt]; out->words_cap = c->words_cap(nxt); out->off = c->off[nxt]; out->fp = c->lvl_fp; out->cap = c->opt.frontier_states;
    return 0;
  }
  if ((size_t)k > c->scratch.size()) return fail(VSRMC_E_REP, "deep search: a nesting level without a planned scratch buffer");
  *out = c->scratch[(size_t)k - 1];
  return 0;
}

void deep_acc(vsrmc_level_info* t, const LevelCtl& h, double ms) {
  t->generated += h.generated;
  t->deadlocks += h.deadlocks;
  t->probes += h.probes;
  for (int a = 0; a < 16; a++) t->act_generated[a] += h.act_generated[a];
  t->expand_ms += ms;
}

int deep_run_pass(DeepRun& R, const u64* sw, const u64* so, u64 n, u64 p_off, int level, int mode, u64 bag, const PassDst* dst) {
  R.c->extra_launches = 0;
  if (R.io) return R.io->pass(R.io->ctx, sw, so, n, p_off, level, mode, bag, dst);
  return expand_pass(R.c, sw, so, n, p_off, level, mode, bag, dst);   // (expand_pass finds the claim bitmap of the first seen-set-only level by itself)
}

// One bit per (parent of the base level, ordinal): 4 x ceil(ordinals / 32) bytes per parent (README: 20 B x 2.6e8 parents = 5 GB), from the reserve.
// No memory for it: the regenerating passes keep asking the seen-set (MODE_REGEN's other rule).
void deep_claim_bits_alloc(vsrmc_checker* c, u64 src_max_bag) {
  if (c->claim_bits) { (void)hipFree(c->claim_bits); c->claim_bits = nullptr; }
  c->claim_w = c->claim_parents = 0;
  if (c->opt.world > 1 || !c->regen_bits_kernel || !c->insert_kernel || std::getenv("VSRMC_NO_CLAIM_BITS")) return;   // (two instantiations of k_expand know the bitmap)
  const Model& M = c->model.M;
  const u64 ords = (u64)M.m0 + std::min<u64>(src_max_bag, (u64)M.max_bag) * (u64)(M.R + 1) + 1;
  const u64 w = (ords + 31) / 32, bytes = c->n_frontier * w * 4;
  if (w > 8) return;                                             // (k_expand<.., 3> reads two words per thread, four threads per record at 64-record tiles)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || (double)bytes > 0.4 * (double)free_b) return;
  if (hipMalloc((void**)&c->claim_bits, std::max<u64>(bytes, 4)) != hipSuccess) { (void)hipGetLastError(); c->claim_bits = nullptr; return; }
  if (hipMemsetAsync(c->claim_bits, 0, std::max<u64>(bytes, 4), c->stream) != hipSuccess) { (void)hipFree(c->claim_bits); c->claim_bits = nullptr; return; }
  c->claim_w = w;
  c->claim_parents = c->n_frontier;
}
bool deep_more(DeepRun& R, bool mine, int* rc) {               // does ANY rank have another slice?  (unsharded: this one)
  u64 f = (mine ? 1 : 0) | (R.err ? 2 : 0);
  if (R.io) *rc = R.io->any(R.io->ctx, &f);                     // the maximum over the ranks: >= 2 = some rank failed
  if (!*rc && f >= 2) *rc = R.err ? R.err : fail(VSRMC_E_STATE, "deep pass: another rank failed between two exchanges");
  return !*rc && (f & 1) != 0;
}
int deep_local_fail(DeepRun& R, int rc) {                      // unsharded: the error itself; sharded: remembered, the caller goes on to the next exchange
  if (!R.io) return rc;
  if (!R.err) R.err = rc;
  return 0;
}
u64 deep_cand_bound(const DeepRun& R, u64 g1) {
  if (!R.io || R.io->world <= 1) return ~(u64)0;
  return std::max<u64>(128, (u64)(0.45 * (double)R.io->cand_cap * (double)R.io->world / (double)std::max<u64>(1, g1)) & ~(u64)127);
}

// one probe pass over `n` indices of a buffer holding (part of) level lv: the invariants of their successors, nothing inserted
int deep_probe(DeepRun& R, const PassDst& B, u64 n, int lv, u64 bag) {
  vsrmc_checker* c = R.c;
  if (!n) return 0;
  c->expand_ms = 0;
  if (R.err) return 0;
  int rc = expand_pass(c, B.words, B.off, n, 0, lv + 1, MODE_PROBE, bag);
  R.launches += c->extra_launches;
  if (!rc && c->h.limit_unchecked) {                            // a record at a representation limit beside instances the footprint filter skipped: once more, every action applied
    R.prb.limit_rechecked += c->h.limit_unchecked;
    c->probe_all_actions = true;
    rc = expand_pass(c, B.words, B.off, n, 0, lv + 1, MODE_PROBE, bag);
    c->probe_all_actions = false;
    R.launches += 1 + c->extra_launches;
  }
  if (rc) return deep_local_fail(R, rc);
  R.launches++;
  deep_acc(&R.prb, c->h, c->expand_ms);
  if (c->h.n_pending) {
    R.mask_prb |= c->h.viol_mask;
    if (c->h.n_pending > c->opt.pending_entries || R.bad.size() / 2 + c->h.n_pending > ((u64)1 << 24))
      return deep_local_fail(R, fail(VSRMC_E_REP, "more violating successors in the probed level than the pending list holds (pending_entries)"));
    const size_t at = R.bad.size();
    R.bad.resize(at + 2 * c->h.n_pending);
    if (hipMemcpy(R.bad.data() + at, c->pending, 16 * c->h.n_pending, hipMemcpyDeviceToHost) != hipSuccess)
      return deep_local_fail(R, fail(VSRMC_E_HIP, "deep search: copy of the probe's candidates"));
  }
  return 0;
}

// re-basing: the valid records of a regenerated slice of the target level, appended (packed: no holes, no chunk slack) to the idle record buffer
int deep_export(DeepRun& R, const PassDst& B, u64 part) {
  vsrmc_checker* c = R.c;
  if (!part) return 0;
  const int nxt = c->cur ^ 1;
  hipLaunchKernelGGL(k_export, dim3((unsigned)((part + 63) / 64)), dim3(64), 0, c->stream, (const u64*)B.words, B.off, B.fp, (u64)0, part,
                     c->words[nxt], c->scratch_front ? c->scratch_front : c->words_cap(nxt), c->off[nxt], c->lvl_fp, c->opt.frontier_states, R.d_cnt, R.d_err);   // (never past the front: scratch lies behind it)
  if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) return fail(VSRMC_E_HIP, "deep search: k_export (re-basing)");
  return 0;
}

// `n_idx` indices (holes included) of level lv in (src_words, src_off); k = nesting depth = which buffer takes what this level yields
int deep_descend(DeepRun& R, const u64* src_words, const u64* src_off, u64 n_idx, u64 src_bag, int lv, int k) {
  vsrmc_checker* c = R.c;
  const Model& M = c->model.M;
  PassDst B;
  int rc = deep_buffer(c, k, &B);
  if (rc) return rc;
  Slicer sl;
  sl.cap_n = B.cap; sl.cap_w = B.words_cap; sl.g1 = std::max<u64>(1, c->deep_g); sl.stride = (u64)c->lds_stride;
  sl.cap_c = deep_cand_bound(R, sl.g1);
  const bool regen = lv + 1 <= R.last_regen;
  const DeepLevel& next = c->deep_lv[(size_t)(lv - R.base)];   // level lv + 1
  if (regen) {
    // MODE_REGEN writes a state where its min-key parent sits — keys order by the parent's fingerprint, so the states of the next level
    // spread evenly over the parents: (states of level lv+1 per valid parent, known exactly) x (a record of the source + 2 words)
    const u64 parents = lv == R.base ? c->n_valid : c->deep_lv[(size_t)(lv - R.base - 1)].n_local;   // (a rank regenerates about what it inserted)
    const double per_n = (double)next.n_local / (double)std::max<u64>(1, parents);
    const double wbar = (lv == R.base ? (double)c->cur_w / (double)std::max<u64>(1, c->n_valid) : (double)(M.fixed + (int)src_bag)) + 2.0;
    sl.expect(per_n, per_n * wbar);
  }
  const u64 out_bag = regen ? std::min<u64>(next.max_bag, (u64)M.max_bag) : 0;
  for (u64 a = 0; deep_more(R, a < n_idx, &rc);) {
    if (rc) return rc;
    const u64 n = a < n_idx ? std::min<u64>(sl.next(), n_idx - a) : 0;   // (a rank that has run out still takes part in the others' exchanges)
    c->expand_ms = 0;
    rc = deep_run_pass(R, src_words, src_off + a, n, (R.io && c->opt.world > 1) ? 0 : a, lv + 1, regen ? MODE_REGEN : MODE_NORMAL, src_bag, &B);
    if (rc) return rc;
    if (n) R.launches += 1 + c->extra_launches;
    a += n;
    if (n) (k == 0 ? R.n_slices : R.n_subs)++;
    const u64 part = c->h.n_new;                               // index range written (holes included)
    sl.observe(n, part, c->h.words_new);
    if (regen) {
      R.ins.materialize_ms += c->expand_ms;                    // time spent regenerating (reported beside the level's own expansion)
      if (lv + 1 == R.last_regen && !R.insert) rc = R.rebase ? deep_export(R, B, part) : deep_probe(R, B, part, lv + 1, out_bag);
      else rc = deep_descend(R, B.words, B.off, part, out_bag, lv + 1, k + 1);
      if (rc) return rc;
      continue;
    }
    // ---- the level that is inserted: new states of level lv + 1, in B until the next sub-slice overwrites them
    deep_acc(&R.ins, c->h, c->expand_ms);
    R.ins.record_words += c->h.rec_words;
    R.ins.max_bag = std::max<u64>(R.ins.max_bag, c->h.max_bag);
    if (c->h.viol_fp != ~(u64)0) {
      R.mask_ins |= c->h.viol_mask;
      R.viol_ins = std::min<u64>(R.viol_ins, c->h.viol_fp);
    }
    if (!part) continue;
    const u64 bag_new = std::min<u64>(c->h.max_bag, (u64)M.max_bag);
    {   // checksums of the level: the sub-slice's fingerprints sit in the scratch buffer until it is reused
      u64 hsum[3] = {0, 0, 0};
      bool ok = hipMemsetAsync(R.d_sum, 0, 24, c->stream) == hipSuccess;
      if (ok) hipLaunchKernelGGL(k_level_checksum, dim3(1024), dim3(256), 0, c->stream, B.fp, part, R.d_sum);
      ok = ok && hipGetLastError() == hipSuccess && hipMemcpyAsync(hsum, R.d_sum, 24, hipMemcpyDeviceToHost, c->stream) == hipSuccess &&
           hipStreamSynchronize(c->stream) == hipSuccess;
      if (!ok) {
        rc = deep_local_fail(R, fail(VSRMC_E_HIP, "deep search: k_level_checksum"));
        if (rc) return rc;
        continue;
      }
      R.ins.fp_xor ^= hsum[0];
      R.ins.fp_sum += hsum[1];
      R.ins.n_new += hsum[2];                                  // the states that stayed (a sharded pass has withdrawn the announced ones that lost)
    }
    if (R.viol_ins != ~(u64)0) continue;                       // a violation in the inserted level: it is completed, nothing deeper is probed
    rc = deep_probe(R, B, part, lv + 1, bag_new);
    if (rc) return rc;
  }
  return rc;
}

// which of the collected violating successors are states of the probed level (not of an earlier one)?  The smallest such fingerprint is
// the violation; the smallest key among its copies names its parent, a state of the level below that is in the seen-set.
int deep_resolve(DeepRun& R) {
  vsrmc_checker* c = R.c;
  const u64 nbad = R.bad.size() / 2;
  if (!nbad) return 0;
  const int plevel = R.probed_level();
  std::vector<std::pair<u64, u64>> pairs(nbad);
  for (u64 i = 0; i < nbad; i++) pairs[i] = std::make_pair(R.bad[2 * i], R.bad[2 * i + 1]);
  std::sort(pairs.begin(), pairs.end());                       // by fingerprint, then key
  std::vector<u64> fps;
  for (u64 i = 0; i < nbad; i++)
    if (i == 0 || pairs[i].first != pairs[i - 1].first) fps.push_back(pairs[i].first);
  std::vector<uint8_t> seen8(fps.size(), 0);
  int rc = vsrmc_checker_seen_batch(c, fps.data(), (u64)fps.size(), plevel, seen8.data());
  if (rc) return rc;
  u64 first = ~(u64)0, key = ~(u64)0;
  size_t g = 0;
  for (u64 i = 0; i < nbad; i++) {
    if (i && pairs[i].first != pairs[i - 1].first) g++;
    if (seen8[g]) continue;
    R.prb.pending++;                                           // violating successors seen, duplicates included
    if (first == ~(u64)0) { first = pairs[i].first; key = pairs[i].second; }
    if (c->probe_viol.empty() || c->probe_viol.back() != pairs[i].first) {   // (sorted: ascending, distinct; the first copy of a fingerprint carries its smallest key)
      c->probe_viol.push_back(pairs[i].first);
      c->probe_viol_key.push_back(pairs[i].second);
    }
  }
  c->probe_viol_level = plevel;
  if (first == ~(u64)0) return 0;
  int found = 0;
  u64 pfp = 0, pmeta = 0;
  rc = table_lookup(c, meta_pfp(key), plevel - 1, 1, &found, &pfp, &pmeta);
  if (rc) return rc;
  if (found > 1) return fail(VSRMC_E_STATE, "ambiguous predecessor pointer: several states of the parent's level share the 45 fingerprint bits the violating successor keeps of its parent");
  R.prb.viol_fp = first;
  R.prb.viol_mask = (int32_t)R.mask_prb;
  if (found) { c->probe_fp = pfp; c->probe_level = plevel - 1; c->probe_extra_fp = first; }
  return 0;
}

int deep_check_ready(vsrmc_checker* c, int extra_levels, bool sharded) {
  if (c->opt.exact_ties) return fail(VSRMC_E_STATE, "levels beyond the record buffers need a single-pass checker");
  if (!sharded && c->opt.world > 1) return fail(VSRMC_E_STATE, "a sharded checker goes beyond its record buffers with vsrmc_shard_loop_deepen");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  if (c->level + c->deep + extra_levels >= 511) return fail(VSRMC_E_REP, "more than 510 BFS levels");
  return 0;
}

// pass 1: the newest materialised level expanded in MODE_INSERT — level L+1 exists as seen-set entries from here on.  One launch over
// the whole level; a sharded run slices it so that a slice's announcements fit the exchange buffers.
int deep_first_pass(vsrmc_checker* c, vsrmc_level_info* ins, const DeepIo* io = nullptr) {
  const Model& M = c->model.M;
  DeepRun R;
  R.c = c; R.io = io; R.base = c->level; R.last_regen = c->level; R.insert = true;
  std::memset(&R.ins, 0, sizeof(R.ins));
  std::memset(&R.prb, 0, sizeof(R.prb));
  R.ins.viol_fp = R.ins.viol_index = R.prb.viol_fp = R.prb.viol_index = ~(u64)0;
  const double t0 = now_s();
  const u64 bagL = c->bag_known ? c->cur_max_bag : (u64)M.max_bag;
  // whatever happens next, the seen-set holds a level that has no frontier from here on: vsrmc_checker_step is refused
  c->deep = 1;
  c->deep_lv.assign(1, DeepLevel());
  const u64 cb = deep_cand_bound(R, std::max<u64>(c->deep_g, c->g_last));
  if (!io || c->opt.world == 1) deep_claim_bits_alloc(c, bagL);   // (one rank through the level loop: nothing is remote, the unsharded mechanism applies)
  int rc = 0;
  for (u64 a = 0; deep_more(R, a < c->n_frontier, &rc);) {
    if (rc) break;
    const u64 n = a < c->n_frontier ? std::min<u64>(cb, c->n_frontier - a) : 0;
    c->expand_ms = 0;
    rc = deep_run_pass(R, c->words[c->cur], c->off[c->cur] + a, n, (io && c->opt.world > 1) ? 0 : a, c->level + 1, MODE_INSERT, bagL, nullptr);
    if (rc) break;
    a += n;
    if (n) R.launches += 1 + c->extra_launches;
    deep_acc(&R.ins, c->h, c->expand_ms);
    R.ins.n_new += c->h.n_new;
    R.ins.max_bag = std::max<u64>(R.ins.max_bag, c->h.max_bag);
    R.ins.fp_xor ^= c->h.fp_xor;
    R.ins.fp_sum += c->h.fp_sum;
    if (c->h.viol_fp != ~(u64)0) { R.mask_ins |= c->h.viol_mask; R.viol_ins = std::min<u64>(R.viol_ins, c->h.viol_fp); }
  }
  if (rc) { c->failed = 1; return rc; }
  const u64 n_local = R.ins.n_new, frontier_local = c->n_valid;
  R.ins.frontier = c->n_valid;
  R.ins.viol_fp = R.viol_ins;
  R.ins.viol_mask = (int32_t)R.mask_ins;
  if (io && (rc = io->reduce(io->ctx, &R)) != 0) { c->failed = 1; return rc; }
  DeepLevel& d = c->deep_lv[0];
  d.n_new = R.ins.n_new; d.n_local = n_local; d.generated = R.ins.generated; d.max_bag = R.ins.max_bag; d.frontier = R.ins.frontier;
  c->deep_g = std::max<u64>(c->deep_g, (R.ins.generated + R.ins.frontier - 1) / std::max<u64>(1, R.ins.frontier) + 1);
  (void)frontier_local;
  *ins = R.ins;
  ins->level = c->level + (R.ins.n_new ? 1 : 0);               // like vsrmc_checker_step: an empty level leaves the depth where it was
  if (R.ins.n_new == 0) { c->deep = 0; c->deep_lv.clear(); }   // exhausted: nothing was inserted, the checker is where it was
  ins->pending = R.launches;                                   // k_expand launches of this pass
  c->deep_distinct = c->distinct + (io ? n_local : R.ins.n_new);   // sharded: this rank's share of both — what its own seen-set holds (the level loop keeps the run's total)
  c->deep_generated = c->total_generated + R.ins.generated;
  ins->distinct = c->deep_distinct;
  ins->total_generated = c->deep_generated;
  ins->seconds = now_s() - t0;
  ins->viol_index = ~(u64)0;
  if (ins->viol_fp != ~(u64)0) {                               // the violator is in the seen-set: walk from the violator itself
    c->probe_fp = ins->viol_fp;
    c->probe_level = c->level + 1;
    c->probe_extra_fp = 0;
  } else {
    ins->viol_mask = 0;
  }
  return 0;
}

// a descent from the base: regenerates levels base+1 .. last_regen, then inserts (insert) or only probes the level after them
int deep_pass(vsrmc_checker* c, int last_regen, bool insert, vsrmc_level_info* ins, vsrmc_level_info* prb, const DeepIo* io = nullptr) {
  const Model& M = c->model.M;
  DeepRun R;
  R.c = c; R.io = io; R.base = c->level; R.last_regen = last_regen; R.insert = insert;
  std::memset(&R.ins, 0, sizeof(R.ins));
  std::memset(&R.prb, 0, sizeof(R.prb));
  R.ins.viol_fp = R.ins.viol_index = R.prb.viol_fp = R.prb.viol_index = ~(u64)0;
  const double t0 = now_s();
  if (io && c->opt.world > 1) {
    c->wepoch++;                                               // sharded: a regenerating lane takes a state by raising its epoch in the rank's winner set — nothing to clear
  } else if (c->deep_regen_done) {                             // the taken bits an earlier descent left in the levels beyond the base
    hipLaunchKernelGGL(k_table_untake, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1, c->level + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->deep_regen_done = true;
  int rc = 0;
  if (hipMalloc((void**)&R.d_sum, 24) != hipSuccess) { (void)hipGetLastError(); R.d_sum = nullptr; rc = fail(VSRMC_E_HIP, "deep pass: hipMalloc of the checksum words"); }
  struct FreeSum { u64* p; ~FreeSum() { if (p) (void)hipFree(p); } } free_sum{R.d_sum};
  if (insert) { c->deep_lv.resize((size_t)(last_regen + 1 - c->level)); c->deep_lv.back() = DeepLevel(); }
  if (!rc) rc = deep_plan_scratch(c, last_regen - c->level - (insert ? 0 : 1));   // buffer k takes what level base + k yields
  if (io) {                                                    // a rank that could not allocate takes the others with it, before any of them waits for it
    u64 bad_plan = rc ? 1 : 0;                                 // (round-4 advice: NO local failure of this prologue returns before this exchange)
    const int arc = io->any(io->ctx, &bad_plan);
    if (!rc && (arc || bad_plan)) rc = arc ? arc : fail(VSRMC_E_HIP, "deep pass: another rank could not allocate its scratch buffers");
  }
  if (rc) { c->failed = 1; return rc; }
  if (io) { u64 sync = 0; rc = io->any(io->ctx, &sync); }      // every rank has cleared its taken bits
  if (!rc) rc = deep_descend(R, c->words[c->cur], c->off[c->cur], c->n_frontier, c->bag_known ? c->cur_max_bag : (u64)M.max_bag, c->level, 0);
  if (rc) { c->failed = 1; return rc; }
  const u64 n_local = R.ins.n_new;
  R.ins.viol_fp = R.viol_ins;
  R.ins.viol_mask = (int32_t)R.mask_ins;
  R.prb.viol_mask = (int32_t)R.mask_prb;
  if (io && (rc = io->reduce(io->ctx, &R)) != 0) { c->failed = 1; return rc; }
  const double dt = now_s() - t0;
  const DeepLevel& below = c->deep_lv[(size_t)(last_regen - c->level - 1)];
  if (insert) {
    DeepLevel& d = c->deep_lv.back();
    d.n_new = R.ins.n_new; d.n_local = n_local; d.generated = R.ins.generated; d.max_bag = R.ins.max_bag; d.frontier = below.n_new;
    c->deep_g = std::max<u64>(c->deep_g, (R.ins.generated + below.n_new - 1) / std::max<u64>(1, below.n_new) + 1);
    c->deep_distinct += io ? n_local : R.ins.n_new;            // (sharded: the rank's share, see deep_first_pass)
    c->deep_generated += R.ins.generated;
    *ins = R.ins;
    ins->level = last_regen + (R.ins.n_new ? 1 : 0);            // like vsrmc_checker_step: an empty level leaves the depth where it was
    ins->frontier = below.n_new;
    ins->distinct = c->deep_distinct;
    ins->total_generated = c->deep_generated;
    ins->pending = R.launches;                                 // k_expand launches of the whole pass (regenerating, inserting and probing ones)
    ins->words_new = (R.n_slices << 32) | (R.n_subs & 0xFFFFFFFFull);   // (slices of the base level) << 32 | slices of the levels below it
    ins->viol_index = ~(u64)0;
    const double ms_all = R.ins.expand_ms + R.ins.materialize_ms + R.prb.expand_ms;
    ins->seconds = dt * ((R.ins.expand_ms + R.ins.materialize_ms) / std::max(1e-9, ms_all));   // the levels share the pass: split by kernel time
    if (R.ins.n_new) c->deep = last_regen + 1 - c->level;        // an empty level is no level: the search is exhausted at the one below
    else c->deep_lv.pop_back();
    if (ins->viol_fp != ~(u64)0) {
      c->probe_fp = ins->viol_fp;
      c->probe_level = last_regen + 1;
      c->probe_extra_fp = 0;
      return 0;
    }
    ins->viol_mask = 0;
  }
  R.prb.level = R.probed_level();
  R.prb.frontier = insert ? R.ins.n_new : below.n_new;
  R.prb.distinct = c->deep_distinct;
  R.prb.total_generated = c->deep_generated + R.prb.generated;
  R.prb.seconds = insert ? dt - ins->seconds : dt;
  R.prb.viol_fp = R.prb.viol_index = ~(u64)0;
  R.prb.viol_mask = 0;
  if (!insert) {                                               // probe2: the regeneration is this level's cost
    R.prb.record_words = R.ins.record_words;
    R.prb.materialize_ms = R.ins.materialize_ms;
    R.prb.words_new = (R.n_slices << 32) | (R.n_subs & 0xFFFFFFFFull);
  }
  rc = io ? io->resolve(io->ctx, &R) : deep_resolve(R);        // the level below the probed one is complete now: which collected successors are new states?
  if (rc) return rc;
  *prb = R.prb;
  return 0;
}

// ---- re-basing (round 5) ----------------------------------------------------------------------------------------------------------------
// Every pass of the deep search regenerates ALL the levels between the base and the one it inserts.  When the levels shrink again — the analysis
// models after their peak: level 38 of VR_STATE_TRANSFER has 1.3e8 states against 7.2e8 at level 32 — the newest seen-set-only level fits the idle
// record buffer, and one descent that regenerates it and KEEPS what it regenerates (k_export packs every slice into the idle buffer: no holes,
// no chunk slack, so the slices of a hundred launches append without loss) makes it the new stored base: the rest of the run is ordinary stored
// levels.  The 47-level runs of round 4 re-expanded levels 26 .. L in every pass; 196 of model 2's 333 s went into its last seven small levels.
bool rebase_pays(const vsrmc_checker* c) {
  if (!c->deep || c->failed || c->opt.world > 1 || c->opt.exact_ties || c->rebase_off) return false;
  const DeepLevel& top = c->deep_lv.back();
  if (top.n_new == 0) return false;
  if (!(c->deep >= 2 || top.n_new <= c->n_valid)) return false;           // (a level that did not fit a moment ago: wait until the run has shown where it goes)
  const Model& M = c->model.M;
  const int nxt = c->cur ^ 1;
  const double words = (double)top.n_new * (double)(M.fixed + (int)std::min<u64>(top.max_bag, (u64)M.max_bag));   // upper bound: every record with the level's largest bag
  return words <= 0.8 * (double)c->words_cap(nxt) && (double)top.n_new <= 0.8 * (double)c->opt.frontier_states;
}

int deep_rebase(vsrmc_checker* c, vsrmc_level_info* out) {
  const Model& M = c->model.M;
  const int K = c->level + c->deep;
  DeepRun R;
  R.c = c; R.base = c->level; R.last_regen = K; R.insert = false; R.rebase = true;
  std::memset(&R.ins, 0, sizeof(R.ins));
  std::memset(&R.prb, 0, sizeof(R.prb));
  R.ins.viol_fp = R.ins.viol_index = R.prb.viol_fp = R.prb.viol_index = ~(u64)0;
  const double t0 = now_s();
  // buffer 0 of an ordinary descent IS the idle record buffer: here it is the destination, so the levels base+1 .. K go through scratch buffers 1 .. deep
  // the new base level goes to the FRONT of the idle record buffer, packed: at most n x (fixed + largest bag) words (rebase_pays holds that under 80 % of
  // the buffer); the descent's buffers 1 .. deep are cut from what lies behind it and from the reserve
  const u64 front_w = c->deep_lv.back().n_new * (u64)(M.fixed + (int)std::min<u64>(c->deep_lv.back().max_bag, (u64)M.max_bag));
  int rc = deep_plan_scratch(c, c->deep, true, front_w);
  if (rc) { c->rebase_off = true; return 1; }                    // no memory for one more scratch buffer: the search goes on without re-basing
  // (nothing has changed yet: a failure of these small allocations is not fatal either — the search goes on with ordinary descents, as above)
  struct FreeCnt { u64* p; u32* q; ~FreeCnt() { if (p) (void)hipFree(p); if (q) (void)hipFree(q); } } free_cnt{nullptr, nullptr};
  if (hipMalloc((void**)&R.d_cnt, 16) != hipSuccess) { (void)hipGetLastError(); R.d_cnt = nullptr; c->rebase_off = true; return 1; }
  free_cnt.p = R.d_cnt;
  if (hipMalloc((void**)&R.d_err, 4) != hipSuccess) { (void)hipGetLastError(); R.d_err = nullptr; c->rebase_off = true; return 1; }
  free_cnt.q = R.d_err;
  if (hipMemsetAsync(R.d_cnt, 0, 16, c->stream) != hipSuccess || hipMemsetAsync(R.d_err, 0, 4, c->stream) != hipSuccess) { (void)hipGetLastError(); c->rebase_off = true; return 1; }
  if (c->deep_regen_done) {
    hipLaunchKernelGGL(k_table_untake, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1, c->level + 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->deep_regen_done = true;
  rc = deep_descend(R, c->words[c->cur], c->off[c->cur], c->n_frontier, c->bag_known ? c->cur_max_bag : (u64)M.max_bag, c->level, 1);
  if (rc) { c->failed = 1; return rc; }
  u64 cnt[2] = {0, 0};
  u32 xerr = 0;
  HIPCHK(hipMemcpy(cnt, R.d_cnt, 16, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(&xerr, R.d_err, 4, hipMemcpyDeviceToHost));
  const DeepLevel top = c->deep_lv.back();
  if (xerr || cnt[0] != top.n_new) {                            // the level's size is known exactly: a descent that regenerates another number of states is a bug
    c->failed = 1;
    return fail(xerr ? VSRMC_E_REP : VSRMC_E_STATE, xerr ? std::string("re-basing: the idle record buffer is too small for the level after all")
                : "re-basing: level " + std::to_string(K) + " has " + std::to_string((unsigned long long)top.n_new) + " states, the descent regenerated " +
                  std::to_string((unsigned long long)cnt[0]));
  }
  const u64 below = c->deep >= 2 ? c->deep_lv[c->deep_lv.size() - 2].n_new : c->n_valid;
  c->cur ^= 1;
  c->level = K;
  c->n_frontier = cnt[0];
  c->n_valid = cnt[0];
  c->cur_w = cnt[1];
  c->cur_rec_w = cnt[1];
  c->distinct = c->deep_distinct;
  c->total_generated = c->deep_generated;
  c->cur_max_bag = std::min<u64>(top.max_bag, (u64)M.max_bag);
  c->bag_known = true;
  c->hist_new[0] = below;
  c->hist_new[1] = top.n_new;
  c->g_last = (top.generated + std::max<u64>(1, top.frontier) - 1) / std::max<u64>(1, top.frontier) + 1;
  c->deep = 0;
  c->deep_lv.clear();
  if (c->claim_bits) { (void)hipFree(c->claim_bits); c->claim_bits = nullptr; c->claim_w = c->claim_parents = 0; }   // (it described the old base's successors)
  c->deep_regen_done = false;                                   // lvl_fp holds the new base level's fingerprints (k_export carried them along)
  std::memset(out, 0, sizeof(*out));
  out->level = K;
  out->frontier = below;
  out->n_new = top.n_new;
  out->distinct = c->distinct;
  out->total_generated = c->total_generated;
  out->max_bag = top.max_bag;
  out->words_new = cnt[1];
  out->record_words = cnt[1];
  out->pending = R.launches;
  out->materialize_ms = R.ins.materialize_ms;
  out->seconds = now_s() - t0;
  out->viol_fp = out->viol_index = ~(u64)0;
  return 0;
}

}  // namespace

extern "C" {

// One more level beyond the record buffers (see the head of this file).  The first call after the last vsrmc_checker_step inserts level
// L+1 as a virtual level (inserted->level = L+1, probed->level = 0: nothing probed); every further call descends from the base level,
// inserts the next level and probes the one after it.  A violation is reported in whichever of the two infos it falls
// (vsrmc_checker_probe_trace reconstructs the counter-example); inserted->n_new == 0: the search is exhausted.
int32_t vsrmc_checker_deepen(vsrmc_checker* c, vsrmc_level_info* inserted, vsrmc_level_info* probed) {
  if (!c || !inserted || !probed) return fail(VSRMC_E_ARG, "NULL argument");
  int rc = deep_check_ready(c, 2, false);
  if (rc) return rc;
  HIPCHK(hipSetDevice(c->opt.device));
  std::memset(inserted, 0, sizeof(*inserted));
  std::memset(probed, 0, sizeof(*probed));
  inserted->viol_fp = inserted->viol_index = probed->viol_fp = probed->viol_index = ~(u64)0;
  c->probe_fp = 0;
  c->probe_level = 0;
  c->probe_extra_fp = 0;
  c->probe_viol.clear();
  c->probe_viol_key.clear();
  c->probe_viol_level = 0;
  if (c->deep == 0) return deep_first_pass(c, inserted);
  return deep_pass(c, c->level + c->deep, true, inserted, probed);
}

// Two levels beyond the last materialised one: level L+1 a virtual level, level L+2 probed from its regenerated slices.  Costs one extra
// expansion of the newest level and no memory beyond a slice.
int32_t vsrmc_checker_probe2(vsrmc_checker* c, vsrmc_level_info* virt, vsrmc_level_info* probe) {
  if (!c || !virt || !probe) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->deep) return fail(VSRMC_E_STATE, "vsrmc_checker_probe2 starts from a materialised level");
  vsrmc_level_info none;
  int rc = vsrmc_checker_deepen(c, virt, probe);
  if (rc || virt->viol_mask || virt->n_new == 0) return rc;
  rc = deep_pass(c, c->level + 1, false, &none, probe);
  return rc;
}

// Three levels: level L+1 virtual, level L+2 streamed through a scratch buffer (inserted, counted, never kept), level L+3 probed.
int32_t vsrmc_checker_probe3(vsrmc_checker* c, vsrmc_level_info* virt1, vsrmc_level_info* virt2, vsrmc_level_info* probe) {
  if (!c || !virt1 || !virt2 || !probe) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->deep) return fail(VSRMC_E_STATE, "vsrmc_checker_probe3 starts from a materialised level");
  std::memset(virt2, 0, sizeof(*virt2));
  virt2->viol_fp = virt2->viol_index = ~(u64)0;
  int rc = vsrmc_checker_deepen(c, virt1, probe);
  if (rc || virt1->viol_mask || virt1->n_new == 0) return rc;
  return vsrmc_checker_deepen(c, virt2, probe);
}

}  // extern "C"

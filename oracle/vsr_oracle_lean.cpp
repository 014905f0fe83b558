// oracle/vsr_oracle_lean.cpp — CPU ORACLE, memory-lean multi-threaded driver (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// The same level-synchronous BFS as vsr_oracle_mt.cpp over the same restatement (successors() / fingerprint() /
// check_invariants() of vsr_oracle.cpp = VSR.tla), for runs whose last levels do not fit a host's memory as records:
// the README defect configuration of the reference (/root/reference/README.md:13-18: 3 replicas, {v1,v2,v3}, limit 3) has
// 1.82e9 distinct states within depth 23, 0.5 KB each as an unpacked record.  What is kept:
//   * the seen-set: 8 bytes of fingerprint + 2 bytes (level, canonical auxkey) per slot, one table of the final size;
//   * the records of ONE base level B (--base-level), wire format;
//   * one bit per slot, cleared before every pass.
// Levels 1..B are explored the ordinary way.  Every deeper level l is one PASS over the base level: a base state is expanded,
// each successor that is a state of level B+1 (seen-set says so) and whose bit this worker is the first to set is expanded in
// turn, and so on down to the states of level l-1, whose successors are inserted as level l (or, for the last pass
// --probe-level, only looked up: a successor that is in no earlier level gets its invariants checked, nothing is stored —
// TLC reports a violation while expanding the level before).  A state of levels B+1..l-1 is therefore expanded
// exactly once per pass, by whichever parent reaches it first; since the level is complete in the seen-set before the pass
// starts, the per-level figures — new states, successors generated in total and per action, deadlocks, largest bag, xor and sum
// of the new fingerprints, smallest violating fingerprint — are those of the ordinary BFS and independent of the thread count.
// Cost: level d is expanded once for every deeper level, about 2.2 times the work of the ordinary run for growth x1.75.
// Same-level VIEW ties (never observed) are min-merged on the canonical auxkey and only the copy with the stored
// auxkey is ever expanded, as the ordinary drivers do; a level with ties is flagged (the verdict of a tied state is not re-evaluated here).
//
// Output: the JSON lines of vsr_oracle_mt (one per level, then a summary); the probe pass prints
//   {"probe_level": l, "generated": .., "deadlocks": .., "violating_successors": .., "viol_fp": "..", "viol_mask": ..}.
// CLI: vsr_oracle_lean R C nValues L --base-level B --slots N [--probe-level P] [--max-depth D] [--threads T] [--inv-mask M]
//                      [--no-symmetry] [--assume-commit-number] [--verify-fp-all | --verify-fp-every N]
//
// Collision hunt (oracles that declare fingerprint_with_seed: the second model's): --hunt-seed HEX [--hunt-slots N] keeps a second set, of the
// fingerprints of every NEW state under another member of the function family.  A new state whose audit fingerprint is in that set already is a
// 64-bit collision of the audit function between two states the run's own function tells apart: printed with its record
//   {"fp_collision": true, "audit_seed": "..", "audit_fp": "..", "level": l, "fp": "..", "words": ["..", ..]}
// and from then on every state the later passes meet again (levels >= the base level) is hashed under the audit seed too; those with that audit
// fingerprint are printed as {"fp_collision_member": true, ..}: the other state of the pair, if it lives at or beyond the base level.
// --dump-audit-fp HEX prints the members of a known audit fingerprint from the start (new states of every level).  The audit fingerprint's lowest
// bit is forced to 1 (0 marks an empty slot of the set: 63 bits compared); --hunt-mask HEX shortens it further (tests: forced collisions).
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifndef ORACLE_HPP
#define ORACLE_HPP "vsr_oracle.hpp"
#define ORACLE_NS vsr_oracle
#endif
#include ORACLE_HPP

using namespace ORACLE_NS;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// --assume-commit-number: only VSR.tla's restatement has the field (the driver is compiled once per restatement)
template <typename T>
auto set_assume(T& p, int) -> decltype(p.assume_commit_number = true, void()) { p.assume_commit_number = true; }
template <typename T>
void set_assume(T&, long) { std::fprintf(stderr, "--assume-commit-number: not an option of this model\n"); std::exit(2); }
void set_assume_commit_number(Params& P) { set_assume(P, 0); }

void* map_zero(u64 bytes) {
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
  if (p == MAP_FAILED) { std::fprintf(stderr, "error: mmap of %llu bytes failed\n", (unsigned long long)bytes); std::exit(1); }
  madvise(p, bytes, MADV_HUGEPAGE);
  return p;
}

// seen-set: fingerprint word (0 = empty) + side word (0 = not yet published; else level << 9 | canonical auxkey) per slot
struct Table {
  std::atomic<u64>* fp = nullptr;
  std::atomic<uint16_t>* side = nullptr;
  std::atomic<u64>* bits = nullptr;   // "expanded in this pass", one bit per slot
  u64 slots = 0;
  void alloc(u64 n) {
    slots = n;
    fp = (std::atomic<u64>*)map_zero(n * 8);
    side = (std::atomic<uint16_t>*)map_zero(n * 2);
    bits = (std::atomic<u64>*)map_zero((n + 63) / 64 * 8);
  }
  // the canonical fingerprint is a MINIMUM over the value permutations: its high bits are not uniform (density up to |Values|! times
  // the average near 0), and a home slot taken from them clusters the table to death; scramble first, then reduce to [0, slots)
  u64 home(u64 f) const { return (u64)(((unsigned __int128)(f * 0x9E3779B97F4A7C15ULL) * slots) >> 64); }
};
inline uint16_t side_make(u32 level, u32 auxkey) { return (uint16_t)((level << 9) | (auxkey & 511)); }
inline u32 side_level(uint16_t s) { return s >> 9; }
inline u32 side_auxkey(uint16_t s) { return s & 511; }

struct Piece {
  std::vector<u64> words;
  std::vector<u64> off;
  size_t n() const { return off.empty() ? 0 : off.size() - 1; }
  void reset() { words.clear(); off.assign(1, 0); }
};

struct Stats {
  u64 generated = 0, deadlocks = 0, n_new = 0, fp_xor = 0, fp_sum = 0, viol_fp = ~0ull, ties = 0, viol_seen = 0, expanded = 0;
  u64 act[16] = {0};
  int viol_mask = 0;
  size_t max_bag = 0;
  void add(const Stats& o) {
    generated += o.generated; deadlocks += o.deadlocks; n_new += o.n_new; fp_xor ^= o.fp_xor; fp_sum += o.fp_sum; ties += o.ties;
    viol_seen += o.viol_seen; expanded += o.expanded;
    for (int a = 0; a < 16; a++) act[a] += o.act[a];
    if (o.viol_mask && o.viol_fp < viol_fp) { viol_fp = o.viol_fp; viol_mask = o.viol_mask; }
    max_bag = std::max(max_bag, o.max_bag);
  }
};

// ---- fingerprint on the encoded record --------------------------------------------------------------------------------
// fingerprint() of vsr_oracle.cpp builds, for every permutation of Values, the permuted State (copies, re-sorted sets), encodes
// it and sums one fmix64 term per word: six heap-allocating copies per successor under config 3, 80 % of this driver's time.
// The sum does not depend on the order of a bag or set, and a permutation only rewrites the value field of the in-use log-entry
// bytes of the codec's words (enc_entry / enc_log / enc_msg / enc_replica of vsr_oracle.cpp: own log in bytes 0-2 of replica
// word 1, DVC-slot logs in bytes 5-7 of word 1 and bytes 1-3 / 5-7 of words >= 2, entry / log of a message in bytes 4-6), so the
// same minimum can be taken over the ONE encoded record.  fingerprint() stays the definition: every `verify_every`-th call (and
// every call under --verify-fp-all, which the tests use) compares the two and aborts the run on the first difference.
inline u64 perm_bytes(u64 w, u64 bytemask, const int* pi) {
  for (int b = 0; b < 8; b++) {
    if (!((bytemask >> (8 * b)) & 1)) continue;
    const u64 e = (w >> (8 * b)) & 0xFF;
    if (!(e & 7)) continue;                                     // no entry in this byte (view number 0)
    const u64 v = (e >> 3) & 3;
    w = (w & ~((u64)3 << (8 * b + 3))) | ((u64)pi[v] << (8 * b + 3));
  }
  return w;
}
#ifdef ORACLE_VRST
// the analysis models have no SYMMETRY: one encoding, one sum — fingerprint() of the restatement is the fast path already
struct FastFp {
  u64 calls = 0, verify_every = 0;
  Fp operator()(const Params& P, const State& st, const u64*, size_t) { return fingerprint(P, st); }
};
#else
struct FastFp {
  u64 salt[8][4];
  u64 calls = 0, verify_every = 4096;
  FastFp() {
    for (int r = 0; r < 8; r++)
      for (int k = 0; k < 4; k++) salt[r][k] = fmix64(0xA0761D6478BD642FULL + (u64)(8 * r + k)) ^ fp_seed();   // (VSR_ORACLE_FP_SEED: second-hash audit)
  }
  Fp operator()(const Params& P, const State& st, const u64* rec, size_t nwords) {
    const int wpr = words_per_replica(P), fixed = fixed_words(P);
    // words without an in-use entry byte contribute the same term under every permutation: summed once
    u64 inv = 0;
    const u64* vw[64]; u64 vm[64], vs[64];
    int nv = 0;
    auto add = [&](const u64* w, u64 mask, u64 salt_) {
      const u64 x = *w;
      if (mask && (((x | (x >> 1) | (x >> 2)) & mask) != 0) && nv < 64) { vw[nv] = w; vm[nv] = mask; vs[nv] = salt_; nv++; }
      else if (mask && (((x | (x >> 1) | (x >> 2)) & mask) != 0)) throw RepError("more than 64 value-carrying words");
      else inv += fmix64(x ^ salt_);
    };
    for (int r = 1; r <= P.R; r++) {
      const u64* b = rec + 1 + (size_t)(r - 1) * wpr;
      add(b, 0, salt[r][0]);
      add(b + 1, 0x0101010000010101ULL, salt[r][1]);
      for (int k = 2; k < wpr; k++) add(b + k, 0x0101010001010100ULL, salt[r][k]);
    }
    for (size_t j = fixed; j < nwords; j++) add(rec + j, 0x0001010100000000ULL, 0x9E3779B97F4A7C15ULL ^ fp_seed());
    int pi[4] = {0, 1, 2, 3};
    Fp best;
    best.fp = 0; best.auxkey = 0; best.argmin = -1;
    int idx = 0;
    do {
      u64 sum = inv;
      for (int q = 0; q < nv; q++) sum += fmix64(perm_bytes(*vw[q], vm[q], pi) ^ vs[q]);
      u32 ak = (u32)st.aux_svc;
      for (int v = 0; v < P.n; v++) ak |= (u32)st.acked[v] << (3 + 2 * pi[v]);
      if (best.argmin < 0 || sum < best.fp || (sum == best.fp && ak < best.auxkey)) { best.fp = sum; best.auxkey = ak; best.argmin = idx; }
      idx++;
    } while (P.symmetry && std::next_permutation(pi, pi + P.n));
    if (best.fp == 0) best.fp = 1;
    if (verify_every && (calls++ % verify_every) == 0) {
      const Fp f = fingerprint(P, st);
      if (f.fp != best.fp || f.auxkey != best.auxkey) throw RepError("fingerprint on the encoded record differs from fingerprint()");
    }
    return best;
  }
};
#endif

// second-hash collision hunt (see the header)
struct Hunt {
  bool on = false;
  u64 seed = 0, slots = 0, mask = ~(u64)0;   // mask: tests shorten the audit fingerprint to force collisions
  std::atomic<u64>* keys = nullptr;
  std::atomic<u64> dump_fp{0};   // audit fingerprint whose states are printed when met (0 = none yet)
  std::atomic<u64> found{0};
  std::mutex mu;
  void alloc(u64 n) { slots = n; keys = (std::atomic<u64>*)map_zero(n * 8); }
  u64 home(u64 f) const { return (u64)(((unsigned __int128)(f * 0x9E3779B97F4A7C15ULL) * slots) >> 64); }
  bool put(u64 f) {              // true when f was in the set already
    u64 i = home(f);
    for (u64 probes = 0;; probes++) {
      u64 cur = keys[i].load(std::memory_order_acquire);
      if (cur == 0) {
        u64 exp = 0;
        if (keys[i].compare_exchange_strong(exp, f, std::memory_order_acq_rel)) return false;
        cur = exp;
      }
      if (cur == f) return true;
      if (++i == slots) i = 0;
      if (probes > slots) throw RepError("audit set full");
    }
  }
};

struct Ctx {
  Params P;
  Table tab;
  u32 target = 0;        // the level whose states this pass inserts (or probes)
  bool probe = false;
  Hunt hunt;
};

#ifdef ORACLE_HAS_SEEDED_FP
void hunt_print(Ctx& c, const char* what, u64 audit_fp, u32 level, u64 fp, const u64* rec, size_t n) {
  std::lock_guard<std::mutex> g(c.hunt.mu);
  std::printf("{\"%s\": true, \"audit_seed\": \"%016llx\", \"audit_fp\": \"%016llx\", \"level\": %u, \"fp\": \"%016llx\", \"words\": [", what,
              (unsigned long long)c.hunt.seed, (unsigned long long)audit_fp, level, (unsigned long long)fp);
  for (size_t k = 0; k < n; k++) std::printf("\"%016llx\"%s", (unsigned long long)rec[k], k + 1 < n ? ", " : "");
  std::printf("]}\n");
  std::fflush(stdout);
}
// a state this run has just inserted as new
inline void hunt_new(Ctx& c, const State& st, u64 fp, const u64* rec, size_t n) {
  if (!c.hunt.on) return;
  const u64 g = (fingerprint_with_seed(c.P, st, c.hunt.seed).fp & c.hunt.mask) | 1;   // (0 = empty slot)
  const bool dup = c.hunt.keys ? c.hunt.put(g) : false;
  if (dup) {
    c.hunt.found.fetch_add(1);
    u64 none = 0;
    c.hunt.dump_fp.compare_exchange_strong(none, g);             // the first collision's members are printed from now on
    hunt_print(c, "fp_collision", g, c.target, fp, rec, n);
  } else if (g == c.hunt.dump_fp.load(std::memory_order_relaxed)) {
    hunt_print(c, "fp_collision_member", g, c.target, fp, rec, n);
  }
}
// a state of an earlier level that a pass meets again
inline void hunt_revisit(Ctx& c, const State& st, u32 level, u64 fp, const u64* rec, size_t n) {
  const u64 want = c.hunt.dump_fp.load(std::memory_order_relaxed);
  if (!c.hunt.on || !want) return;
  if (((fingerprint_with_seed(c.P, st, c.hunt.seed).fp & c.hunt.mask) | 1) == want) hunt_print(c, "fp_collision_member", want, level, fp, rec, n);
}
#else
inline void hunt_new(Ctx&, const State&, u64, const u64*, size_t) {}
inline void hunt_revisit(Ctx&, const State&, u32, u64, const u64*, size_t) {}
#endif

// insert-or-find of fingerprint f as a state of level ctx.target; returns true when this call inserted it
inline bool insert_level(Ctx& c, const Fp& f, Stats& st) {
  Table& tab = c.tab;
  if (side_auxkey((uint16_t)f.auxkey) != f.auxkey) throw RepError("canonical auxkey beyond 9 bits");
  u64 i = tab.home(f.fp);
  for (u64 probes = 0;; probes++) {
    u64 cur = tab.fp[i].load(std::memory_order_acquire);
    if (cur == 0) {
      u64 exp = 0;
      if (tab.fp[i].compare_exchange_strong(exp, f.fp, std::memory_order_acq_rel)) {
        tab.side[i].store(side_make(c.target, f.auxkey), std::memory_order_release);
        return true;
      }
      cur = exp;
    }
    if (cur == f.fp) {
      uint16_t s;
      while ((s = tab.side[i].load(std::memory_order_acquire)) == 0) std::this_thread::yield();   // the inserter is publishing
      if (side_level(s) == c.target && side_auxkey(s) != f.auxkey) {     // same-level VIEW tie: the smallest auxkey keeps the slot
        st.ties++;
        while (side_level(s) == c.target && f.auxkey < side_auxkey(s) &&
               !tab.side[i].compare_exchange_weak(s, side_make(c.target, f.auxkey), std::memory_order_acq_rel)) {}
      }
      return false;
    }
    if (++i == tab.slots) i = 0;
    if (probes > tab.slots) throw RepError("seen-set full");
  }
}

// slot of f, or ~0 if f is not in the table
inline u64 lookup(const Table& tab, u64 f) {
  u64 i = tab.home(f);
  for (;;) {
    const u64 cur = tab.fp[i].load(std::memory_order_acquire);
    if (cur == f) return i;
    if (cur == 0) return ~(u64)0;
    if (++i == tab.slots) i = 0;
  }
}

// expand state s of level d (< target) and everything below it that this worker is the first to reach
void descend(Ctx& c, const State& s, u32 d, Stats& st, std::vector<std::vector<Succ>>& pool, FastFp& ffp, std::vector<u64>& rec) {
  std::vector<Succ>& succ = pool[d];
  succ.clear();
  successors(c.P, s, succ);
  st.expanded++;
  const bool last = d + 1 == c.target;
  if (last) {
    if (succ.empty()) st.deadlocks++;
    st.generated += succ.size();
  }
  for (Succ& sc : succ) {
    rec.clear();
    encode(c.P, sc.st, rec);
    const Fp f = ffp(c.P, sc.st, rec.data(), rec.size());
    if (last) {
      st.act[sc.action & 15]++;
      if (c.probe) {
        if (lookup(c.tab, f.fp) != ~(u64)0) continue;             // a state of an earlier level
        const int inv = check_invariants(c.P, sc.st);
        if (inv) {
          st.viol_seen++;
          if (f.fp < st.viol_fp) { st.viol_fp = f.fp; st.viol_mask = inv; }
        }
        continue;
      }
      if (!insert_level(c, f, st)) continue;
      hunt_new(c, sc.st, f.fp, rec.data(), rec.size());
      st.n_new++;
      st.fp_xor ^= f.fp;
      st.fp_sum += f.fp;
      st.max_bag = std::max(st.max_bag, sc.st.messages.size());
      const int inv = check_invariants(c.P, sc.st);
      if (inv && f.fp < st.viol_fp) { st.viol_fp = f.fp; st.viol_mask = inv; }
      continue;
    }
    const u64 i = lookup(c.tab, f.fp);
    if (i == ~(u64)0) throw RepError("lean pass: a successor of a complete level is missing from the seen-set");
    const uint16_t sv = c.tab.side[i].load(std::memory_order_acquire);
    if (side_level(sv) > d + 1) throw RepError("lean pass: a successor carries a deeper level than its parent's + 1");
    if (side_level(sv) != d + 1 || side_auxkey(sv) != f.auxkey) continue;   // an older state, or the loser of a VIEW tie
    const u64 bit = (u64)1 << (i & 63);
    if (c.tab.bits[i >> 6].fetch_or(bit, std::memory_order_acq_rel) & bit) continue;   // expanded already in this pass
    hunt_revisit(c, sc.st, d + 1, f.fp, rec.data(), rec.size());
    descend(c, sc.st, d + 1, st, pool, ffp, rec);
  }
}

void print_level(u32 level, const Stats& s, u64 distinct, double seconds) {
  std::printf("{\"level\": %u, \"new\": %llu, \"generated\": %llu, \"ties\": %llu, \"deadlocks\": %llu, \"distinct\": %llu, \"max_bag\": %zu, "
              "\"fp_xor\": \"%016llx\", \"fp_sum\": \"%016llx\", \"act_generated\": [",
              level, (unsigned long long)s.n_new, (unsigned long long)s.generated, (unsigned long long)s.ties, (unsigned long long)s.deadlocks,
              (unsigned long long)distinct, s.max_bag, (unsigned long long)s.fp_xor, (unsigned long long)s.fp_sum);
  for (int a = 0; a < 16; a++) std::printf("%llu%s", (unsigned long long)s.act[a], a < 15 ? "," : "");
  std::printf("], \"seconds\": %.3f}\n", seconds);
  std::fflush(stdout);
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s R C nValues L --base-level B --slots N [--probe-level P] [--max-depth D] [--threads T] [--inv-mask M] "
                         "[--no-symmetry]\n", argv[0]);
    return 2;
  }
  Ctx c;
  Params& P = c.P;
  P.R = std::atoi(argv[1]); P.C = std::atoi(argv[2]); P.n = std::atoi(argv[3]); P.L = std::atoi(argv[4]);
  int T = (int)std::thread::hardware_concurrency();
  u32 base_level = 0, probe_level = 0, max_depth = 1u << 30;
  u64 slots = 0, verify_every = 4096, hunt_slots = 0;
  bool hunt_set = false;
  for (int i = 5; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--threads" && i + 1 < argc) T = std::atoi(argv[++i]);
    else if (a == "--base-level" && i + 1 < argc) base_level = (u32)std::atoi(argv[++i]);
    else if (a == "--probe-level" && i + 1 < argc) probe_level = (u32)std::atoi(argv[++i]);
    else if (a == "--max-depth" && i + 1 < argc) max_depth = (u32)std::atoi(argv[++i]);
    else if (a == "--slots" && i + 1 < argc) slots = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--inv-mask" && i + 1 < argc) P.invariant_mask = std::atoi(argv[++i]);
    else if (a == "--no-symmetry") P.symmetry = false;
    else if (a == "--assume-commit-number") set_assume_commit_number(P);   // policy for VSR.tla:421 (BASELINE configs[3]: ClientCount = 2)
    else if (a == "--verify-fp-all") verify_every = 1;
    else if (a == "--verify-fp-every" && i + 1 < argc) verify_every = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--hunt-seed" && i + 1 < argc) { c.hunt.on = true; c.hunt.seed = std::strtoull(argv[++i], nullptr, 16); hunt_set = true; }
    else if (a == "--hunt-slots" && i + 1 < argc) hunt_slots = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--hunt-mask" && i + 1 < argc) c.hunt.mask = std::strtoull(argv[++i], nullptr, 16);
    else if (a == "--dump-audit-fp" && i + 1 < argc) c.hunt.dump_fp.store(std::strtoull(argv[++i], nullptr, 16) | 1);
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (T < 1) T = 1;
  if (!base_level || !slots) { std::fprintf(stderr, "--base-level and --slots are required\n"); return 2; }
  if (probe_level && probe_level <= base_level) { std::fprintf(stderr, "--probe-level must lie beyond the base level\n"); return 2; }
  c.tab.alloc(slots);
  if (hunt_set || c.hunt.dump_fp.load()) {
#ifdef ORACLE_HAS_SEEDED_FP
    if (!hunt_set) { std::fprintf(stderr, "--dump-audit-fp needs --hunt-seed (the seed the fingerprint was computed under)\n"); return 2; }
    if (c.hunt.seed == fp_seed()) { std::fprintf(stderr, "--hunt-seed equals the run's own seed (VSR_ORACLE_FP_SEED): nothing to compare\n"); return 2; }
    if (hunt_slots) c.hunt.alloc(hunt_slots);                     // without a set: only --dump-audit-fp
    else if (!c.hunt.dump_fp.load()) { std::fprintf(stderr, "--hunt-seed needs --hunt-slots N (or --dump-audit-fp)\n"); return 2; }
#else
    (void)hunt_slots;
    std::fprintf(stderr, "this oracle has no fingerprint_with_seed: no collision hunt\n");
    return 2;
#endif
  }

  auto run_threads = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& x : th) x.join();
  };

  std::vector<Piece> frontier(T), next(T);
  for (Piece& p : frontier) p.reset();
  u64 distinct = 0, total_generated = 0, n_frontier = 0;
  u32 depth = 1;
  int viol_mask = 0;
  u64 viol_fp = ~0ull;
  size_t max_bag = 0;
  std::string error;
  const char* why = "exhausted";
  try {
    State s0 = init_state(P);
    Fp f = fingerprint(P, s0);
    c.target = 1;
    Stats st;
    insert_level(c, f, st);
    encode(P, s0, frontier[0].words);
    frontier[0].off.push_back(frontier[0].words.size());
    distinct = n_frontier = 1;
    viol_mask = check_invariants(P, s0);
    Stats s1;
    s1.n_new = 1; s1.fp_xor = s1.fp_sum = f.fp;
    print_level(1, s1, 1, 0.0);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  const double t0 = now_s();
  std::vector<Stats> wstats(T);
  std::vector<std::string> werr(T);

  while (!viol_mask) {
    if (depth >= max_depth) { why = "max-depth"; break; }
    const double tl = now_s();
    c.target = depth + 1;
    c.probe = probe_level && c.target == probe_level;
    const bool lean = depth >= base_level;                       // passes over the base level
    if (distinct + 4 * n_frontier > c.tab.slots && !c.probe)
      std::fprintf(stderr, "warning: level %u may overflow the seen-set (%llu slots, %llu states)\n", c.target, (unsigned long long)c.tab.slots, (unsigned long long)distinct);
    if (lean) {                                                   // nothing is expanded twice in one pass
      const u64 nw = (c.tab.slots + 63) / 64;
      run_threads([&](int t) {
        for (u64 k = nw * (u64)t / (u64)T; k < nw * (u64)(t + 1) / (u64)T; k++) c.tab.bits[k].store(0, std::memory_order_relaxed);
      });
    }
    struct Chunk { u32 piece; u32 first, count; };
    std::vector<Chunk> chunks;
    {
      const size_t csz = lean ? 8 : std::max<size_t>(16, std::min<size_t>(4096, n_frontier / ((size_t)T * 16) + 1));
      for (int p = 0; p < T; p++)
        for (size_t a = 0; a < frontier[p].n(); a += csz) chunks.push_back(Chunk{(u32)p, (u32)a, (u32)std::min(csz, frontier[p].n() - a)});
    }
    std::atomic<size_t> cursor{0};
    std::atomic<int> abort_flag{0};
    std::atomic<u64> progress{0};
    const u32 src_level = lean ? base_level : depth;
    run_threads([&](int t) {
      Stats& st = wstats[t];
      st = Stats();
      werr[t].clear();
      Piece& out = next[t];
      if (!lean) out.reset();
      std::vector<std::vector<Succ>> pool(64);
      std::vector<Succ> succ;
      std::vector<u64> rec;
      FastFp ffp;
      ffp.verify_every = verify_every;
      double last_report = now_s();
      try {
        for (;;) {
          const size_t ci = cursor.fetch_add(1, std::memory_order_relaxed);
          if (ci >= chunks.size() || abort_flag.load(std::memory_order_relaxed)) break;
          const Chunk ch = chunks[ci];
          const Piece& pc = frontier[ch.piece];
          for (u32 k = ch.first; k < ch.first + ch.count; k++) {
            State s = decode(P, &pc.words[pc.off[k]], nullptr);
            if (lean) {
              if (c.hunt.on && c.hunt.dump_fp.load(std::memory_order_relaxed))
                hunt_revisit(c, s, src_level, 0, &pc.words[pc.off[k]], (size_t)(pc.off[k + 1] - pc.off[k]));
              descend(c, s, src_level, st, pool, ffp, rec);
              continue;
            }
            // ---- ordinary level: expand, insert, keep the records of the new states
            succ.clear();
            successors(P, s, succ);
            st.expanded++;
            if (succ.empty()) st.deadlocks++;
            st.generated += succ.size();
            for (Succ& sc : succ) {
              st.act[sc.action & 15]++;
              rec.clear();
              encode(P, sc.st, rec);
              const Fp f = ffp(P, sc.st, rec.data(), rec.size());
              if (!insert_level(c, f, st)) continue;
              hunt_new(c, sc.st, f.fp, rec.data(), rec.size());
              out.words.insert(out.words.end(), rec.begin(), rec.end());
              out.off.push_back(out.words.size());
              st.n_new++;
              st.fp_xor ^= f.fp;
              st.fp_sum += f.fp;
              st.max_bag = std::max(st.max_bag, sc.st.messages.size());
              const int inv = check_invariants(P, sc.st);
              if (inv && f.fp < st.viol_fp) { st.viol_fp = f.fp; st.viol_mask = inv; }
            }
          }
          if (t == 0 && now_s() - last_report > 120.0) {
            last_report = now_s();
            std::fprintf(stderr, "  level %u: chunk %zu of %zu, %llu states expanded by this worker, %.0f s\n", c.target, ci, chunks.size(),
                         (unsigned long long)st.expanded, last_report - tl);
          }
        }
      } catch (const std::exception& e) {
        werr[t] = e.what();
        abort_flag.store(1);
      }
    });
    bool failed = false;
    for (int t = 0; t < T; t++)
      if (!werr[t].empty()) { error = werr[t]; failed = true; }
    if (failed) { why = "error"; break; }
    Stats lv;
    for (Stats& w : wstats) lv.add(w);
    total_generated += lv.generated;
    if (c.probe) {
      std::printf("{\"probe_level\": %u, \"generated\": %llu, \"deadlocks\": %llu, \"violating_successors\": %llu, \"viol_fp\": \"%016llx\", "
                  "\"viol_mask\": %d, \"expanded\": %llu, \"seconds\": %.3f}\n",
                  c.target, (unsigned long long)lv.generated, (unsigned long long)lv.deadlocks, (unsigned long long)lv.viol_seen,
                  (unsigned long long)(lv.viol_mask ? lv.viol_fp : 0), lv.viol_mask, (unsigned long long)lv.expanded, now_s() - tl);
      std::fflush(stdout);
      if (lv.viol_mask) { viol_mask = lv.viol_mask; viol_fp = lv.viol_fp; why = "violation"; depth++; }
      else why = "probe";
      break;
    }
    if (lv.n_new == 0) break;
    if (!lean) {
      for (int t = 0; t < T; t++) std::swap(frontier[t], next[t]);
      if (depth + 1 >= base_level)
        for (int t = 0; t < T; t++) {
          next[t].words.clear(); next[t].words.shrink_to_fit(); next[t].off.clear(); next[t].off.shrink_to_fit();
          if (depth + 1 == base_level) { frontier[t].words.shrink_to_fit(); frontier[t].off.shrink_to_fit(); }
        }
    }
    max_bag = std::max(max_bag, lv.max_bag);
    n_frontier = lv.n_new;
    distinct += lv.n_new;
    depth++;
    print_level(depth, lv, distinct, now_s() - tl);
    if (lv.ties) std::fprintf(stderr, "warning: level %u has %llu same-level VIEW ties\n", depth, (unsigned long long)lv.ties);
    if (lv.viol_mask) { viol_mask = lv.viol_mask; viol_fp = lv.viol_fp; why = "violation"; break; }
  }
  const double dt = now_s() - t0;
  std::printf("{\"summary\": true, \"stop\": \"%s\", \"depth\": %u, \"distinct\": %llu, \"generated\": %llu, \"seconds\": %.3f, \"states_per_s\": %.1f, "
              "\"max_bag\": %zu, \"viol_mask\": %d, \"viol_fp\": \"%016llx\", \"error\": \"%s\", \"threads\": %d, \"fp_version\": %d, "
              "\"base_level\": %u, \"slots\": %llu, \"audit_collisions\": %llu}\n",
              why, depth, (unsigned long long)distinct, (unsigned long long)total_generated, dt, distinct / (dt > 0 ? dt : 1e-9), max_bag,
              viol_mask, (unsigned long long)(viol_mask ? viol_fp : 0), error.c_str(), T, FP_VERSION, base_level, (unsigned long long)c.tab.slots,
              (unsigned long long)c.hunt.found.load());
  return 0;
}

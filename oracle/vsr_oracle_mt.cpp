// oracle/vsr_oracle_mt.cpp — CPU ORACLE, multi-threaded driver (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// The same level-synchronous BFS as vsr_oracle_bfs.cpp — successors() / fingerprint() / check_invariants() of vsr_oracle.cpp,
// i.e. the restatement of VSR.tla — spread over std::thread workers that share ONE lock-free seen-set, the way TLC spreads
// Worker threads over one FPSet (SURVEY.md §3.1, §8d "CPU baseline timed beside it").  Two uses:
//   * bench.py's `cpu_baseline` leg ("port", cores = threads);
//   * the whole-workload fixtures of tests/golden/oracle_levels_*.json (tools/make_oracle_levels.py): per level the number of
//     new states, successors generated (total and per action), deadlocks, largest bag, xor and sum of the new fingerprints —
//     every figure is independent of the order in which the workers discover the states.
// Checked against the single-threaded oracle's level fixtures by tests/test_oracle_mt.py.
//
// Per level: the frontier (T pieces, one per worker of the previous level) is cut into chunks drawn from an atomic cursor;
// a worker expands a chunk, fingerprints every successor and inserts it into the open-addressing table with a CAS on the
// fingerprint word.  The worker whose CAS inserts the fingerprint owns the new state: it checks the invariants and appends the
// packed record to its own piece of the next frontier.  A same-level duplicate whose canonical auxkey differs from the
// owner's (the VIEW tie of SURVEY F2, never observed) is set aside and resolved after the level's barrier exactly like the
// single-threaded oracle does: the smallest canonical auxkey keeps the slot.  The table grows (parallel rehash) between levels.
// CLI: vsr_oracle_mt R C nValues L [--threads T] [--max-depth D] [--max-seconds S] [--max-states N] [--no-symmetry]
//                    [--assume-commit-number] [--inv-mask M] [--count-only-from D] [--quiet]
// --count-only-from D: the states of levels >= D are inserted, counted, checksummed and invariant-checked, but their records are
// not kept (the search stops after the first such level): one more level for the same memory.
#include <sys/mman.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#ifndef ORACLE_HPP          // model-independent driver, compiled once per restatement (see vsr_oracle_bfs.cpp)
#define ORACLE_HPP "vsr_oracle.hpp"
#define ORACLE_NS vsr_oracle
#endif
#include ORACLE_HPP

using namespace ORACLE_NS;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---- seen-set: fp word (0 = empty) + meta word (0 = not yet published) per slot, linear probing ---------------------------
// meta = level(12) << 52 | auxkey(12) << 40 | worker(10) << 30 | index in the worker's piece (30) ... + 1 so that it is never 0
struct Table {
  std::atomic<u64>* fp = nullptr;
  std::atomic<u64>* meta = nullptr;
  u64 slots = 0, mask = 0;
  static void* map(u64 bytes) {
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (p == MAP_FAILED) { std::fprintf(stderr, "error: mmap of %llu bytes failed\n", (unsigned long long)bytes); std::exit(1); }
    return p;
  }
  void alloc(u64 n) {
    slots = n;
    mask = n - 1;
    fp = (std::atomic<u64>*)map(n * 8);     // anonymous pages are zero: every slot empty
    meta = (std::atomic<u64>*)map(n * 8);
  }
  void release() {
    if (fp) munmap((void*)fp, slots * 8);
    if (meta) munmap((void*)meta, slots * 8);
    fp = meta = nullptr;
  }
};

inline u64 meta_make(u32 level, u32 auxkey, u32 worker, u64 idx) {
  return (((u64)level << 52) | ((u64)auxkey << 40) | ((u64)worker << 30) | idx) + 1;
}
inline u32 meta_level(u64 m) { return (u32)((m - 1) >> 52); }
inline u32 meta_auxkey(u64 m) { return (u32)(((m - 1) >> 40) & 0xFFF); }
inline u32 meta_worker(u64 m) { return (u32)(((m - 1) >> 30) & 0x3FF); }
inline u64 meta_idx(u64 m) { return (m - 1) & 0x3FFFFFFFull; }

struct Piece {                 // one worker's part of a frontier
  std::vector<u64> words;
  std::vector<u64> off;        // n + 1 offsets
  size_t n() const { return off.empty() ? 0 : off.size() - 1; }
  void reset() { words.clear(); off.assign(1, 0); }
};

struct Tie { u64 fp; u32 auxkey; std::vector<u64> rec; };

struct Worker {
  Piece out;
  std::vector<Tie> ties;
  u64 generated = 0, deadlocks = 0, n_new = 0, fp_xor = 0, fp_sum = 0, viol_fp = ~0ull;
  u64 act[16] = {0};
  int viol_mask = 0;
  size_t max_bag = 0;          // largest bag among the states this worker added (this level)
  std::string error;
  int error_code = 0;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s R C nValues L [--threads T] [--max-depth D] [--max-seconds S] [--max-states N] [--no-symmetry] "
                         "[--assume-commit-number] [--inv-mask M] [--count-only-from D] [--quiet]\n", argv[0]);
    return 2;
  }
  Params P;
  P.R = std::atoi(argv[1]); P.C = std::atoi(argv[2]); P.n = std::atoi(argv[3]); P.L = std::atoi(argv[4]);
  int T = (int)std::thread::hardware_concurrency();
  int max_depth = 1 << 30;
  double max_seconds = 1e30;
  u64 max_states = ~0ull;
  bool quiet = false;
  int count_only_from = 1 << 30;
  for (int i = 5; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--threads" && i + 1 < argc) T = std::atoi(argv[++i]);
    else if (a == "--max-depth" && i + 1 < argc) max_depth = std::atoi(argv[++i]);
    else if (a == "--max-seconds" && i + 1 < argc) max_seconds = std::atof(argv[++i]);
    else if (a == "--max-states" && i + 1 < argc) max_states = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--inv-mask" && i + 1 < argc) P.invariant_mask = std::atoi(argv[++i]);
    else if (a == "--count-only-from" && i + 1 < argc) count_only_from = std::atoi(argv[++i]);
    else if (a == "--no-symmetry") P.symmetry = false;
#ifndef ORACLE_VRST
    else if (a == "--assume-commit-number") P.assume_commit_number = true;
#endif
    else if (a == "--quiet") quiet = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (T < 1) T = 1;
  if (T > 1023) T = 1023;

  Table tab;
  tab.alloc((u64)1 << 16);
  std::vector<Worker> workers(T);
  std::vector<Piece> frontier(T);
  for (Piece& p : frontier) p.reset();
  u64 distinct = 0, total_generated = 0, n_frontier = 0;
  int depth = 1, viol_mask = 0;
  u64 viol_fp = ~0ull;
  std::string error;
  size_t max_bag = 0;
  try {
    State s0 = init_state(P);
    Fp f = fingerprint(P, s0);
    u64 i = f.fp & tab.mask;
    tab.fp[i].store(f.fp);
    tab.meta[i].store(meta_make(1, f.auxkey, 0, 0));
    encode(P, s0, frontier[0].words);
    frontier[0].off.push_back(frontier[0].words.size());
    distinct = n_frontier = 1;
    viol_mask = check_invariants(P, s0);
    if (!quiet)
      std::printf("{\"level\": 1, \"new\": 1, \"generated\": 0, \"ties\": 0, \"deadlocks\": 0, \"distinct\": 1, \"max_bag\": 0, "
                  "\"fp_xor\": \"%016llx\", \"fp_sum\": \"%016llx\", \"act_generated\": [0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0], \"seconds\": 0.000}\n",
                  (unsigned long long)f.fp, (unsigned long long)f.fp);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  const double t0 = now_s();
  const char* why = "exhausted";
  auto run_threads = [&](auto&& fn) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back([&fn, t]() { fn(t); });
    fn(0);
    for (auto& x : th) x.join();
  };

  while (!viol_mask) {
    if (depth >= max_depth) { why = "max-depth"; break; }
    if (now_s() - t0 > max_seconds) { why = "max-seconds"; break; }
    if (distinct > max_states) { why = "max-states"; break; }
    const double tl = now_s();
    const u32 new_level = (u32)depth + 1;
    // ---- room for this level: at most `generated` new states; out-degree is bounded by the ordinal count, but a level has
    // never grown more than x4 — the table is doubled until load <= 1/2 under that bound (an overflow aborts the run)
    u64 need = 2 * (distinct + 4 * n_frontier + 1024);
    if (need > tab.slots) {
      u64 ns = tab.slots;
      while (ns < need) ns <<= 1;
      Table nt;
      nt.alloc(ns);
      const u64 old_slots = tab.slots;
      run_threads([&](int t) {
        const u64 lo = old_slots * (u64)t / (u64)T, hi = old_slots * (u64)(t + 1) / (u64)T;
        for (u64 k = lo; k < hi; k++) {
          const u64 f = tab.fp[k].load(std::memory_order_relaxed);
          if (!f) continue;
          u64 i = f & nt.mask;
          for (;;) {
            u64 exp = 0;
            if (nt.fp[i].compare_exchange_strong(exp, f, std::memory_order_relaxed)) break;
            i = (i + 1) & nt.mask;
          }
          nt.meta[i].store(tab.meta[k].load(std::memory_order_relaxed), std::memory_order_relaxed);
        }
      });
      tab.release();
      tab = nt;
    }
    // ---- work list: chunks of (piece, first, count)
    struct Chunk { u32 piece; u32 first, count; };
    std::vector<Chunk> chunks;
    {
      const size_t csz = std::max<size_t>(16, std::min<size_t>(4096, n_frontier / ((size_t)T * 16) + 1));
      for (int p = 0; p < T; p++)
        for (size_t a = 0; a < frontier[p].n(); a += csz) chunks.push_back(Chunk{(u32)p, (u32)a, (u32)std::min(csz, frontier[p].n() - a)});
    }
    std::atomic<size_t> cursor{0};
    std::atomic<int> abort_flag{0};                              // 1 = a worker raised an error, 2 = --max-seconds passed inside the level
    run_threads([&](int t) {
      Worker& w = workers[t];
      w.out.reset();
      w.ties.clear();
      w.generated = w.deadlocks = w.n_new = w.fp_xor = w.fp_sum = 0;
      w.viol_fp = ~0ull;
      w.viol_mask = 0;
      w.max_bag = 0;
      w.error_code = 0;
      for (u64& a : w.act) a = 0;
      std::vector<Succ> succ;
      try {
        for (;;) {
          const size_t ci = cursor.fetch_add(1, std::memory_order_relaxed);
          if (ci >= chunks.size() || abort_flag.load(std::memory_order_relaxed)) break;
          if (t == 0 && (ci & 63) == 0 && now_s() - t0 > max_seconds) {   // the bound also ends a level in progress (it is discarded)
            abort_flag.store(2);
            break;
          }
          const Chunk c = chunks[ci];
          const Piece& pc = frontier[c.piece];
          for (u32 k = c.first; k < c.first + c.count; k++) {
            State s = decode(P, &pc.words[pc.off[k]], nullptr);
            succ.clear();
            successors(P, s, succ);
            if (succ.empty()) w.deadlocks++;
            w.generated += succ.size();
            for (Succ& sc : succ) {
              w.act[sc.action & 15]++;
              const Fp f = fingerprint(P, sc.st);
              u64 i = f.fp & tab.mask;
              bool mine = false;
              u64 probes = 0;
              for (;;) {
                u64 cur = tab.fp[i].load(std::memory_order_acquire);
                if (cur == 0) {
                  u64 exp = 0;
                  if (tab.fp[i].compare_exchange_strong(exp, f.fp, std::memory_order_acq_rel)) { mine = true; break; }
                  cur = exp;
                }
                if (cur == f.fp) break;
                i = (i + 1) & tab.mask;
                if (++probes > tab.mask) throw RepError("seen-set full");
              }
              if (mine) {
                const u64 idx = w.out.n();
                if (idx >= ((u64)1 << 30)) throw RepError("more than 2^30 new states in one worker's piece");
                tab.meta[i].store(meta_make(new_level, f.auxkey, (u32)t, idx), std::memory_order_release);
                if ((int)new_level < count_only_from) encode(P, sc.st, w.out.words);
                w.out.off.push_back(w.out.words.size());
                w.n_new++;
                w.fp_xor ^= f.fp;
                w.fp_sum += f.fp;
                w.max_bag = std::max(w.max_bag, sc.st.messages.size());
                const int inv = check_invariants(P, sc.st);
                if (inv && f.fp < w.viol_fp) { w.viol_fp = f.fp; w.viol_mask = inv; }
              } else {
                u64 m;
                while ((m = tab.meta[i].load(std::memory_order_acquire)) == 0) std::this_thread::yield();   // owner is publishing
                if (meta_level(m) == new_level && meta_auxkey(m) != f.auxkey) {
                  Tie tie;
                  tie.fp = f.fp;
                  tie.auxkey = f.auxkey;
                  encode(P, sc.st, tie.rec);
                  w.ties.push_back(std::move(tie));
                }
              }
            }
          }
        }
      } catch (const EvalError& e) {
        w.error = e.what();
        w.error_code = -1;
        abort_flag.store(1);
      } catch (const RepError& e) {
        w.error = e.what();
        w.error_code = -2;
        abort_flag.store(1);
      }
    });
    bool failed = false;
    for (Worker& w : workers)
      if (w.error_code) { error = w.error; failed = true; }
    if (failed) { why = "error"; break; }
    if (abort_flag.load() == 2) { why = "max-seconds"; break; }
    // ---- same-level VIEW ties: the smallest canonical auxkey keeps the slot (sequential; never observed)
    u64 ties = 0;
    for (Worker& w : workers)
      for (Tie& tie : w.ties) {
        ties++;
        u64 i = tie.fp & tab.mask;
        while (tab.fp[i].load() != tie.fp) i = (i + 1) & tab.mask;
        const u64 m = tab.meta[i].load();
        if (tie.auxkey < meta_auxkey(m) && (int)new_level < count_only_from) {
          Worker& ow = workers[meta_worker(m)];
          const u64 k = meta_idx(m);
          std::copy(tie.rec.begin(), tie.rec.end(), ow.out.words.begin() + ow.out.off[k]);   // same VIEW: same length
          tab.meta[i].store(meta_make(new_level, tie.auxkey, meta_worker(m), k));
          // the invariant verdict depends on the aux variables: re-evaluate for the replaced record
          State s = decode(P, &ow.out.words[ow.out.off[k]], nullptr);
          const int inv = check_invariants(P, s);
          if (inv && tie.fp < ow.viol_fp) { ow.viol_fp = tie.fp; ow.viol_mask = inv; }
        }
      }
    // ---- level summary
    u64 nn = 0, gen = 0, dl = 0, fx = 0, fs = 0, act[16] = {0};
    size_t lvl_bag = 0;
    for (Worker& w : workers) {
      nn += w.n_new; gen += w.generated; dl += w.deadlocks; fx ^= w.fp_xor; fs += w.fp_sum;
      lvl_bag = std::max(lvl_bag, w.max_bag);
      for (int a = 0; a < 16; a++) act[a] += w.act[a];
      if (w.viol_mask && w.viol_fp < viol_fp) { viol_fp = w.viol_fp; viol_mask = w.viol_mask; }
    }
    max_bag = std::max(max_bag, lvl_bag);
    total_generated += gen;
    if (nn == 0) break;
    for (int t = 0; t < T; t++) std::swap(frontier[t], workers[t].out);
    n_frontier = nn;
    distinct += nn;
    depth++;
    if (!quiet) {
      std::printf("{\"level\": %d, \"new\": %llu, \"generated\": %llu, \"ties\": %llu, \"deadlocks\": %llu, \"distinct\": %llu, \"max_bag\": %zu, "
                  "\"fp_xor\": \"%016llx\", \"fp_sum\": \"%016llx\", \"act_generated\": [",
                  depth, (unsigned long long)nn, (unsigned long long)gen, (unsigned long long)ties, (unsigned long long)dl,
                  (unsigned long long)distinct, lvl_bag, (unsigned long long)fx, (unsigned long long)fs);
      for (int a = 0; a < 16; a++) std::printf("%llu%s", (unsigned long long)act[a], a < 15 ? "," : "");
      std::printf("], \"seconds\": %.3f}\n", now_s() - tl);
      std::fflush(stdout);
    }
    if (viol_mask) { why = "violation"; break; }
    if (depth >= count_only_from) { why = "count-only"; break; }
  }
  const double dt = now_s() - t0;
  std::printf("{\"summary\": true, \"stop\": \"%s\", \"depth\": %d, \"distinct\": %llu, \"generated\": %llu, \"seconds\": %.3f, \"states_per_s\": %.1f, "
              "\"max_bag\": %zu, \"viol_mask\": %d, \"viol_fp\": \"%016llx\", \"error\": \"%s\", \"threads\": %d, \"fp_version\": %d}\n",
              why, depth, (unsigned long long)distinct, (unsigned long long)total_generated, dt, distinct / (dt > 0 ? dt : 1e-9), max_bag,
              viol_mask, (unsigned long long)(viol_mask ? viol_fp : 0), error.c_str(), T, FP_VERSION);
  return 0;
}

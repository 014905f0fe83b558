// oracle/vsr_oracle_mt.cpp — CPU ORACLE, multi-threaded timing driver (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// The same level-synchronous BFS as vsr_oracle_bfs.cpp — successors() / fingerprint() / check_invariants() of vsr_oracle.cpp,
// i.e. the restatement of VSR.tla — spread over std::thread workers, the way TLC spreads Worker threads over one FPSet
// (SURVEY.md §3.1, §8d "CPU baseline timed beside it").  It exists for bench.py's `cpu_baseline` leg ("port", cores = threads)
// and is checked against the single-threaded oracle's level counts by tests/test_oracle_mt.py.
//
// Per level:  (1) the frontier is cut into T contiguous slices; worker t expands its slice and files every successor
//                 (fp, auxkey, violated-invariant mask, packed record) under shard = fp >> 58 (64 shards);
//             (2) worker s owns shards s, s+T, ...: it walks the filed successors of its shards in slice order (= frontier
//                 order, so "first discoverer wins" and the smallest-auxkey tie rule behave exactly as in the single-threaded
//                 oracle), inserts into the shard's hash map and appends the new states to the shard's part of the next frontier.
// CLI: vsr_oracle_mt R C nValues L [--threads T] [--max-depth D] [--max-seconds S] [--no-symmetry] [--assume-commit-number] [--quiet]
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "vsr_oracle.hpp"

using namespace vsr_oracle;

namespace {

const int NSHARD = 64;

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Filed {        // one generated successor, filed under its shard
  u64 fp;
  u32 auxkey;
  u32 inv;
  u32 off, len;       // record words in the owning worker's word pool
};
struct SeenEntry { u32 level; u32 auxkey; u64 slot; };   // slot = index in the shard's next-frontier part

struct Worker {
  std::vector<Filed> filed[NSHARD];
  std::vector<u64> pool;
  u64 generated = 0, deadlocks = 0;
  size_t max_bag = 0;
  std::string error;
  int error_code = 0;
};

struct ShardOut {
  std::vector<u64> words;
  std::vector<u64> off;        // n+1 offsets
  u64 n_new = 0, ties = 0;
  int viol_mask = 0;
  u64 viol_fp = ~0ull;
};

}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s R C nValues L [--threads T] [--max-depth D] [--max-seconds S] [--no-symmetry] [--assume-commit-number] [--quiet]\n", argv[0]);
    return 2;
  }
  Params P;
  P.R = std::atoi(argv[1]); P.C = std::atoi(argv[2]); P.n = std::atoi(argv[3]); P.L = std::atoi(argv[4]);
  int T = (int)std::thread::hardware_concurrency();
  int max_depth = 1 << 30;
  double max_seconds = 1e30;
  bool quiet = false;
  for (int i = 5; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--threads" && i + 1 < argc) T = std::atoi(argv[++i]);
    else if (a == "--max-depth" && i + 1 < argc) max_depth = std::atoi(argv[++i]);
    else if (a == "--max-seconds" && i + 1 < argc) max_seconds = std::atof(argv[++i]);
    else if (a == "--no-symmetry") P.symmetry = false;
    else if (a == "--assume-commit-number") P.assume_commit_number = true;
    else if (a == "--quiet") quiet = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  if (T < 1) T = 1;
  if (T > 256) T = 256;

  std::vector<std::unordered_map<u64, SeenEntry>> seen(NSHARD);
  std::vector<u64> fr_words, fr_off;
  u64 distinct = 0, total_generated = 0;
  int depth = 1, viol_mask = 0;
  std::string error;
  size_t max_bag = 0;
  try {
    State s0 = init_state(P);
    Fp f = fingerprint(P, s0);
    seen[f.fp >> 58][f.fp] = SeenEntry{1, f.auxkey, 0};
    fr_off.push_back(0);
    encode(P, s0, fr_words);
    fr_off.push_back(fr_words.size());
    distinct = 1;
    viol_mask = check_invariants(P, s0);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  const double t0 = now_s();
  const char* why = "exhausted";
  std::vector<Worker> workers(T);
  std::vector<ShardOut> outs(NSHARD);
  while (!viol_mask) {
    if (depth >= max_depth) { why = "max-depth"; break; }
    if (now_s() - t0 > max_seconds) { why = "max-seconds"; break; }
    const double tl = now_s();
    const size_t nfront = fr_off.size() - 1;
    const u32 new_level = (u32)depth + 1;
    // ---- phase 1: expand
    std::vector<std::thread> th;
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        Worker& w = workers[t];
        for (int s = 0; s < NSHARD; s++) w.filed[s].clear();
        w.pool.clear();
        w.generated = w.deadlocks = 0;
        w.error_code = 0;
        const size_t lo = nfront * (size_t)t / (size_t)T, hi = nfront * (size_t)(t + 1) / (size_t)T;
        std::vector<Succ> succ;
        try {
          for (size_t i = lo; i < hi; i++) {
            State s = decode(P, &fr_words[fr_off[i]], nullptr);
            succ.clear();
            successors(P, s, succ);
            if (succ.empty()) w.deadlocks++;
            w.generated += succ.size();
            for (Succ& sc : succ) {
              Fp f = fingerprint(P, sc.st);
              w.max_bag = std::max(w.max_bag, sc.st.messages.size());
              size_t before = w.pool.size();
              encode(P, sc.st, w.pool);
              w.filed[f.fp >> 58].push_back(Filed{f.fp, f.auxkey, (u32)check_invariants(P, sc.st), (u32)before, (u32)(w.pool.size() - before)});
            }
          }
        } catch (const EvalError& e) {
          w.error = e.what();
          w.error_code = -1;
        } catch (const RepError& e) {
          w.error = e.what();
          w.error_code = -2;
        }
      });
    for (auto& x : th) x.join();
    th.clear();
    bool failed = false;
    for (Worker& w : workers)
      if (w.error_code) { error = w.error; failed = true; }
    if (failed) { why = "error"; break; }
    // ---- phase 2: per-shard insert, in frontier order
    for (int t = 0; t < T; t++)
      th.emplace_back([&, t]() {
        for (int s = t; s < NSHARD; s += T) {
          ShardOut& o = outs[s];
          o.words.clear();
          o.off.assign(1, 0);
          o.n_new = o.ties = 0;
          o.viol_mask = 0;
          o.viol_fp = ~0ull;
          auto& map = seen[s];
          for (int wt = 0; wt < T; wt++) {
            const Worker& w = workers[wt];
            for (const Filed& c : w.filed[s]) {
              auto it = map.find(c.fp);
              if (it == map.end()) {
                map.emplace(c.fp, SeenEntry{new_level, c.auxkey, o.n_new});
                o.words.insert(o.words.end(), w.pool.begin() + c.off, w.pool.begin() + c.off + c.len);
                o.off.push_back(o.words.size());
                o.n_new++;
                if (c.inv && c.fp < o.viol_fp) { o.viol_fp = c.fp; o.viol_mask = (int)c.inv; }
              } else if (it->second.level == new_level && it->second.auxkey != c.auxkey) {
                o.ties++;                                    // same-level VIEW collision: smallest canonical auxkey survives
                if (c.auxkey < it->second.auxkey) {
                  it->second.auxkey = c.auxkey;
                  u64 k = it->second.slot;
                  std::copy(w.pool.begin() + c.off, w.pool.begin() + c.off + c.len, o.words.begin() + o.off[k]);
                }
              }
            }
          }
        }
      });
    for (auto& x : th) x.join();
    // ---- next frontier = concatenation of the shards' parts
    u64 nn = 0, gen = 0, dl = 0, ties = 0, vfp = ~0ull;
    fr_words.clear();
    fr_off.assign(1, 0);
    for (ShardOut& o : outs) {
      for (u64 k = 0; k < o.n_new; k++) {
        fr_words.insert(fr_words.end(), o.words.begin() + o.off[k], o.words.begin() + o.off[k + 1]);
        fr_off.push_back(fr_words.size());
      }
      nn += o.n_new;
      ties += o.ties;
      if (o.viol_mask && o.viol_fp < vfp) { vfp = o.viol_fp; viol_mask = o.viol_mask; }
    }
    for (Worker& w : workers) { gen += w.generated; dl += w.deadlocks; max_bag = std::max(max_bag, w.max_bag); }
    total_generated += gen;
    if (nn == 0) break;
    distinct += nn;
    depth++;
    if (!quiet)
      std::printf("{\"level\": %d, \"new\": %llu, \"generated\": %llu, \"ties\": %llu, \"deadlocks\": %llu, \"distinct\": %llu, \"seconds\": %.3f}\n",
                  depth, (unsigned long long)nn, (unsigned long long)gen, (unsigned long long)ties, (unsigned long long)dl,
                  (unsigned long long)distinct, now_s() - tl);
    if (viol_mask) { why = "violation"; break; }
  }
  const double dt = now_s() - t0;
  std::printf("{\"summary\": true, \"stop\": \"%s\", \"depth\": %d, \"distinct\": %llu, \"generated\": %llu, \"seconds\": %.3f, \"states_per_s\": %.1f, \"max_bag\": %zu, \"viol_mask\": %d, \"error\": \"%s\", \"threads\": %d}\n",
              why, depth, (unsigned long long)distinct, (unsigned long long)total_generated, dt, distinct / (dt > 0 ? dt : 1e-9), max_bag,
              viol_mask, error.c_str(), T);
  return 0;
}

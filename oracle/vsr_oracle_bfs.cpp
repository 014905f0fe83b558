// oracle/vsr_oracle_bfs.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
// Level-synchronous BFS over the oracle's successor function + a flat C API for ctypes + a small CLI.
// Restates the TLC loop of SURVEY.md §3.1 (Worker.run / ModelChecker.doNext / FPSet.put / TLCTrace):
//   dequeue state -> all actions -> canonical VIEW fingerprint -> seen-set put -> if new: invariant, enqueue.
// "parity unpinned" vs TLC (see vsr_oracle.hpp).
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <thread>
#include <unordered_map>

// The driver is model-independent: it is compiled once per restatement (ORACLE_HPP / ORACLE_NS: vsr_oracle.hpp = VSR.tla, the
// default; vrst_oracle.hpp = analysis/03-state-transfer/VR_STATE_TRANSFER.tla -> liborc2.so, vrst_oracle, vrst_oracle_mt).
#ifndef ORACLE_HPP
#define ORACLE_HPP "vsr_oracle.hpp"
#define ORACLE_NS vsr_oracle
#endif
#include ORACLE_HPP

using namespace ORACLE_NS;

namespace {

thread_local std::string g_err;

Params params_from(const int* p) { return params_from_array(p); }

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct SeenEntry {
  u32 level;
  u32 auxkey;
  u64 gid;   // global id = discovery order with the level-synchronous tie rule applied
};

struct Bfs {
  Params P;
  std::unordered_map<u64, SeenEntry> seen;
  // trace store (TLCTrace): per gid fingerprint and parent gid
  std::vector<u64> fp_of;
  std::vector<u64> parent_of;
  std::vector<u64> level_base;     // gid of the first state of each level (level 1 = Init)
  // frontier = newest completed level, packed stream + offsets
  std::vector<u64> fr_words;
  std::vector<u64> fr_off;
  // stats of the newest level
  u64 generated = 0, n_new = 0, ties = 0, deadlocks = 0, total_generated = 0;
  int depth = 0;                    // number of completed levels (Init = 1)
  int viol_mask = 0;
  u64 viol_gid = ~0ull;
  u64 viol_fp = ~0ull;
  std::string error;
  int error_code = 0;               // 0 ok, -1 EvalError, -2 RepError
  double level_seconds = 0;
  size_t max_bag = 0;

  void start() {
    State s0 = init_state(P);
    Fp f = fingerprint(P, s0);
    seen.reserve(1 << 20);
    seen[f.fp] = SeenEntry{1, f.auxkey, 0};
    fp_of.push_back(f.fp);
    parent_of.push_back(~0ull);
    level_base.push_back(0);
    fr_words.clear();
    fr_off.clear();
    fr_off.push_back(0);
    encode(P, s0, fr_words);
    fr_off.push_back(fr_words.size());
    depth = 1;
    n_new = 1;
    generated = 0;
    int bad = check_invariants(P, s0);
    if (bad) { viol_mask = bad; viol_gid = 0; viol_fp = f.fp; }
  }

  // expand the newest level; returns number of new states (0 = search finished)
  u64 step() {
    double t0 = now_s();
    std::vector<u64> nx_words;
    std::vector<u64> nx_off;
    nx_off.push_back(0);
    u64 base_gid = fp_of.size();
    u64 cur_base = level_base.back();
    u64 gen = 0, dl = 0, tie = 0;
    u32 new_level = (u32)depth + 1;
    size_t nfront = fr_off.size() - 1;
    std::vector<Succ> succ;
    std::vector<u64> tmp;
    try {
      for (size_t i = 0; i < nfront; i++) {
        State s = decode(P, &fr_words[fr_off[i]], nullptr);
        succ.clear();
        successors(P, s, succ);
        if (succ.empty()) dl++;
        gen += succ.size();
        for (Succ& sc : succ) {
          Fp f = fingerprint(P, sc.st);
          max_bag = std::max(max_bag, sc.st.messages.size());
          auto it = seen.find(f.fp);
          if (it == seen.end()) {
            u64 gid = fp_of.size();
            seen.emplace(f.fp, SeenEntry{new_level, f.auxkey, gid});
            fp_of.push_back(f.fp);
            parent_of.push_back(cur_base + i);
            encode(P, sc.st, nx_words);
            nx_off.push_back(nx_words.size());
          } else if (it->second.level == new_level && it->second.auxkey != f.auxkey) {
            // Same-level VIEW collision with different aux variables.  Deterministic
            // rule: the representative with the smallest canonical auxkey survives.
            tie++;
            if (f.auxkey < it->second.auxkey) {
              it->second.auxkey = f.auxkey;
              u64 k = it->second.gid - base_gid;
              tmp.clear();
              encode(P, sc.st, tmp);
              if (tmp.size() != nx_off[k + 1] - nx_off[k]) throw RepError("tie replacement changed the record size");
              std::copy(tmp.begin(), tmp.end(), nx_words.begin() + nx_off[k]);
              parent_of[it->second.gid] = cur_base + i;
            }
          }
        }
      }
    } catch (const EvalError& e) {
      error = e.what();
      error_code = -1;
    } catch (const RepError& e) {
      error = e.what();
      error_code = -2;
    }
    if (error_code) {   // a TLC evaluation error aborts the run; the partial level is not committed
      generated = gen;
      n_new = 0;
      level_seconds = now_s() - t0;
      return 0;
    }
    // invariants are evaluated on the survivors of the level (B7: on each new state)
    size_t nn = nx_off.size() - 1;
    for (size_t k = 0; k < nn && error_code == 0; k++) {
      State s = decode(P, &nx_words[nx_off[k]], nullptr);
      int bad = check_invariants(P, s);
      if (bad && fp_of[base_gid + k] < viol_fp) {   // report the violating state with the smallest fingerprint
        if (viol_gid == ~0ull || (viol_gid >= base_gid)) {
          viol_mask = bad;
          viol_gid = base_gid + k;
          viol_fp = fp_of[base_gid + k];
        }
      }
    }
    generated = gen;
    total_generated += gen;
    deadlocks = dl;
    ties = tie;
    n_new = nn;
    if (nn > 0) {
      level_base.push_back(base_gid);
      fr_words.swap(nx_words);
      fr_off.swap(nx_off);
      depth++;
    } else {
      fr_words.clear();
      fr_off.assign(1, 0);
    }
    level_seconds = now_s() - t0;
    return nn;
  }
};

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}

}  // namespace

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }
int orc_fp_version() { return FP_VERSION; }
void orc_set_fp_seed(u64 seed) { set_fp_seed(seed); }
u64 orc_fp_seed() { return fp_seed(); }

int orc_layout(const int* params, int* wpr, int* fixed) {
  Params P = params_from(params);
  *wpr = words_per_replica(P);
  *fixed = fixed_words(P);
  return 0;
}

// writes the packed Init record; returns its length in words
int orc_init_record(const int* params, u64* out, int cap) {
  try {
    Params P = params_from(params);
    std::vector<u64> rec;
    encode(P, init_state(P), rec);
    if ((int)rec.size() > cap) return fail(-3, "buffer too small");
    std::copy(rec.begin(), rec.end(), out);
    return (int)rec.size();
  } catch (const std::exception& e) {
    return fail(-2, e.what());
  }
}

int orc_fingerprint(const int* params, const u64* rec, u64* fp, u32* auxkey) {
  try {
    Params P = params_from(params);
    State s = decode(P, rec, nullptr);
    Fp f = fingerprint(P, s);
    *fp = f.fp;
    *auxkey = f.auxkey;
    return 0;
  } catch (const std::exception& e) {
    return fail(-2, e.what());
  }
}

int orc_invariants(const int* params, const u64* rec) {
  try {
    Params P = params_from(params);
    return check_invariants(P, decode(P, rec, nullptr));
  } catch (const std::exception& e) {
    return fail(-2, e.what());
  }
}

// re-encode a record through decode/encode (normalises the bag order); returns words
int orc_normalise(const int* params, const u64* rec, u64* out, int cap) {
  try {
    Params P = params_from(params);
    std::vector<u64> r;
    encode(P, decode(P, rec, nullptr), r);
    if ((int)r.size() > cap) return fail(-3, "buffer too small");
    std::copy(r.begin(), r.end(), out);
    return (int)r.size();
  } catch (const std::exception& e) {
    return fail(-2, e.what());
  }
}

// All successors of one packed state, in Next order.  meta has 5 u64 per successor:
// [action id, fingerprint, auxkey, record length in words, violated-invariant mask].
// Returns the number of successors, -1 on a TLC evaluation error, -2 on a representation error.
int orc_successors(const int* params, const u64* rec, u64* out_words, int cap_words, u64* meta, int cap_succ, int* words_used) {
  try {
    Params P = params_from(params);
    State s = decode(P, rec, nullptr);
    std::vector<Succ> succ;
    successors(P, s, succ);
    if ((int)succ.size() > cap_succ) return fail(-3, "successor buffer too small");
    std::vector<u64> w;
    for (size_t k = 0; k < succ.size(); k++) {
      size_t before = w.size();
      encode(P, succ[k].st, w);
      Fp f = fingerprint(P, succ[k].st);
      meta[5 * k + 0] = (u64)succ[k].action;
      meta[5 * k + 1] = f.fp;
      meta[5 * k + 2] = f.auxkey;
      meta[5 * k + 3] = w.size() - before;
      meta[5 * k + 4] = (u64)check_invariants(P, succ[k].st);
    }
    if ((int)w.size() > cap_words) return fail(-3, "word buffer too small");
    std::copy(w.begin(), w.end(), out_words);
    *words_used = (int)w.size();
    return (int)succ.size();
  } catch (const EvalError& e) {
    return fail(-1, e.what());
  } catch (const std::exception& e) {
    return fail(-2, e.what());
  }
}

// ---- BFS handle ---------------------------------------------------------------------------------
void* orc_bfs_create(const int* params) {
  try {
    Bfs* b = new Bfs();
    b->P = params_from(params);
    b->start();
    return b;
  } catch (const std::exception& e) {
    g_err = e.what();
    return nullptr;
  }
}
void orc_bfs_destroy(void* h) { delete (Bfs*)h; }

// info[0..11] = depth, n_new, generated(level), ties(level), deadlocks(level), total distinct, total generated,
//               viol_mask, viol_gid, error_code, max_bag, frontier words
long long orc_bfs_step(void* h, u64* info) {
  Bfs* b = (Bfs*)h;
  u64 nn = b->step();
  if (b->error_code) g_err = b->error;
  info[0] = b->depth;
  info[1] = nn;
  info[2] = b->generated;
  info[3] = b->ties;
  info[4] = b->deadlocks;
  info[5] = b->fp_of.size();
  info[6] = b->total_generated;
  info[7] = (u64)b->viol_mask;
  info[8] = b->viol_gid;
  info[9] = (u64)(long long)b->error_code;
  info[10] = b->max_bag;
  info[11] = b->fr_words.size();
  return (long long)nn;
}
double orc_bfs_level_seconds(void* h) { return ((Bfs*)h)->level_seconds; }

// sorted fingerprints of level `level` (1-based; Init = level 1); returns count
long long orc_bfs_level_fps(void* h, int level, u64* out, long long cap) {
  Bfs* b = (Bfs*)h;
  if (level < 1 || level > (int)b->level_base.size()) return -1;
  u64 lo = b->level_base[level - 1];
  u64 hi = (level < (int)b->level_base.size()) ? b->level_base[level] : b->fp_of.size();
  if ((long long)(hi - lo) > cap) return -(long long)(hi - lo);
  std::copy(b->fp_of.begin() + lo, b->fp_of.begin() + hi, out);
  std::sort(out, out + (hi - lo));
  return (long long)(hi - lo);
}

// the packed frontier (newest level): words + offsets (n+1 entries)
long long orc_bfs_frontier(void* h, u64* words, long long cap_words, u64* off, long long cap_states) {
  Bfs* b = (Bfs*)h;
  long long n = (long long)b->fr_off.size() - 1;
  if ((long long)b->fr_words.size() > cap_words || n + 1 > cap_states) return -1;
  std::copy(b->fr_words.begin(), b->fr_words.end(), words);
  std::copy(b->fr_off.begin(), b->fr_off.end(), off);
  return n;
}

// fingerprints along the parent chain Init -> gid; returns the chain length
int orc_bfs_trace_fps(void* h, u64 gid, u64* out, int cap) {
  Bfs* b = (Bfs*)h;
  std::vector<u64> chain;
  while (gid != ~0ull) {
    chain.push_back(b->fp_of[gid]);
    gid = b->parent_of[gid];
  }
  if ((int)chain.size() > cap) return -1;
  std::reverse(chain.begin(), chain.end());
  std::copy(chain.begin(), chain.end(), out);
  return (int)chain.size();
}

}  // extern "C"

#ifdef ORACLE_MAIN
// CLI: vsr_oracle R C n L [--max-depth D] [--max-seconds S] [--max-states N] [--no-symmetry] [--assume-commit-number]
// Prints one JSON object per completed level and a final summary line.
int main(int argc, char** argv) {
  if (argc < 5) {
    std::fprintf(stderr, "usage: %s R C nValues L [--max-depth D] [--max-seconds S] [--max-states N] [--no-symmetry] [--assume-commit-number] [--quiet]\n", argv[0]);
    return 2;
  }
  int params[8] = {std::atoi(argv[1]), std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]), 0, 0, 1, 1};
  int max_depth = 1 << 30;
  double max_seconds = 1e30;
  u64 max_states = ~0ull;
  bool quiet = false;
  for (int i = 5; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--max-depth" && i + 1 < argc) max_depth = std::atoi(argv[++i]);
    else if (a == "--max-seconds" && i + 1 < argc) max_seconds = std::atof(argv[++i]);
    else if (a == "--max-states" && i + 1 < argc) max_states = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "--no-symmetry") params[6] = 0;
    else if (a == "--assume-commit-number") params[5] = 1;
    else if (a == "--quiet") quiet = true;
    else { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
  }
  Bfs* b = (Bfs*)orc_bfs_create(params);
  if (!b) { std::fprintf(stderr, "error: %s\n", orc_last_error()); return 1; }
  double t0 = now_s();
  const char* why = "exhausted";
  while (true) {
    if (b->depth >= max_depth) { why = "max-depth"; break; }
    if (now_s() - t0 > max_seconds) { why = "max-seconds"; break; }
    if (b->fp_of.size() > max_states) { why = "max-states"; break; }
    u64 nn = b->step();
    if (b->error_code) { why = "error"; break; }
    if (!quiet && nn)
      std::printf("{\"level\": %d, \"new\": %llu, \"generated\": %llu, \"ties\": %llu, \"deadlocks\": %llu, \"distinct\": %zu, \"seconds\": %.3f}\n",
                  b->depth, (unsigned long long)nn, (unsigned long long)b->generated, (unsigned long long)b->ties,
                  (unsigned long long)b->deadlocks, b->fp_of.size(), b->level_seconds);
    if (b->viol_mask) { why = "violation"; break; }
    if (nn == 0) break;
  }
  double dt = now_s() - t0;
  std::printf("{\"summary\": true, \"stop\": \"%s\", \"depth\": %d, \"distinct\": %zu, \"generated\": %llu, \"seconds\": %.3f, \"states_per_s\": %.1f, \"max_bag\": %zu, \"viol_mask\": %d, \"error\": \"%s\", \"threads\": 1}\n",
              why, b->depth, b->fp_of.size(), (unsigned long long)b->total_generated, dt, b->fp_of.size() / (dt > 0 ? dt : 1e-9),
              b->max_bag, b->viol_mask, b->error.c_str());
  delete b;
  return 0;
}
#endif

// vsrmc — command-line front end over the C ABI (include/vsrmc.h), keeping TLC's surface for this model:
//     vsrmc -config VSR.cfg VSR.tla [-deadlock] [-maxDepth N] [-device D] [-tableLog2 N] [-frontierGiB G] [-noTLA] [-json]
// ≙ `java -cp tla2tools.jar tlc2.TLC -config VSR.cfg VSR.tla -deadlock` (README:20 of the reference).  Output follows TLC's
// wording (progress lines, "Invariant ... is violated", "State k: <Action>" + the state in TLC value syntax, final counts).
// `-deadlock` (= do NOT check deadlock) is the default here because the spec has terminal states; pass
// -checkDeadlock to stop at the first state without successors like stock TLC would.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
#include <unistd.h>

#include "../../include/vsrmc.h"

// invariant_mask bits (include/vsrmc.h): 0-1 in all three models, 2-3 in VSR.tla (NoLogDivergence only) and the analysis models, 4 only in VR_APP_STATE
static const char* const INVARIANT_NAMES[5] = {"AcknowledgedWriteNotLost", "AcknowledgedWritesExistOnMajority", "NoLogDivergence",
                                               "CommitNumberNeverHigherThanOpNumber", "NoAppStateDivergence"};

static void usage() {
  std::printf(
      "usage: vsrmc -config <file.cfg> <spec.tla> [options]\n"
      "  -config FILE      TLC configuration (grammar of VSR.cfg: CONSTANTS / INIT / NEXT / VIEW / SYMMETRY / INVARIANT)\n"
      "  -deadlock         do not check for deadlock (default)      -checkDeadlock   report the first terminal state\n"
      "  -maxDepth N       stop after N BFS levels (Init = level 1)\n"
      "  -device D         HIP device ordinal (default 0)\n"
      "  -gpus N           N > 1: run the sharded checker on N GPUs of this node (re-executes as\n"
      "                    python3 -m torch.distributed.run ... -m vsr_tlaplus_amd.sharded_cli with the other arguments;\n"
      "                    -tableLog2 / -frontierGiB are then PER RANK; see that module for -replicateBelow, -backend, -exactTies,\n"
      "                    -checkpoint PREFIX / -recover PREFIX (one file per rank), -probeAt N)\n"
      "  -tableLog2 N      seen-set slots = 2^N x 16 B; -frontierGiB G: size of each of the two record buffers (-frontierBGiB G: the second one).\n"
      "                    Default for both: sized from the free memory of the device (the largest power of two of slots within 30 %% of it, the\n"
      "                    rest in two equal record buffers).  Levels are stored while the next one is predicted to fit; beyond that the search goes\n"
      "                    on through the seen-set alone (Virtual(L) / Probe(L+1) lines).  A seen-set that fills up is re-hashed into twice the\n"
      "                    slots while the device has the memory; when it cannot grow and is 85 %% full the run ends \"incomplete at depth N\" (exit 4).\n"
      "  -simulate         random walks instead of BFS (TLC -simulate): -depth N (default 100) -walkers N (131072) -seed S -maxSeconds T\n"
      "  -validateTrace F  read a TLC trace (trace expression, or console \"State k:\" form) and check on the GPU that it is a\n"
      "                    behaviour of the model: Init, then one generated successor after the other; reports the invariants\n"
      "                    its last state violates\n"
      "  -hostFrontier     keep the two record buffers (-frontierGiB each) in pinned host memory, read / written over PCIe\n"
      "                    (state spaces whose frontier outgrows HBM; TLC's DiskStateQueue)\n"
      "  -hostFrontierMask M   the same per buffer: bit 0 = first buffer (levels 1, 3, ..), bit 1 = second (levels 2, 4, ..)\n"
      "  -probe2At N       when level N-1 is complete: level N as a VIRTUAL level (seen-set entries and invariants only) and level N+1\n"
      "                    as a PROBE level (nothing stored) — two levels beyond the last frontier that fits; the search ends there\n"
      "  -probe3At N       the same with TWO virtual levels (N, N+1) and level N+2 probed: three levels beyond the last stored one\n"
      "                    (level N-1 is expanded three times, level N twice; scratch buffers of a quarter of -frontierGiB)\n"
      "  -probeLast        when the next level does not fit the frontier buffers, still check its states' invariants without\n"
      "                    storing them (finds a violation one level beyond memory; the search ends there)\n"
      "  -coverage         print the successors generated per action of Next at the end (TLC's -coverage, totals only)\n"
      "  -dumpTrace tla FILE   write the counter-example as a TLA+ trace expression (TLC's -dumpTrace tla; the form of the reference's\n"
      "                    state_transfer_violation_trace.txt); -validateTrace reads it back\n"
      "  -dump FILE        write every distinct state in the text form of `tlc2.TLC -dump` (State k: + /\\ var = value conjuncts), level by\n"
      "                    level; for cross-checking small configurations against a real TLC run (refused beyond -dumpMax states, 1e6)\n"
      "  -checkpoint FILE  write a checkpoint between levels (stored ones and Virtual(L) ones alike), at most every -checkpointMinutes M\n"
      "                    (default 30; 0 = after every level)\n"
      "  -audit            second-hash audit (TLC: a rerun under another -fp N): when the search has ended, run it again to the same depth under\n"
      "                    another member of the fingerprint family and compare every level's new / generated / deadlock counts: a 64-bit\n"
      "                    collision that hid a state under one function would have to repeat under the other (exit 13 if they differ)\n"
      "  -recover FILE     continue the search a checkpoint stopped at (same constants; buffer sizes may differ)\n"
      "  -noTLA            do not read / hash-check the .tla file (only the cfg)\n"
      "  -json             one JSON object per level on stdout instead of TLC-style progress lines\n");
}

// -gpus N: hand the run to the multi-GPU front end (one process per GPU under torch.distributed.run)
static int exec_sharded(int argc, char** argv, int gpus_at) {
  std::string exe = argv[0];
  char real[4096];
  ssize_t n = readlink("/proc/self/exe", real, sizeof(real) - 1);
  if (n > 0) { real[n] = 0; exe = real; }
  std::string root = exe.substr(0, exe.find_last_of('/'));          // .../vsr-tlaplus_amd
  root = root.substr(0, root.find_last_of('/'));                    // repository root (holds the vsr_tlaplus_amd import shim)
  const char* old = getenv("PYTHONPATH");
  setenv("PYTHONPATH", old && *old ? (root + ":" + old).c_str() : root.c_str(), 1);
  const char* port = getenv("MASTER_PORT");
  std::vector<std::string> args = {"python3", "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", argv[gpus_at + 1],
                                   "--master-addr", "127.0.0.1", "--master-port", port ? port : "29500", "-m", "vsr_tlaplus_amd.sharded_cli"};
  for (int i = 1; i < argc; i++)
    if (i != gpus_at && i != gpus_at + 1) args.push_back(argv[i]);
  std::vector<char*> cargs;
  for (auto& a : args) cargs.push_back(const_cast<char*>(a.c_str()));
  cargs.push_back(nullptr);
  execvp("python3", cargs.data());
  std::perror("vsrmc: cannot start python3");
  return 1;
}

// -dumpTrace tla FILE: the counter-example as a TLA+ trace expression — the form of the reference's state_transfer_violation_trace.txt
// (a sequence of records, each with TLC's _TEAction field in front of the variables); source locations are not tracked by the
// lowering, so `location` says so for every step.  The product's reader (-validateTrace) takes the file back.
static bool write_trace_expression(const std::string& path, const vsrmc_model* m, const std::vector<uint64_t>& words,
                                   const std::vector<uint64_t>& off, const std::vector<int32_t>& acts, uint64_t n_states) {
  FILE* f = std::fopen(path.c_str(), "w");
  if (!f) return false;
  std::fprintf(f, "<<\n");
  for (uint64_t t = 0; t < n_states; t++) {
    int64_t need = 0;
    vsrmc_model_format_state(m, &words[off[t]], nullptr, 0, &need);
    std::string buf((size_t)need, '\0');
    vsrmc_model_format_state(m, &words[off[t]], &buf[0], need, &need);
    while (!buf.empty() && buf.back() == '\0') buf.pop_back();
    const size_t first_nl = buf.find('\n');                     // "[\nvar |-> value,\n...\n]": splice the _TEAction field after the bracket
    std::fprintf(f, "[\n _TEAction |-> [\n   position |-> %llu,\n   name |-> \"%s\",\n   location |-> \"Unknown location\"\n ],\n%s%s\n",
                 (unsigned long long)(t + 1), vsrmc_action_name(acts[t]), first_nl == std::string::npos ? "" : buf.c_str() + first_nl + 1,
                 t + 1 < n_states ? "," : "");
  }
  std::fprintf(f, ">>\n");
  return std::fclose(f) == 0;
}

int main(int argc, char** argv) {
  for (int i = 1; i + 1 < argc; i++)
    if (std::string(argv[i]) == "-gpus" && std::atoi(argv[i + 1]) > 1) return exec_sharded(argc, argv, i);
  std::string cfg, tla, trace_file, chk_file, recover_file, dump_file, dump_trace_file;
  unsigned long long dump_max = 1000000ull, dumped = 0;
  double chk_minutes = 30.0;
  bool check_deadlock = false, no_tla = false, json = false, simulate = false, host_frontier = false, probe_last = false, coverage = false, audit = false;
  int sim_depth = 100;
  unsigned sim_walkers = 1u << 17;
  unsigned long long sim_seed = 1;
  double sim_seconds = 60.0;
  int max_depth = 1 << 30, device = 0, table_log2 = 0, probe2_at = 0, probe3_at = 0, host_mask = 0;   // 0 = sized from the free device memory
  double frontier_gib = 0.0, frontier_b_gib = 0.0;
  for (int i = 1; i < argc; i++) {
    std::string a = argv[i];
    if (a == "--help" || a == "-h" || a == "-help") { usage(); return 0; }
    else if (a == "-config" && i + 1 < argc) cfg = argv[++i];
    else if (a == "-deadlock") check_deadlock = false;
    else if (a == "-checkDeadlock") check_deadlock = true;
    else if (a == "-maxDepth" && i + 1 < argc) max_depth = std::atoi(argv[++i]);
    else if (a == "-device" && i + 1 < argc) device = std::atoi(argv[++i]);
    else if (a == "-gpus" && i + 1 < argc) ++i;                   // 1: this process
    else if (a == "-tableLog2" && i + 1 < argc) table_log2 = std::atoi(argv[++i]);
    else if (a == "-frontierGiB" && i + 1 < argc) frontier_gib = std::atof(argv[++i]);
    else if (a == "-frontierBGiB" && i + 1 < argc) frontier_b_gib = std::atof(argv[++i]);
    else if (a == "-noTLA") no_tla = true;
    else if (a == "-hostFrontier") host_frontier = true;
    else if (a == "-probeLast") probe_last = true;
    else if (a == "-probe2At" && i + 1 < argc) probe2_at = std::atoi(argv[++i]);
    else if (a == "-probe3At" && i + 1 < argc) probe3_at = std::atoi(argv[++i]);
    else if (a == "-hostFrontierMask" && i + 1 < argc) host_mask = std::atoi(argv[++i]);
    else if (a == "-validateTrace" && i + 1 < argc) trace_file = argv[++i];
    else if (a == "-checkpoint" && i + 1 < argc) chk_file = argv[++i];
    else if (a == "-checkpointMinutes" && i + 1 < argc) chk_minutes = std::atof(argv[++i]);
    else if (a == "-recover" && i + 1 < argc) recover_file = argv[++i];
    else if (a == "-dump" && i + 1 < argc) dump_file = argv[++i];
    else if (a == "-dumpTrace" && i + 2 < argc && std::string(argv[i + 1]) == "tla") { dump_trace_file = argv[i + 2]; i += 2; }
    else if (a == "-dumpMax" && i + 1 < argc) dump_max = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "-simulate") simulate = true;
    else if (a == "-depth" && i + 1 < argc) sim_depth = std::atoi(argv[++i]);
    else if (a == "-walkers" && i + 1 < argc) sim_walkers = (unsigned)std::strtoul(argv[++i], nullptr, 10);
    else if (a == "-seed" && i + 1 < argc) sim_seed = std::strtoull(argv[++i], nullptr, 10);
    else if (a == "-maxSeconds" && i + 1 < argc) sim_seconds = std::atof(argv[++i]);
    else if (a == "-json") json = true;
    else if (a == "-coverage") coverage = true;
    else if (a == "-audit") audit = true;
    else if (a == "-workers" && i + 1 < argc) ++i;   // accepted for command-line compatibility; the GPU is the worker pool
    else if (!a.empty() && a[0] != '-') tla = a;
    else { std::fprintf(stderr, "vsrmc: unknown option %s\n", a.c_str()); usage(); return 2; }
  }
  if (cfg.empty()) { usage(); return 2; }
  vsrmc_model* m = nullptr;
  if (vsrmc_model_load(no_tla || tla.empty() ? nullptr : tla.c_str(), cfg.c_str(), &m) != 0) {
    std::fprintf(stderr, "Error: %s\n", vsrmc_last_error());
    return 1;
  }
  vsrmc_layout lay;
  vsrmc_model_info(m, &lay);
  if (lay.check_deadlock) check_deadlock = true;
  if (!trace_file.empty()) {   // import of a TLC trace: is it a behaviour of the lowered model?
    std::ifstream f(trace_file, std::ios::binary);
    if (!f) { std::fprintf(stderr, "Error: cannot read %s\n", trace_file.c_str()); return 1; }
    std::stringstream ss;
    ss << f.rdbuf();
    const std::string text = ss.str();
    uint64_t n = 0;
    if (vsrmc_model_parse_states(m, text.c_str(), nullptr, 0, nullptr, nullptr, 0, &n) != 0) {
      std::fprintf(stderr, "Error: %s\n", vsrmc_last_error());
      return 1;
    }
    std::vector<uint64_t> words((n + 1) * 320), off(n + 1);
    std::vector<int32_t> named(n + 1), acts(n + 1, -1);
    std::vector<uint32_t> ords(n + 1);
    int64_t bad = -1;
    int32_t inv = 0;
    if (vsrmc_model_parse_states(m, text.c_str(), words.data(), words.size(), off.data(), named.data(), n + 1, &n) != 0 ||
        vsrmc_model_check_trace(m, device, words.data(), off.data(), n, ords.data(), acts.data(), &bad, &inv) != 0) {
      std::fprintf(stderr, "Error: %s\n", vsrmc_last_error());
      return 1;
    }
    std::printf("%llu states read from %s.\n", (unsigned long long)n, trace_file.c_str());
    const uint64_t good = bad < 0 ? n : (uint64_t)bad;
    for (uint64_t t = 0; t < good; t++) {
      const char* mark = (named[t] >= 0 && named[t] != acts[t]) ? "   (the file names a different action for the same step)" : "";
      if (t == 0) std::printf("State 1: <Initial predicate>%s\n", mark);
      else std::printf("State %llu: <%s> ordinal %u%s\n", (unsigned long long)(t + 1), vsrmc_action_name(acts[t]), ords[t - 1], mark);
    }
    int code = 0;
    if (bad >= 0) {
      std::printf("Error: state %lld is not %s.\n", (long long)(bad + 1), bad == 0 ? "the initial state" : "a successor of its predecessor");
      code = 1;
    } else {
      std::printf("The trace is a behaviour of the model.\n");
      const char* const* names = INVARIANT_NAMES;
      for (int b = 0; b < 5; b++)
        if (inv & (1 << b)) { std::printf("Its last state violates invariant %s.\n", names[b]); code = 12; }
    }
    vsrmc_model_destroy(m);
    return code;
  }
  if (simulate) {   // ≙ tlc2.TLC -simulate -depth N
    vsrmc_sim_result r;
    if (vsrmc_simulate(m, device, sim_walkers, sim_depth, sim_seed, sim_seconds, &r) != 0) {
      std::fprintf(stderr, "Error: %s\n", vsrmc_last_error());
      return 1;
    }
    std::printf("Running Random Simulation with seed %llu: %u walkers on the GPU, depth %d.\n", sim_seed, sim_walkers, sim_depth);
    int code = 0;
    if (r.found == 1) {
      const char* const* names = INVARIANT_NAMES;
      for (int b = 0; b < 5; b++)
        if (r.viol_mask & (1 << b)) std::printf("Error: Invariant %s is violated.\n", names[b]);
      std::printf("Error: The behavior up to this point is:\n");
      uint64_t cap_w = (uint64_t)(r.viol_steps + 3) * 256, n_states = 0;
      std::vector<uint64_t> words(cap_w), off(r.viol_steps + 3);
      std::vector<int32_t> acts(r.viol_steps + 3);
      if (vsrmc_model_replay(m, device, r.ords, r.viol_steps, words.data(), cap_w, off.data(), acts.data(), off.size(), &n_states) != 0) {
        std::printf("Error: %s\n", vsrmc_last_error());
        return 1;
      }
      for (uint64_t t = 0; t < n_states; t++) {
        int64_t need = 0;
        vsrmc_model_format_state(m, &words[off[t]], nullptr, 0, &need);
        std::string buf((size_t)need, '\0');
        vsrmc_model_format_state(m, &words[off[t]], &buf[0], need, &need);
        std::printf("State %llu: <%s>\n%s\n\n", (unsigned long long)(t + 1), vsrmc_action_name(acts[t]), buf.c_str());
      }
      code = 12;
    } else if (r.found == 2) {
      std::printf("Error: a walk raised device error %d after %d steps.\n", r.viol_mask, r.viol_steps);
      code = 1;
    } else {
      std::printf("Simulation stopped after %.1f s without a violation.\n", r.seconds);
    }
    std::printf("%llu states checked in %llu walks, %.3f s (%.3g steps/s).\n", (unsigned long long)r.steps, (unsigned long long)r.walks,
                r.seconds, r.seconds > 0 ? (double)r.steps / r.seconds : 0.0);
    vsrmc_model_destroy(m);
    return code;
  }
  vsrmc_options o;
  vsrmc_options_default(&o);
  o.device = device;
  o.table_log2 = table_log2;
  o.host_frontier = host_frontier ? 3 : (host_mask & 3);
  o.frontier_words = (uint64_t)(frontier_gib * 1024.0 * 1024.0 * 1024.0 / 8.0);
  o.frontier_words_b = (uint64_t)(frontier_b_gib * 1024.0 * 1024.0 * 1024.0 / 8.0);   // 0 = like the first
  o.frontier_states = o.frontier_words / 24;                   // (0 with -frontierGiB 0: derived with the words)
  o.pending_entries = (uint64_t)1 << 20;   // single-pass levels keep no pending list (the buffer only collects violators of probe levels)
  // one entry per state plus the unused tails of the per-block index chunks (<= 4096 per block per level, 510 levels at most)
  o.trace_entries = ((uint64_t)1 << table_log2) / 2 + ((uint64_t)1 << 28);   // + chunk tails: <= 1024 blocks x 8192 indices x ~30 large levels
  vsrmc_checker* c = nullptr;
  if ((recover_file.empty() ? vsrmc_checker_create(m, &o, &c) : vsrmc_checker_load(m, &o, recover_file.c_str(), &c)) != 0) {
    std::fprintf(stderr, "Error: %s\n", vsrmc_last_error());
    return 1;
  }
  static const char* const MODULES[3] = {"VSR.tla", "VR_STATE_TRANSFER.tla", "VR_APP_STATE.tla"};
  std::printf("vsrmc: %s lowered: ReplicaCount=%d ClientCount=%d |Values|=%d StartViewOnTimerLimit=%d, %d permutation(s), "
              "invariant mask %d\n", MODULES[lay.module >= 0 && lay.module < 3 ? lay.module : 0], lay.replica_count, lay.client_count, lay.value_count, lay.start_view_on_timer_limit,
              lay.permutations, lay.invariant_mask);
  auto t0 = std::chrono::steady_clock::now();
  auto t_chk = t0;
  vsrmc_level_info info;
  vsrmc_checker_status(c, &info);
  if (recover_file.empty()) std::printf("Finished computing initial states: 1 distinct state generated.\n");
  else std::printf("Recovered from %s: level %d, %llu distinct states found, %llu states left on queue.\n", recover_file.c_str(), info.level,
                   (unsigned long long)info.distinct, (unsigned long long)info.n_new);
  int rc = 0;
  FILE* dump = nullptr;
  if (!dump_file.empty()) {
    dump = std::fopen(dump_file.c_str(), "w");
    if (!dump) { std::fprintf(stderr, "Error: cannot write %s\n", dump_file.c_str()); return 1; }
  }
  // the newest level in TLC's -dump text form: "State k:" + one "/\ var = value" conjunct per variable
  auto dump_level = [&](uint64_t n_states) -> bool {
    if (!dump || n_states == 0) return true;
    if (dumped + n_states > dump_max) {
      std::fprintf(stderr, "Error: -dump stops at %llu states (-dumpMax)\n", dump_max);
      return false;
    }
    std::vector<uint64_t> words(n_states * (uint64_t)lay.max_record_words), off(n_states + 1);
    uint64_t n = 0;
    if (vsrmc_checker_frontier(c, words.data(), words.size(), off.data(), off.size(), &n) != 0) return false;
    for (uint64_t t = 0; t < n; t++) {
      int64_t need = 0;
      vsrmc_model_format_state(m, &words[off[t]], nullptr, 0, &need);
      std::string buf((size_t)need, '\0');
      vsrmc_model_format_state(m, &words[off[t]], &buf[0], need, &need);
      std::fprintf(dump, "State %llu:\n", ++dumped);
      size_t pos = 0;
      while (pos < buf.size()) {                              // "[\nvar |-> value,\n...\n]" -> conjuncts
        size_t eol = buf.find('\n', pos);
        if (eol == std::string::npos) eol = buf.size();
        std::string line = buf.substr(pos, eol - pos);
        pos = eol + 1;
        size_t arrow = line.find(" |-> ");
        if (arrow == std::string::npos) continue;             // the brackets
        if (!line.empty() && line.back() == ',') line.pop_back();
        std::fprintf(dump, "/\\ %s = %s\n", line.substr(0, arrow).c_str(), line.substr(arrow + 5).c_str());
      }
      std::fprintf(dump, "\n");
    }
    return true;
  };
  if (dump && recover_file.empty() && !dump_level(1)) return 1;
  bool violated = false, deadlocked = false, probed_violation = false, incomplete = false;
  unsigned long long cov[16] = {0};
  uint64_t viol_level = 0, viol_index = 0;
  int depth = info.level + info.reserved0;                        // (a recovered deep search: the levels beyond the stored one)
  struct Row { int level; unsigned long long n_new, generated, deadlocks; };
  std::vector<Row> rows;                                          // what -audit compares
  auto checkpoint_now = [&](int level) {
    if (chk_file.empty() || std::chrono::duration<double>(std::chrono::steady_clock::now() - t_chk).count() < 60.0 * chk_minutes) return;
    if (vsrmc_checker_save(c, chk_file.c_str()) != 0) std::printf("Warning: %s\n", vsrmc_last_error());
    else std::printf("Checkpointing of run %s completed (level %d).\n", chk_file.c_str(), level);
    t_chk = std::chrono::steady_clock::now();
  };
  while (depth < max_depth) {
    if ((probe2_at > 0 && depth + 1 == probe2_at) || (probe3_at > 0 && depth + 1 == probe3_at)) {
      const int nv = probe3_at > 0 && depth + 1 == probe3_at ? 2 : 1;        // virtual levels before the probed one
      vsrmc_level_info li[3];
      rc = nv == 2 ? vsrmc_checker_probe3(c, &li[0], &li[1], &li[2]) : vsrmc_checker_probe2(c, &li[0], &li[1]);
      if (rc != 0) break;
      double dtp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      for (int k = 0; k <= nv && !probed_violation; k++) {
        const vsrmc_level_info& x = li[k];
        if (x.level == 0) break;
        if (k < nv) {
          std::printf("Virtual(%d): %llu states generated, %llu distinct states found, %llu states in the level (not stored). (%.2f s)\n", x.level,
                      (unsigned long long)x.total_generated, (unsigned long long)x.distinct, (unsigned long long)x.n_new, dtp);
          info.distinct = x.distinct;
          info.n_new = x.n_new;
        } else {
          std::printf("Probe(%d): %llu states generated from %llu states, %llu violating successors seen. (%.2f s)\n", x.level,
                      (unsigned long long)x.generated, (unsigned long long)x.frontier, (unsigned long long)x.pending, dtp);
        }
        info.total_generated = x.total_generated;
        depth = x.level;
        if (x.viol_mask) {
          probed_violation = true;
          info.viol_mask = x.viol_mask;
          viol_level = (uint64_t)x.level;
        } else if (k == nv) {
          std::printf("No violation up to level %d; the search is incomplete beyond it.\n", x.level);
        }
      }
      break;
    }
    // the automatic level scheme: an ordinary level while the next one is predicted to fit the record buffers, else one more level through
    // the seen-set alone (vsrmc_checker_deepen: inserted, counted and checked, its records regenerated when needed) and a probe of the one after
    int32_t what = 0, room = 0;
    vsrmc_level_info probed;
    rc = vsrmc_checker_room(c, &room);                            // grows the seen-set while the device has the memory
    if (rc != 0) break;
    if (room == 1 && !json) std::printf("The seen-set was re-hashed into twice the slots.\n");
    if (room == 2) { incomplete = true; break; }
    rc = vsrmc_checker_advance(c, &info, &probed, &what);
    if (rc != 0) break;
    if (what == 3) {                                              // no new level: the deep search was re-based (the levels shrink again)
      if (!json) std::printf("Re-based(%d): %llu states regenerated into the record buffers (%llu launches, %.2f s); stored levels from here on.\n", info.level,
                             (unsigned long long)info.n_new, (unsigned long long)info.pending, info.seconds);
      continue;
    }
    if (what == 2) {
      for (int a2 = 1; a2 < 16; a2++) cov[a2] += info.act_generated[a2];
      const double dtp = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dump) { std::printf("Level %d is not stored (it does not fit the record buffers): -dump ends at level %d.\n", info.level, depth); std::fclose(dump); dump = nullptr; }
      if (info.n_new == 0) break;
      depth = info.level;
      rows.push_back(Row{info.level, (unsigned long long)info.n_new, (unsigned long long)info.generated, (unsigned long long)info.deadlocks});
      if (json)
        std::printf("{\"level\": %d, \"stored\": false, \"generated\": %llu, \"new\": %llu, \"distinct\": %llu, \"deadlocks\": %llu, \"launches\": %llu, \"seconds\": %.4f}\n",
                    info.level, (unsigned long long)info.generated, (unsigned long long)info.n_new, (unsigned long long)info.distinct,
                    (unsigned long long)info.deadlocks, (unsigned long long)info.pending, dtp);
      else
      std::printf("Virtual(%d): %llu states generated, %llu distinct states found, %llu states in the level (not stored; %llu launches). (%.2f s)\n", info.level,
                  (unsigned long long)info.total_generated, (unsigned long long)info.distinct, (unsigned long long)info.n_new, (unsigned long long)info.pending, dtp);
      if (info.viol_mask) { probed_violation = true; viol_level = (uint64_t)info.level; break; }
      if (probed.level) {
        if (json)
          std::printf("{\"probed_level\": %d, \"generated\": %llu, \"from\": %llu, \"violating_successors\": %llu, \"seconds\": %.4f}\n", probed.level,
                      (unsigned long long)probed.generated, (unsigned long long)probed.frontier, (unsigned long long)probed.pending, dtp);
        else
        std::printf("Probe(%d): %llu states generated from %llu states, %llu violating successors seen. (%.2f s)\n", probed.level,
                    (unsigned long long)probed.generated, (unsigned long long)probed.frontier, (unsigned long long)probed.pending, dtp);
        if (probed.viol_mask) {
          probed_violation = true;
          info.viol_mask = probed.viol_mask;
          info.total_generated = probed.total_generated;
          viol_level = (uint64_t)probed.level;
          break;
        }
      }
      checkpoint_now(info.level);
      continue;
    }
    if (rc != 0) break;
    for (int a2 = 1; a2 < 16; a2++) cov[a2] += info.act_generated[a2];
    if (info.n_new) depth = info.level;
    if (info.n_new) rows.push_back(Row{info.level, (unsigned long long)info.n_new, (unsigned long long)info.generated, (unsigned long long)info.deadlocks});
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (json)
      std::printf("{\"level\": %d, \"generated\": %llu, \"new\": %llu, \"distinct\": %llu, \"deadlocks\": %llu, \"seconds\": %.4f}\n",
                  info.level, (unsigned long long)info.generated, (unsigned long long)info.n_new, (unsigned long long)info.distinct,
                  (unsigned long long)info.deadlocks, dt);
    else if (info.n_new)
      std::printf("Progress(%d): %llu states generated, %llu distinct states found, %llu states left on queue. (%.2f s)\n", info.level,
                  (unsigned long long)info.total_generated, (unsigned long long)info.distinct, (unsigned long long)info.n_new, dt);
    if (!dump_level(info.n_new)) { rc = -1; break; }
    if (info.viol_mask) { violated = true; viol_level = (uint64_t)info.level; viol_index = info.viol_index; break; }
    if (check_deadlock && info.deadlocks) { deadlocked = true; break; }
    if (info.n_new == 0) break;
    checkpoint_now(info.level);
  }
  double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  int exit_code = 0;
  if (rc != 0) {
    std::printf("Error: %s\n", vsrmc_last_error());
    exit_code = rc == VSRMC_E_EVAL ? 12 : 1;
  } else if (violated || probed_violation) {
    const char* const* names = INVARIANT_NAMES;
    for (int b = 0; b < 5; b++)
      if (info.viol_mask & (1 << b)) std::printf("Error: Invariant %s is violated.\n", names[b]);
    std::printf("Error: The behavior up to this point is:\n");
    uint64_t cap_w = (viol_level + 2) * (uint64_t)lay.max_record_words, n_states = 0;
    std::vector<uint64_t> words(cap_w), off(viol_level + 2);
    std::vector<int32_t> acts(viol_level + 2);
    if ((probed_violation ? vsrmc_checker_probe_trace(c, words.data(), cap_w, off.data(), acts.data(), off.size(), &n_states)
                          : vsrmc_checker_trace(c, (int32_t)viol_level, viol_index, words.data(), cap_w, off.data(), acts.data(), off.size(),
                                                &n_states)) != 0) {
      std::printf("Error: %s\n", vsrmc_last_error());
      exit_code = 1;
    } else {
      for (uint64_t t = 0; t < n_states; t++) {
        int64_t need = 0;
        vsrmc_model_format_state(m, &words[off[t]], nullptr, 0, &need);
        std::string buf((size_t)need, '\0');
        vsrmc_model_format_state(m, &words[off[t]], &buf[0], need, &need);
        std::printf("State %llu: <%s>\n%s\n\n", (unsigned long long)(t + 1), vsrmc_action_name(acts[t]), buf.c_str());
      }
      if (!dump_trace_file.empty()) {
        if (write_trace_expression(dump_trace_file, m, words, off, acts, n_states))
          std::printf("The counter-example was written to %s (TLA+ trace expression).\n", dump_trace_file.c_str());
        else
          std::printf("Warning: cannot write %s\n", dump_trace_file.c_str());
      }
      exit_code = 12;   // TLC's exit code for a safety violation
    }
  } else if (deadlocked) {
    std::printf("Error: Deadlock reached (%llu state(s) of level %d have no successor).\n", (unsigned long long)info.deadlocks, info.level - 1);
    exit_code = 11;
  } else if (incomplete) {
    std::printf("Model checking INCOMPLETE at depth %d: the seen-set is more than 85 %% full and the device has no memory for a larger one "
                "(no error has been found up to that depth).\n", depth);
    exit_code = 4;
  } else if (info.n_new == 0) {
    std::printf("Model checking completed. No error has been found.\n");
  }
  if (coverage) {   // ≙ tlc2.TLC -coverage, per action of Next: successors generated in the stored levels (probed / virtual levels not included)
    std::printf("The coverage statistics (successors generated per action):\n");
    for (int a2 = 1; a2 < 16; a2++) std::printf("  %s: %llu\n", vsrmc_action_name(a2), (unsigned long long)cov[a2]);
  }
  {  // TLC prints the same estimate: every generated state that was judged "seen" could be a 64-bit fingerprint collision
    const double n = (double)info.distinct, g = (double)info.total_generated;
    std::printf("The probability of a fingerprint collision hiding a state is estimated (optimistically) at %.1e.\n",
                n * (g > n ? g - n : 0.0) / 18446744073709551616.0);
  }
  std::printf("%llu states generated, %llu distinct states found, %llu states left on queue.\n", (unsigned long long)info.total_generated,
              (unsigned long long)info.distinct, (unsigned long long)(info.n_new));
  std::printf("The depth of the complete state graph search is %d.\nFinished in %.3f s (%.3g distinct states/s).\n", depth, dt,
              dt > 0 ? (double)info.distinct / dt : 0.0);
  if (dump) {
    std::fclose(dump);
    std::printf("%llu states dumped to %s.\n", dumped, dump_file.c_str());
  }
  if (audit && rc == 0 && recover_file.empty() && !rows.empty()) {
    // the second-hash audit: the same search under another member of the fingerprint family, to the depth the first one reached; only the counts are compared
    vsrmc_checker_destroy(c);
    c = nullptr;
    const uint64_t seed2 = 0x5EED5EED5EED5EEDull;
    size_t k = 0, first_diff = (size_t)-1;
    if (vsrmc_model_set_fp_seed(m, seed2) != 0 || vsrmc_checker_create(m, &o, &c) != 0) {
      std::printf("Error: audit run: %s\n", vsrmc_last_error());
      exit_code = exit_code ? exit_code : 1;
    } else {
      const int last = rows.back().level;
      int rc2 = 0;
      while (k < rows.size()) {
        int32_t what = 0, room = 0;
        vsrmc_level_info a, b;
        if ((rc2 = vsrmc_checker_room(c, &room)) != 0 || room == 2) break;
        if ((rc2 = vsrmc_checker_advance(c, &a, &b, &what)) != 0 || a.n_new == 0) break;
        if (what == 3) continue;
        const Row& r = rows[k];
        if (first_diff == (size_t)-1 && (a.level != r.level || a.n_new != r.n_new || a.generated != r.generated || a.deadlocks != r.deadlocks)) first_diff = k;
        k++;
        if (a.level >= last) break;
      }
      if (rc2 != 0) { std::printf("Error: audit run: %s\n", vsrmc_last_error()); exit_code = exit_code ? exit_code : 1; }
      else if (first_diff != (size_t)-1 || k != rows.size()) {
        const size_t d = first_diff != (size_t)-1 ? first_diff : k;
        std::printf("Audit: the per-level counts under fingerprint seed %016llx DIFFER from level %d on: a 64-bit fingerprint collision hid a state "
                    "under one of the two functions (TLC: rerun with another -fp).\n", (unsigned long long)seed2, d < rows.size() ? rows[d].level : last);
        exit_code = exit_code ? exit_code : 13;
      } else {
        std::printf("Audit: %zu levels, every new / generated / deadlock count equal under fingerprint seed %016llx.\n", rows.size(), (unsigned long long)seed2);
      }
    }
  }
  if (c) vsrmc_checker_destroy(c);
  vsrmc_model_destroy(m);
  return exit_code;
}

// oracle/vrst_oracle.hpp — CPU ORACLE for the SECOND model (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A literal C++17 restatement of the next-state relation of
//   /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.tla   (cited as VRST.tla:NNN)
// under its shipped configuration
//   /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.cfg   (VRST.cfg:NN):
// VIEW view, no SYMMETRY, INVARIANT AcknowledgedWritesExistOnMajority / NoLogDivergence / CommitNumberNeverHigherThanOpNumber,
// NoProgressChangeLimit = 0 (the NoProgressChange action, VRST.tla:757-767, is then dead and no_progress stays FALSE).
// SURVEY.md §8(f) rank 2.  Same shape as vsr_oracle.hpp (unpacked structs, sorted bag, full recomputation), same driver
// (vsr_oracle_bfs.cpp / vsr_oracle_mt.cpp compiled against this header), shares no code with the HIP path.
//
// Differences from VSR.tla that matter for the lowering: no clients and no client table; a log entry is [operation |-> v] only
// (VRST.tla:100-101); there are no rep_svc_recv / rep_dvc_recv variables — received StartViewChange / DoViewChange messages are
// counted IN THE BAG as keys with delivery count 0 (VRST.tla:603-607, 664-668); a third status StateTransfer (VRST.tla:55);
// GetState goes to AnyDest (VRST.tla:477-481) and any replica but the sender may take it (VRST.tla:200-205).
//
// PARITY STATUS: "parity unpinned" against TLC (no JVM here; the reference ships no golden vector for this model: its cfg
// comment only promises "no violation").  Pinned by: an independent Python restatement (oracle/pyoracle2.py) on whole small
// state spaces.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vrst_oracle {

typedef uint64_t u64;
typedef uint32_t u32;

struct Params {
  int R = 3;               // ReplicaCount                  VRST.cfg:4
  int C = 0;               // (no clients in this model; kept so that the drivers' "R C n L" command lines stay uniform)
  int n = 2;               // Cardinality(Values)           VRST.cfg:5
  int L = 2;               // StartViewOnTimerLimit         VRST.cfg:6
  int no_progress_limit = 0;   // NoProgressChangeLimit     VRST.cfg:7 (must be 0)
  bool symmetry = false;   // VRST.cfg:27 keeps SYMMETRY commented out (must be false)
  int invariant_mask = 14; // bit0 AcknowledgedWriteNotLost (VRST.tla:832-837), bit1 AcknowledgedWritesExistOnMajority (:819-825),
                           // bit2 NoLogDivergence (:806-812), bit3 CommitNumberNeverHigherThanOpNumber (:847-849); cfg:35-38 = 14
};
Params params_from_array(const int* p);   // {R, C, n, L, no_progress_limit, -, symmetry, invariant_mask}

enum { Normal = 0, ViewChange = 1, StateTransfer = 2 };                    // VRST.tla:53-55
enum { T_SVC = 1, T_PREPARE = 2, T_PREPAREOK = 3, T_DVC = 4, T_SV = 5, T_GETSTATE = 6, T_NEWSTATE = 7 };   // VRST.tla:56-63
enum { AnyDest = 7 };                                                     // VRST.tla:66 (a model value; 7 in the packed dest field)

// Action ids in `Next` order (VRST.tla:779-799); NoProgressChange (16) is dead at NoProgressChangeLimit = 0
enum { A_TimerSendSVC = 1, A_ReceiveHigherSVC, A_ReceiveMatchingSVC, A_SendDVC, A_ReceiveHigherDVC,
       A_ReceiveMatchingDVC, A_SendSV, A_ReceiveSV, A_ReceiveClientRequest, A_ReceivePrepareMsg,
       A_ReceivePrepareOkMsg, A_ExecuteOp, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState };
extern const char* const ACTION_NAMES[16];
const int FP_VERSION = 2;

struct EvalError : std::runtime_error { explicit EvalError(const std::string& s) : std::runtime_error(s) {} };
struct RepError : std::runtime_error { explicit RepError(const std::string& s) : std::runtime_error(s) {} };

// A function lo..hi -> value index (a log entry is [operation |-> v]); a sequence when lo == 1
struct Log {
  int lo = 1, hi = 0;
  int v[5] = {0, 0, 0, 0, 0};          // indexed by absolute op number 1..4
  int len() const { return hi >= lo ? hi - lo + 1 : 0; }
  bool operator==(const Log& o) const {
    if (len() != o.len()) return false;
    if (len() == 0) return true;
    if (lo != o.lo) return false;
    for (int i = lo; i <= hi; i++) if (v[i] != o.v[i]) return false;
    return true;
  }
  bool operator<(const Log& o) const {
    if (len() != o.len()) return len() < o.len();
    if (len() == 0) return false;
    if (lo != o.lo) return lo < o.lo;
    for (int i = lo; i <= hi; i++) if (v[i] != o.v[i]) return v[i] < o.v[i];
    return false;
  }
};

struct Msg {   // union of the message record types VRST.tla:103-161 (+ GetState.op_number, set at :479); unused fields 0 / empty
  int type = 0, view = 0, dest = 0, source = 0, op = 0, commit = 0, lnv = 0, first_op = 0;
  int entry = -1;    // PrepareMsg.message.operation (value index), -1 = none
  Log log;           // DVC / SV / NewState .log
  bool operator==(const Msg& o) const {
    return type == o.type && view == o.view && dest == o.dest && source == o.source && op == o.op && commit == o.commit &&
           lnv == o.lnv && first_op == o.first_op && entry == o.entry && log == o.log;
  }
  bool operator<(const Msg& o) const {   // any total order consistent with ==
    if (type != o.type) return type < o.type;
    if (view != o.view) return view < o.view;
    if (dest != o.dest) return dest < o.dest;
    if (source != o.source) return source < o.source;
    if (op != o.op) return op < o.op;
    if (commit != o.commit) return commit < o.commit;
    if (lnv != o.lnv) return lnv < o.lnv;
    if (first_op != o.first_op) return first_op < o.first_op;
    if (entry != o.entry) return entry < o.entry;
    return log < o.log;
  }
};

struct Replica {                      // one column of the rep_* variables (VRST.tla:72-81) and of no_progress (:84)
  int status = Normal, view = 1, op = 0, commit = 0, lnv = 1;
  bool sent_dvc = false, sent_sv = false, no_progress = false;
  Log log;
  int peer_op[6] = {0, 0, 0, 0, 0, 0};
};

struct State {
  Replica rep[6];                               // indexed by replica id 1..R
  std::vector<std::pair<Msg, int>> messages;    // the bag, sorted by Msg; zero-count keys stay (they ARE the "received" records)
  int aux_svc = 0;
  int acked[4] = {0, 0, 0, 0};                  // 0 = not in DOMAIN, 1 = FALSE, 2 = TRUE
  int no_progress_ctr = 0;
};

struct Succ { int action; State st; };

State init_state(const Params& P);                                         // VRST.tla:267-283
void successors(const Params& P, const State& s, std::vector<Succ>& out);  // VRST.tla:779-799
int check_invariants(const Params& P, const State& s);                     // mask of VIOLATED invariants

int words_per_replica(const Params& P);                                    // 1
int fixed_words(const Params& P);                                          // 1 + R
void encode(const Params& P, const State& s, std::vector<u64>& out);
State decode(const Params& P, const u64* rec, int* nwords);

struct Fp { u64 fp; u32 auxkey; int argmin; };
Fp fingerprint(const Params& P, const State& s);                           // VIEW view, VRST.tla:96 / VRST.cfg:23
u64 fmix64(u64 x);
void set_fp_seed(u64 seed);   // second-hash audit: xor-ed into every salt (process-global; 0 = the fixtures' function)
u64 fp_seed();
// the same function under an explicit seed (the collision hunt of vsr_oracle_lean.cpp compares two members of the family in one run)
#define ORACLE_HAS_SEEDED_FP 1
Fp fingerprint_with_seed(const Params& P, const State& s, u64 seed);

}  // namespace vrst_oracle

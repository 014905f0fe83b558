// oracle/vras_oracle.hpp — CPU ORACLE for the THIRD model (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A literal C++17 restatement of the next-state relation of
//   /root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.tla   (cited as VRAS.tla:NNN)
// under its shipped configuration
//   /root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.cfg   (VRAS.cfg:NN):
// VIEW view, no SYMMETRY, INVARIANT AcknowledgedWritesExistOnMajority / NoLogDivergence / NoAppStateDivergence /
// CommitNumberNeverHigherThanOpNumber, NoProgressChangeLimit = 0 (NoProgressChange, VRAS.tla:797-807, is then dead).
// SURVEY.md §8(f) rank 2 ("then 04-application-state").  Same shape as vrst_oracle.hpp (unpacked structs, sorted bag, full
// recomputation), same drivers (vsr_oracle_bfs.cpp / vsr_oracle_mt.cpp compiled against this header), shares no code with the
// HIP path.
//
// What VR_APP_STATE adds to VR_STATE_TRANSFER: rep_app_state (the executed operations, VRAS.tla:74) filled by MaybeExecuteOps
// (:277-283) wherever the commit number rises — which also makes the commit number monotonic; rep_recv_dvc (:82), an explicit
// set of the DoViewChange messages a replica counts (the second model counted bag keys with delivery count 0); a guard more on
// ReceiveMatchingSVC (:602); the invariant NoAppStateDivergence (:852-858).  rep_rec_number / rep_rec_recv / aux_restart are
// declared (:83-84, :91) and never written.
//
// PARITY STATUS: "parity unpinned" against TLC (no JVM here; the reference ships no golden vector for this model).  Pinned by
// an independent Python restatement (oracle/pyoracle3.py) on whole small state spaces.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vras_oracle {

typedef uint64_t u64;
typedef uint32_t u32;

struct Params {
  int R = 3;               // ReplicaCount                  VRAS.cfg:4
  int C = 0;               // (no clients in this model; kept so that the drivers' "R C n L" command lines stay uniform)
  int n = 2;               // Cardinality(Values)           VRAS.cfg:5
  int L = 2;               // StartViewOnTimerLimit         VRAS.cfg:6
  int no_progress_limit = 0;   // NoProgressChangeLimit     VRAS.cfg:7 (must be 0)
  bool symmetry = false;   // VRAS.cfg:28 keeps SYMMETRY commented out (must be false)
  int invariant_mask = 30; // bit0 AcknowledgedWriteNotLost (VRAS.tla:877-882), bit1 AcknowledgedWritesExistOnMajority (:865-871),
                           // bit2 NoLogDivergence (:840-845), bit3 CommitNumberNeverHigherThanOpNumber (:892-894),
                           // bit4 NoAppStateDivergence (:852-858); cfg:37-40 = 30
};
Params params_from_array(const int* p);   // {R, C, n, L, no_progress_limit, -, symmetry, invariant_mask}

enum { Normal = 0, ViewChange = 1, StateTransfer = 2 };                    // VRAS.tla:51-53
enum { T_SVC = 1, T_PREPARE = 2, T_PREPAREOK = 3, T_DVC = 4, T_SV = 5, T_GETSTATE = 6, T_NEWSTATE = 7 };   // VRAS.tla:56-62
enum { AnyDest = 7 };                                                     // VRAS.tla:65 (a model value; 7 in the packed dest field)

// Action ids in `Next` order (VRAS.tla:811-831); NoProgressChange (16) is dead at NoProgressChangeLimit = 0
enum { A_TimerSendSVC = 1, A_ReceiveHigherSVC, A_ReceiveMatchingSVC, A_SendDVC, A_ReceiveHigherDVC,
       A_ReceiveMatchingDVC, A_SendSV, A_ReceiveSV, A_ReceiveClientRequest, A_ReceivePrepareMsg,
       A_ReceivePrepareOkMsg, A_PrimaryExecuteOp, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState };
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

struct Msg {   // union of the message record types VRAS.tla:112-167 (+ GetState.op_number, set at :471); unused fields 0 / empty
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

struct Replica {                      // one column of the rep_* variables (VRAS.tla:72-84) and of no_progress (:87)
  int status = Normal, view = 1, op = 0, commit = 0, lnv = 1;
  bool sent_dvc = false, sent_sv = false, no_progress = false;
  Log log;
  std::vector<int> app_state;         // rep_app_state[r]: the executed operations (value indices), a sequence      :74
  std::vector<Msg> recv_dvc;          // rep_recv_dvc[r]: a SET of DoViewChange records, kept sorted and duplicate-free  :82
  int peer_op[6] = {0, 0, 0, 0, 0, 0};
};

struct State {
  Replica rep[6];                               // indexed by replica id 1..R
  std::vector<std::pair<Msg, int>> messages;    // the bag, sorted by Msg; zero-count keys stay
  int aux_svc = 0;
  int acked[4] = {0, 0, 0, 0};                  // 0 = not in DOMAIN, 1 = FALSE, 2 = TRUE
  int no_progress_ctr = 0;
};

struct Succ { int action; State st; };

State init_state(const Params& P);                                         // VRAS.tla:292-315
void successors(const Params& P, const State& s, std::vector<Succ>& out);  // VRAS.tla:811-831
int check_invariants(const Params& P, const State& s);                     // mask of VIOLATED invariants

int words_per_replica(const Params& P);                                    // 2
int fixed_words(const Params& P);                                          // 1 + 2 R
void encode(const Params& P, const State& s, std::vector<u64>& out);
State decode(const Params& P, const u64* rec, int* nwords);

struct Fp { u64 fp; u32 auxkey; int argmin; };
Fp fingerprint(const Params& P, const State& s);                           // VIEW view, VRAS.tla:102 / VRAS.cfg:24
u64 fmix64(u64 x);
void set_fp_seed(u64 seed);   // second-hash audit: xor-ed into every salt (process-global; 0 = the fixtures' function)
u64 fp_seed();

}  // namespace vras_oracle

// oracle/vsr_oracle.hpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// A literal C++17 restatement of the next-state relation of
//   /root/reference/vsr-revisited/paper/VSR.tla   (cited below as VSR.tla:NNN)
// under the checker configuration grammar of
//   /root/reference/vsr-revisited/paper/VSR.cfg   (VSR.cfg:NN),
// i.e. what TLC (tlc2.tool.Worker / Tool.getNextStates / TLCState.fingerPrint / FPSet) computes for
// this one model.  It works on *unpacked* structs (records, sorted sets, a sorted bag) and full
// recomputation everywhere, so that it shares no algorithmic code with the HIP path it checks
// (vsr_tlaplus_amd/csrc/*), which works on the packed record with incremental hashing.
//
// PARITY STATUS: "parity unpinned" against TLC itself — TLC (Java) is not in /root/reference, no JVM
// exists in this image, and the reference pins no TLC version, fingerprints or state counts
// (SURVEY.md §8c).  What pins this oracle:
//   * the reference's only golden vector, state_transfer_violation_trace.txt (24 states), replayed
//     step-by-step (tests/test_oracle_golden.py, fixtures in tests/golden/);
//   * an independently written Python restatement (oracle/pyoracle.py) that canonicalises by explicit
//     value comparison instead of hashing, compared on whole small state spaces.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use anything in oracle/.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace vsr_oracle {

typedef uint64_t u64;
typedef uint32_t u32;

// ---------------------------------------------------------------------------------------------
// Model constants (VSR.tla:92-117, bound in VSR.cfg:4-24)
// ---------------------------------------------------------------------------------------------
struct Params {
  int R = 3;               // ReplicaCount
  int C = 1;               // ClientCount
  int n = 2;               // Cardinality(Values); values are indexed 0..n-1 (v1..vn)
  int L = 2;               // StartViewOnTimerLimit
  int restart_limit = 0;   // RestartEmptyLimit (must be 0: recovery actions are dead)
  bool assume_commit_number = false;  // policy for VSR.tla:421 `m.commit`
  bool symmetry = true;    // SYMMETRY symmValues (VSR.cfg:31)
  int invariant_mask = 1;  // bit0 AcknowledgedWriteNotLost, bit1 AcknowledgedWritesExistOnMajority,
                           // bit2 NoLogDivergence (vacuous, VSR.tla:931), bit3 TestInv
};

Params params_from_array(const int* p);   // {R, C, n, L, restart_limit, assume_commit_number, symmetry, invariant_mask}

enum { Normal = 0, ViewChange = 1, Recovering = 2 };                       // VSR.tla:99-101
enum { T_SVC = 1, T_PREPARE = 2, T_PREPAREOK = 3, T_DVC = 4, T_SV = 5,      // VSR.tla:104-115 (live ones)
       T_GETSTATE = 6, T_NEWSTATE = 7 };

// Action ids in `Next` order (VSR.tla:896-918)
enum { A_TimerSendSVC = 1, A_ReceiveHigherSVC, A_ReceiveMatchingSVC, A_SendDVC, A_ReceiveHigherDVC,
       A_ReceiveMatchingDVC, A_SendSV, A_ReceiveSV, A_ReceiveClientRequest, A_ReceivePrepareMsg,
       A_ReceivePrepareOkMsg, A_ExecuteOp, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState };
extern const char* const ACTION_NAMES[16];

struct EvalError : std::runtime_error {   // TLC evaluation error (e.g. VSR.tla:421 nonexistent field)
  explicit EvalError(const std::string& s) : std::runtime_error(s) {}
};
struct RepError : std::runtime_error {    // a representation invariant or bound was broken
  explicit RepError(const std::string& s) : std::runtime_error(s) {}
};

// LogEntryType (VSR.tla:157-161).  view==0 means "no entry".
struct Entry {
  int view = 0, op = 0, client = 0, req = 0;   // op = value index 0..n-1
  bool operator==(const Entry& o) const { return view == o.view && op == o.op && client == o.client && req == o.req; }
  bool operator!=(const Entry& o) const { return !(*this == o); }
  bool operator<(const Entry& o) const {
    if (view != o.view) return view < o.view;
    if (op != o.op) return op < o.op;
    if (client != o.client) return client < o.client;
    return req < o.req;
  }
};

// A function lo..hi -> Entry; a TLA+ sequence when lo==1 (rep_log, DVC/SV .log); NewState.log has lo=first_op.
struct Log {
  int lo = 1, hi = 0;
  Entry e[5];                         // indexed by absolute op number 1..4
  int len() const { return hi >= lo ? hi - lo + 1 : 0; }
  bool operator==(const Log& o) const {
    if (len() != o.len()) return false;
    if (len() == 0) return true;      // <<>> = empty function whatever the bounds
    if (lo != o.lo) return false;
    for (int i = lo; i <= hi; i++) if (e[i] != o.e[i]) return false;
    return true;
  }
  bool operator<(const Log& o) const {
    if (len() != o.len()) return len() < o.len();
    if (len() == 0) return false;
    if (lo != o.lo) return lo < o.lo;
    for (int i = lo; i <= hi; i++) if (e[i] != o.e[i]) return e[i] < o.e[i];
    return false;
  }
};

// Union of the live message record types (VSR.tla:163-209, 510-514, 533-541); unused fields stay 0/empty.
struct Msg {
  int type = 0, view = 0, dest = 0, source = 0, op = 0, commit = 0, lnv = 0, first_op = 0;
  Entry entry;   // PrepareMsg.message
  Log log;       // DVC / SV / NewState .log
  bool operator==(const Msg& o) const {
    return type == o.type && view == o.view && dest == o.dest && source == o.source && op == o.op &&
           commit == o.commit && lnv == o.lnv && first_op == o.first_op && entry == o.entry && log == o.log;
  }
  bool operator<(const Msg& o) const {   // any total order consistent with ==; used only to keep sets sorted
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

struct ClientRow { int req = 0, op = 0; bool exec = true; };   // EmptyClientTableRow VSR.tla:318-321

struct Replica {                      // one column of the rep_* variables (VSR.tla:120-131)
  int status = Normal, view = 1, op = 0, commit = 0, lnv = 0;
  bool sent_dvc = false, sent_sv = false;
  Log log;
  int peer_op[6] = {0, 0, 0, 0, 0, 0};   // indexed by replica id 1..R
  ClientRow ct[3];                        // indexed by client id 1..C
  std::vector<Msg> svc_recv, dvc_recv;    // sets, kept sorted+unique
};

struct State {
  Replica rep[6];                               // indexed by replica id 1..R
  std::vector<std::pair<Msg, int>> messages;    // the bag: sorted by Msg, zero-count entries kept
  int aux_svc = 0;
  int acked[4] = {0, 0, 0, 0};                  // per value index: 0 = not in DOMAIN, 1 = FALSE, 2 = TRUE
};

struct Succ { int action; State st; };

// ---- semantics -------------------------------------------------------------------------------
State init_state(const Params& P);                                         // VSR.tla:323-348
void successors(const Params& P, const State& s, std::vector<Succ>& out);  // VSR.tla:896-918, all 15 live actions
int check_invariants(const Params& P, const State& s);                     // returns mask of VIOLATED invariants
State permute(const Params& P, const State& s, const int* pi);             // apply a Values permutation (VSR.tla:151)

// ---- packed format v1 (see DESIGN.md "Packed record"); the oracle's own encoder/decoder ---------
int words_per_replica(const Params& P);
int fixed_words(const Params& P);                                   // 1 + R*wpr
u64 enc_msg(const Params& P, const Msg& m, int count);
Msg dec_msg(const Params& P, u64 w, int* count);
void encode(const Params& P, const State& s, std::vector<u64>& out);  // appends one record
State decode(const Params& P, const u64* rec, int* nwords);

// ---- fingerprint of the VIEW under SYMMETRY (VSR.tla:149-151, VSR.cfg:29-31) ---------------------
struct Fp { u64 fp; u32 auxkey; int argmin; };
Fp fingerprint(const Params& P, const State& s);
u64 fmix64(u64 x);
void set_fp_seed(u64 seed);   // second-hash audit: xor-ed into every salt (process-global; 0 = the fixtures' function)
u64 fp_seed();
// version of the fingerprint FUNCTION (not of the state identity it hashes): 1 = per-replica chained hash (round 1),
// 2 = one salted term per word (Zobrist-style sum).  Fixtures that hold fingerprint values say which one they were made with.
const int FP_VERSION = 2;

}  // namespace vsr_oracle

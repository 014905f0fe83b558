// oracle/vsr_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See vsr_oracle.hpp.
// Every function cites the VSR.tla lines it restates.  "parity unpinned" vs TLC (see header).
#include <cstdlib>
#include "vsr_oracle.hpp"

#include <algorithm>
#include <cassert>

namespace vsr_oracle {

Params params_from_array(const int* p) {
  Params P;
  P.R = p[0];
  P.C = p[1];
  P.n = p[2];
  P.L = p[3];
  P.restart_limit = p[4];
  P.assume_commit_number = p[5] != 0;
  P.symmetry = p[6] != 0;
  P.invariant_mask = p[7];
  return P;
}

const char* const ACTION_NAMES[16] = {
    "Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC",
    "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest",
    "ReceivePrepareMsg", "ReceivePrepareOkMsg", "ExecuteOp", "SendGetState", "ReceiveGetState",
    "ReceiveNewState"};

// =============================================================================================
// Bag-of-messages algebra (VSR.tla:228-275)
// =============================================================================================
typedef std::vector<std::pair<Msg, int>> Bag;

static int bag_find(const Bag& b, const Msg& m) {   // index of m in DOMAIN b, or -1
  for (size_t i = 0; i < b.size(); i++)
    if (b[i].first == m) return (int)i;
  return -1;
}

// SendFunc (VSR.tla:228-231): IF m \in DOMAIN msgs THEN [msgs EXCEPT ![m] = @ + 1] ELSE msgs @@ (m :> 1)
static void SendFunc(const Msg& m, Bag& msgs) {
  int i = bag_find(msgs, m);
  if (i >= 0) {
    msgs[i].second += 1;
  } else {
    msgs.push_back(std::make_pair(m, 1));
    std::sort(msgs.begin(), msgs.end(),
              [](const std::pair<Msg, int>& a, const std::pair<Msg, int>& b) { return a.first < b.first; });
  }
}

// BroadcastFunc (VSR.tla:233-240): one copy per r \in replicas \ {source} with dest overwritten.
static void BroadcastFunc(const Params& P, const Msg& msg, int source, Bag& msgs) {
  for (int r = 1; r <= P.R; r++) {
    if (r == source) continue;
    Msg m = msg;
    m.dest = r;
    SendFunc(m, msgs);   // existing key -> +1, new key -> 1  (lines 237-240)
  }
}

// DiscardFunc (VSR.tla:244-245): [msgs EXCEPT ![m] = @ - 1]; the key stays in the domain at count 0.
static void DiscardFunc(const Msg& m, Bag& msgs) {
  int i = bag_find(msgs, m);
  if (i < 0) throw RepError("DiscardFunc: message not in DOMAIN messages");
  msgs[i].second -= 1;
}

// ReceivableMsg (VSR.tla:272-275)
static bool ReceivableMsg(const std::pair<Msg, int>& mc, int type, int r) {
  return mc.first.type == type && mc.first.dest == r && mc.second > 0;
}

// =============================================================================================
// Helpers (VSR.tla:281-308)
// =============================================================================================
static int Primary(const Params& P, int v) { return 1 + ((v - 1) % P.R); }                 // VSR.tla:287-288
static bool IsPrimary(const Params& P, const State& s, int r) { return Primary(P, s.rep[r].view) == r; }  // :290-291

static Msg NewSVCMessage(int r, int view_number) {   // VSR.tla:293-297 (dest Nil, replaced in broadcast)
  Msg m;
  m.type = T_SVC;
  m.view = view_number;
  m.dest = 0;
  m.source = r;
  return m;
}
static void ResetRecvMsgs(State& t, int r) { t.rep[r].svc_recv.clear(); t.rep[r].dvc_recv.clear(); }  // :299-301
static void ResetSentVars(State& t, int r) { t.rep[r].sent_dvc = false; t.rep[r].sent_sv = false; }    // :303-305
static int MinVal(int a, int b) { return a <= b ? a : b; }                                              // :307-308

static void set_insert(std::vector<Msg>& set, const Msg& m) {   // @ \union {m}
  for (const Msg& x : set)
    if (x == m) return;
  set.push_back(m);
  std::sort(set.begin(), set.end());
}

static Log Append(const Log& l, const Entry& e) {
  if (l.len() > 0 && l.lo != 1) throw RepError("Append on a non-sequence");
  Log o = l;
  o.lo = 1;
  o.hi = l.len() + 1;
  if (o.hi > 4) throw RepError("log longer than 4");
  o.e[o.hi] = e;
  return o;
}

// =============================================================================================
// Init (VSR.tla:323-348)
// =============================================================================================
State init_state(const Params& P) {
  if (P.R < 2 || P.R > 5 || P.C < 1 || P.C > 2 || P.n < 1 || P.n > 3 || P.L < 0 || P.L > 6)
    throw RepError("model constants outside the supported bounds (R 2..5, C 1..2, |Values| 1..3, L 0..6)");
  if (P.restart_limit != 0) throw RepError("RestartEmptyLimit > 0 is not supported (recovery actions)");
  State s;   // defaults = Normal, view 1, empty logs, zero counters, empty bag (lines 327-348)
  return s;
}

// =============================================================================================
// The 15 live actions.  Each appends its successors in binding order as written.
// =============================================================================================

// TimerSendSVC (VSR.tla:578-590)
static void TimerSendSVC(const Params& P, const State& s, std::vector<Succ>& out) {
  if (!(s.aux_svc < P.L)) return;                                    // :579
  for (int r = 1; r <= P.R; r++) {                                   // :580
    if (IsPrimary(P, s, r)) continue;                                // :581
    State t = s;
    t.rep[r].view = s.rep[r].view + 1;                               // :582
    t.rep[r].status = ViewChange;                                    // :583
    ResetRecvMsgs(t, r);                                             // :584
    ResetSentVars(t, r);                                             // :585
    t.aux_svc = s.aux_svc + 1;                                       // :586
    BroadcastFunc(P, NewSVCMessage(r, s.rep[r].view + 1), r, t.messages);   // :587
    out.push_back({A_TimerSendSVC, t});
  }
}

// ReceiveHigherSVC (VSR.tla:602-613)
static void ReceiveHigherSVC(const Params& P, const State& s, std::vector<Succ>& out) {
  for (size_t i = 0; i < s.messages.size(); i++) {                   // \E m \in DOMAIN messages, r \in replicas
    const Msg& m = s.messages[i].first;
    for (int r = 1; r <= P.R; r++) {
      if (!ReceivableMsg(s.messages[i], T_SVC, r)) continue;        // :604
      if (!(m.view > s.rep[r].view)) continue;                       // :605
      State t = s;
      t.rep[r].view = m.view;                                        // :606
      t.rep[r].status = ViewChange;                                  // :607
      t.rep[r].svc_recv.clear();                                     // :608  = {m}
      t.rep[r].svc_recv.push_back(m);
      t.rep[r].dvc_recv.clear();                                     // :609
      ResetSentVars(t, r);                                           // :610
      DiscardFunc(m, t.messages);                                    // :611 DiscardAndBroadcast
      BroadcastFunc(P, NewSVCMessage(r, m.view), r, t.messages);
      out.push_back({A_ReceiveHigherSVC, t});
    }
  }
}

// ReceiveMatchingSVC (VSR.tla:625-634)
static void ReceiveMatchingSVC(const Params& P, const State& s, std::vector<Succ>& out) {
  for (size_t i = 0; i < s.messages.size(); i++) {
    const Msg& m = s.messages[i].first;
    for (int r = 1; r <= P.R; r++) {
      if (!ReceivableMsg(s.messages[i], T_SVC, r)) continue;        // :627
      if (!(m.view == s.rep[r].view)) continue;                      // :628
      if (!(s.rep[r].status == ViewChange)) continue;                // :630
      State t = s;
      set_insert(t.rep[r].svc_recv, m);                              // :631
      DiscardFunc(m, t.messages);                                    // :632
      out.push_back({A_ReceiveMatchingSVC, t});
    }
  }
}

// SendDVC (VSR.tla:648-669)
static void SendDVC(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++) {
    const Replica& me = s.rep[r];
    if (!(me.status == ViewChange)) continue;                        // :650
    if (!(me.sent_dvc == false)) continue;                           // :651
    if (!((int)me.svc_recv.size() >= P.R / 2)) continue;             // :652
    State t = s;
    t.rep[r].sent_dvc = true;                                        // :653
    Msg msg;                                                         // :654-661
    msg.type = T_DVC;
    msg.view = me.view;
    msg.log = me.log;
    msg.lnv = me.lnv;
    msg.op = me.op;
    msg.commit = me.commit;
    msg.dest = Primary(P, me.view);
    msg.source = r;
    if (Primary(P, me.view) == r) {                                  // :662-664
      set_insert(t.rep[r].dvc_recv, msg);
    } else {                                                         // :665-667
      SendFunc(msg, t.messages);
    }
    out.push_back({A_SendDVC, t});
  }
}

// ReceiveHigherDVC (VSR.tla:677-688)
static void ReceiveHigherDVC(const Params& P, const State& s, std::vector<Succ>& out) {
  for (size_t i = 0; i < s.messages.size(); i++) {
    const Msg& m = s.messages[i].first;
    for (int r = 1; r <= P.R; r++) {
      if (!ReceivableMsg(s.messages[i], T_DVC, r)) continue;        // :679
      if (!(m.view > s.rep[r].view)) continue;                       // :680
      State t = s;
      t.rep[r].view = m.view;                                        // :681
      t.rep[r].status = ViewChange;                                  // :682
      t.rep[r].svc_recv.clear();                                     // :683
      t.rep[r].dvc_recv.clear();                                     // :684 = {m}
      t.rep[r].dvc_recv.push_back(m);
      ResetSentVars(t, r);                                           // :685
      DiscardFunc(m, t.messages);                                    // :686
      BroadcastFunc(P, NewSVCMessage(r, m.view), r, t.messages);
      out.push_back({A_ReceiveHigherDVC, t});
    }
  }
}

// ReceiveMatchingDVC (VSR.tla:696-703)
static void ReceiveMatchingDVC(const Params& P, const State& s, std::vector<Succ>& out) {
  for (size_t i = 0; i < s.messages.size(); i++) {
    const Msg& m = s.messages[i].first;
    for (int r = 1; r <= P.R; r++) {
      if (!ReceivableMsg(s.messages[i], T_DVC, r)) continue;        // :698
      if (!(s.rep[r].view == m.view)) continue;                      // :699
      State t = s;
      set_insert(t.rep[r].dvc_recv, m);                              // :700
      DiscardFunc(m, t.messages);                                    // :701
      out.push_back({A_ReceiveMatchingDVC, t});
    }
  }
}

// TLC's order on the DVC records of one rep_dvc_recv[r] (all share view_number, type, dest): record fields
// compare in normal-form order [view_number, type, op_number, commit_number, dest, source, log, last_normal_vn]
// (evidenced by trace:566).  CHOOSE returns the first element in that order that satisfies
// the predicate.
static bool tlc_dvc_less(const Msg& a, const Msg& b) {
  if (a.view != b.view) return a.view < b.view;
  if (a.op != b.op) return a.op < b.op;
  if (a.commit != b.commit) return a.commit < b.commit;
  if (a.dest != b.dest) return a.dest < b.dest;
  if (a.source != b.source) return a.source < b.source;
  if (!(a.log == b.log)) return a.log < b.log;
  return a.lnv < b.lnv;
}

// HighestLog (VSR.tla:716-722)
static Log HighestLog(const State& s, int r) {
  std::vector<Msg> set = s.rep[r].dvc_recv;
  std::sort(set.begin(), set.end(), tlc_dvc_less);
  for (const Msg& m : set) {                                         // CHOOSE m \in rep_dvc_recv[r] :
    bool dominated = false;
    for (const Msg& m1 : set) {                                      //   ~\E m1 \in rep_dvc_recv[r] :
      if (m1.lnv > m.lnv) dominated = true;                          //     :719
      if (m1.lnv == m.lnv && m1.op > m.op) dominated = true;         //     :720-721
    }
    if (!dominated) return m.log;
  }
  throw EvalError("HighestLog: CHOOSE over an empty set");
}
// HighestOpNumber (VSR.tla:724-727)
static int HighestOpNumber(const State& s, int r) {
  Log l = HighestLog(s, r);
  return l.len() == 0 ? 0 : l.len();
}
// HighestCommitNumber (VSR.tla:729-733)
static int HighestCommitNumber(const State& s, int r) {
  int best = -1;
  for (const Msg& m : s.rep[r].dvc_recv) best = std::max(best, m.commit);
  if (best < 0) throw EvalError("HighestCommitNumber: CHOOSE over an empty set");
  return best;
}

// SendSV (VSR.tla:735-760)
static void SendSV(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++) {
    const Replica& me = s.rep[r];
    if (!(me.status == ViewChange)) continue;                        // :737
    if (!(me.sent_sv == false)) continue;                            // :738
    if (!((int)me.dvc_recv.size() >= P.R / 2 + 1)) continue;         // :739
    Log new_log = HighestLog(s, r);                                  // :740
    int new_on = HighestOpNumber(s, r);                              // :741
    int new_cn = HighestCommitNumber(s, r);                          // :742
    State t = s;
    t.rep[r].status = Normal;                                        // :744
    t.rep[r].view = me.view;                                         // :745 (no-op, Q4)
    t.rep[r].log = new_log;                                          // :746
    t.rep[r].op = new_on;                                            // :747
    for (int r1 = 1; r1 <= P.R; r1++) t.rep[r].peer_op[r1] = 0;      // :748
    t.rep[r].commit = new_cn;                                        // :749
    t.rep[r].sent_sv = true;                                         // :750
    t.rep[r].lnv = me.view;                                          // :751
    Msg msg;                                                         // :752-758
    msg.type = T_SV;
    msg.view = me.view;
    msg.log = new_log;
    msg.op = new_on;
    msg.commit = new_cn;
    msg.source = r;
    BroadcastFunc(P, msg, r, t.messages);
    out.push_back({A_SendSV, t});
  }
}

// ReceiveSV (VSR.tla:773-793)
static void ReceiveSV(const Params& P, const State& s, std::vector<Succ>& out) {
  for (size_t i = 0; i < s.messages.size(); i++) {
    const Msg& m = s.messages[i].first;
    for (int r = 1; r <= P.R; r++) {
      if (!ReceivableMsg(s.messages[i], T_SV, r)) continue;         // :775
      if (!(m.view >= s.rep[r].view)) continue;                      // :776
      State t = s;
      t.rep[r].status = Normal;                                      // :777
      t.rep[r].view = m.view;                                        // :778
      t.rep[r].log = m.log;                                          // :779
      t.rep[r].op = m.op;                                            // :780
      t.rep[r].commit = m.commit;                                    // :781
      t.rep[r].lnv = m.view;                                         // :782
      ResetRecvMsgs(t, r);                                           // :783
      ResetSentVars(t, r);                                           // :784
      if (s.rep[r].commit < m.op) {                                  // :785 (old commit number)
        Msg ok;                                                      // :786-790
        ok.type = T_PREPAREOK;
        ok.view = m.view;
        ok.op = m.op;
        ok.dest = Primary(P, m.view);
        ok.source = r;
        DiscardFunc(m, t.messages);                                  // DiscardAndSend :267-270
        SendFunc(ok, t.messages);
      } else {
        DiscardFunc(m, t.messages);                                  // :791
      }
      out.push_back({A_ReceiveSV, t});
    }
  }
}

// ReceiveClientRequest (VSR.tla:366-394)
static void ReceiveClientRequest(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (int c = 1; c <= P.C; c++)
      for (int v = 0; v < P.n; v++) {                                // \E r \in replicas, c \in clients, v \in Values
        const Replica& me = s.rep[r];
        if (!IsPrimary(P, s, r)) continue;                           // :368
        if (!(me.status == Normal)) continue;                        // :369
        if (!(s.acked[v] == 0)) continue;                            // :370 v \notin DOMAIN aux_client_acked
        if (!(me.ct[c].exec == true)) continue;                      // :371
        int req_number = me.ct[c].req + 1;                           // :372
        int op_number = me.log.len() + 1;                            // :373
        Entry log_entry;                                             // :374-377
        log_entry.view = me.view;
        log_entry.op = v;
        log_entry.client = c;
        log_entry.req = req_number;
        State t = s;
        t.rep[r].log = Append(me.log, log_entry);                    // :379
        t.rep[r].op = op_number;                                     // :380
        t.rep[r].ct[c].req = req_number;                             // :381-384
        t.rep[r].ct[c].op = op_number;
        t.rep[r].ct[c].exec = false;
        Msg msg;                                                     // :385-391
        msg.type = T_PREPARE;
        msg.view = me.view;
        msg.entry = log_entry;
        msg.op = op_number;
        msg.commit = me.commit;
        msg.source = r;
        BroadcastFunc(P, msg, r, t.messages);
        t.acked[v] = 1;                                              // :392  @@ (v :> FALSE)
        out.push_back({A_ReceiveClientRequest, t});
      }
}

// ReceivePrepareMsg (VSR.tla:405-428)
static void ReceivePrepareMsg(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (size_t i = 0; i < s.messages.size(); i++) {                 // \E r \in replicas, m \in DOMAIN messages
      const Msg& m = s.messages[i].first;
      const Replica& me = s.rep[r];
      if (!ReceivableMsg(s.messages[i], T_PREPARE, r)) continue;    // :407
      if (!(me.status == Normal)) continue;                          // :408
      if (!(m.view == me.view)) continue;                            // :409
      if (!(m.op == me.op + 1)) continue;                            // :410
      State t = s;
      t.rep[r].log = Append(me.log, m.entry);                        // :411
      t.rep[r].op = m.op;                                            // :412
      t.rep[r].commit = m.commit;                                    // :413
      for (int c = 1; c <= P.C; c++) {                               // :414-421
        if (c == m.entry.client) {
          t.rep[r].ct[c].req = m.entry.req;
          t.rep[r].ct[c].op = m.op;
          t.rep[r].ct[c].exec = (m.op <= m.commit);
        } else {
          // VSR.tla:421 reads `m.commit`, a field PrepareMsg does not have -> TLC evaluation error
          if (!P.assume_commit_number)
            throw EvalError("VSR.tla:421: record has no field 'commit' (ReceivePrepareMsg, ClientCount >= 2)");
          t.rep[r].ct[c].exec = (me.ct[c].op <= m.commit);
        }
      }
      Msg ok;                                                        // :422-426
      ok.type = T_PREPAREOK;
      ok.view = me.view;
      ok.op = m.op;
      ok.dest = m.source;
      ok.source = r;
      DiscardFunc(m, t.messages);
      SendFunc(ok, t.messages);
      out.push_back({A_ReceivePrepareMsg, t});
    }
}

// ReceivePrepareOkMsg (VSR.tla:437-447)
static void ReceivePrepareOkMsg(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (size_t i = 0; i < s.messages.size(); i++) {
      const Msg& m = s.messages[i].first;
      const Replica& me = s.rep[r];
      if (!ReceivableMsg(s.messages[i], T_PREPAREOK, r)) continue;  // :439
      if (!IsPrimary(P, s, r)) continue;                             // :440
      if (!(me.status == Normal)) continue;                          // :441
      if (!(m.view == me.view)) continue;                            // :442
      if (!(m.op > me.peer_op[m.source])) continue;                  // :443
      State t = s;
      t.rep[r].peer_op[m.source] = m.op;                             // :444
      DiscardFunc(m, t.messages);                                    // :445
      out.push_back({A_ReceivePrepareOkMsg, t});
    }
}

// IsCommitted (VSR.tla:457-460)
static bool IsCommitted(const Params& P, const State& s, int r, int op_number) {
  int q = 0;
  for (int peer = 1; peer <= P.R; peer++)
    if (s.rep[r].peer_op[peer] >= op_number) q++;
  return q >= P.R / 2;
}

// ExecuteOp (VSR.tla:462-476)
static void ExecuteOp(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++) {
    const Replica& me = s.rep[r];
    if (!IsPrimary(P, s, r)) continue;                               // :464
    if (!(me.status == Normal)) continue;                            // :465
    if (!(me.commit < me.op)) continue;                              // :466
    if (!IsCommitted(P, s, r, me.commit + 1)) continue;              // :467
    int op_number = me.commit + 1;                                   // :468
    if (!(me.log.len() > 0 && me.log.lo == 1 && op_number <= me.log.hi))
      throw EvalError("ExecuteOp: rep_log[r][op_number] out of domain");
    Entry op = me.log.e[op_number];                                  // :469
    State t = s;
    t.rep[r].commit = op_number;                                     // :471
    t.rep[r].ct[op.client].exec = true;                              // :472
    if (s.acked[op.op] == 0) throw EvalError("ExecuteOp: operation not in DOMAIN aux_client_acked");
    t.acked[op.op] = 2;                                              // :473
    out.push_back({A_ExecuteOp, t});
  }
}

// TruncateLogToCommitNumber (VSR.tla:491-494)
static Log TruncateLogToCommitNumber(const State& s, int r, int truncate_to) {
  Log o;
  if (truncate_to == 0) return o;
  o.lo = 1;
  o.hi = truncate_to;
  for (int i = 1; i <= truncate_to; i++) o.e[i] = s.rep[r].log.e[i];
  return o;
}

// SendGetState (VSR.tla:496-516)
static void SendGetState(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (int rDest = 1; rDest <= P.R; rDest++)
      for (size_t i = 0; i < s.messages.size(); i++) {               // \E r, rDest \in replicas, m \in DOMAIN messages
        const Msg& m = s.messages[i].first;
        const Replica& me = s.rep[r];
        if (IsPrimary(P, s, r)) continue;                            // :498
        if (!(r != rDest)) continue;                                 // :499
        if (!ReceivableMsg(s.messages[i], T_PREPARE, r)) continue;  // :500
        if (!(me.status == Normal)) continue;                        // :501
        if (!(m.view > me.view)) continue;                           // :502
        if (!(m.op > me.op + 1)) continue;                           // :503
        int truncate_to = MinVal(me.commit, me.log.len());           // :504
        Msg gs;                                                      // :510-514
        gs.type = T_GETSTATE;
        gs.view = m.view;
        gs.op = truncate_to;
        gs.dest = rDest;
        gs.source = r;
        if (bag_find(s.messages, gs) >= 0) continue;                 // SendOnce :250-251
        State t = s;
        t.rep[r].log = TruncateLogToCommitNumber(s, r, truncate_to); // :506
        t.rep[r].op = truncate_to;                                   // :507
        t.rep[r].view = m.view;                                      // :508
        t.rep[r].lnv = m.view;                                       // :509
        SendFunc(gs, t.messages);                                    // :252
        out.push_back({A_SendGetState, t});
      }
}

// ReceiveGetState (VSR.tla:526-543)
static void ReceiveGetState(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (size_t i = 0; i < s.messages.size(); i++) {
      const Msg& m = s.messages[i].first;
      const Replica& me = s.rep[r];
      if (!ReceivableMsg(s.messages[i], T_GETSTATE, r)) continue;   // :528
      if (!(me.view == m.view)) continue;                            // :529
      if (!(me.status == Normal)) continue;                          // :530
      if (!(me.op > m.op)) continue;                                 // :531
      Msg ns;                                                        // :533-541
      ns.type = T_NEWSTATE;
      ns.view = me.view;
      ns.log.lo = m.op + 1;                                          // :535-536
      ns.log.hi = me.op;
      for (int on = m.op + 1; on <= me.op; on++) {
        if (!(me.log.lo == 1 && on <= me.log.hi)) throw EvalError("ReceiveGetState: rep_log[r][on] out of domain");
        ns.log.e[on] = me.log.e[on];
      }
      ns.first_op = m.op + 1;                                        // :537
      ns.op = me.op;                                                 // :538
      ns.commit = me.commit;                                         // :539
      ns.dest = m.source;                                            // :540
      ns.source = r;                                                 // :541
      State t = s;
      DiscardFunc(m, t.messages);
      SendFunc(ns, t.messages);
      out.push_back({A_ReceiveGetState, t});
    }
}

// ReceiveNewState (VSR.tla:551-567)
static void ReceiveNewState(const Params& P, const State& s, std::vector<Succ>& out) {
  for (int r = 1; r <= P.R; r++)
    for (size_t i = 0; i < s.messages.size(); i++) {
      const Msg& m = s.messages[i].first;
      const Replica& me = s.rep[r];
      if (!ReceivableMsg(s.messages[i], T_NEWSTATE, r)) continue;   // :553
      if (!(me.view == m.view)) continue;                            // :554
      if (!(me.status == Normal)) continue;                          // :555
      if (!(me.op == m.first_op - 1)) continue;                      // :556
      Log nl;                                                        // :557-561
      nl.lo = 1;
      nl.hi = m.op;
      for (int on = 1; on <= m.op; on++) {
        if (on <= me.op) {
          if (!(me.log.lo == 1 && on <= me.log.hi)) throw EvalError("ReceiveNewState: rep_log[r][on] out of domain");
          nl.e[on] = me.log.e[on];
        } else {
          if (!(on >= m.log.lo && on <= m.log.hi)) throw EvalError("ReceiveNewState: m.log[on] out of domain");
          nl.e[on] = m.log.e[on];
        }
      }
      State t = s;
      t.rep[r].log = nl;
      t.rep[r].op = m.op;                                            // :562
      DiscardFunc(m, t.messages);                                    // :564  (client table untouched, :563)
      out.push_back({A_ReceiveNewState, t});
    }
}

// Next (VSR.tla:896-918).  Recovery actions 16-19 (VSR.tla:813-894) are dead: RestartEmpty needs
// aux_restart < RestartEmptyLimit = 0 < 0, the other three need a RecoveryMsg / Recovering status that only
// RestartEmpty creates; init_state() rejects RestartEmptyLimit > 0.
void successors(const Params& P, const State& s, std::vector<Succ>& out) {
  TimerSendSVC(P, s, out);
  ReceiveHigherSVC(P, s, out);
  ReceiveMatchingSVC(P, s, out);
  SendDVC(P, s, out);
  ReceiveHigherDVC(P, s, out);
  ReceiveMatchingDVC(P, s, out);
  SendSV(P, s, out);
  ReceiveSV(P, s, out);
  ReceiveClientRequest(P, s, out);
  ReceivePrepareMsg(P, s, out);
  ReceivePrepareOkMsg(P, s, out);
  ExecuteOp(P, s, out);
  SendGetState(P, s, out);
  ReceiveGetState(P, s, out);
  ReceiveNewState(P, s, out);
}

// =============================================================================================
// Invariants (VSR.tla:926-952)
// =============================================================================================
static bool ReplicaHasOp(const State& s, int r, int v) {             // VSR.tla:933-935
  const Log& l = s.rep[r].log;
  for (int i = l.lo; i <= l.hi; i++)
    if (l.e[i].op == v) return true;
  return false;
}
int check_invariants(const Params& P, const State& s) {
  int bad = 0;
  if (P.invariant_mask & 1) {                                        // AcknowledgedWriteNotLost :945-950
    for (int v = 0; v < P.n; v++) {
      if (s.acked[v] != 2) continue;
      bool any = false;
      for (int r = 1; r <= P.R; r++) any = any || ReplicaHasOp(s, r, v);
      if (!any) bad |= 1;
    }
  }
  if (P.invariant_mask & 2) {                                        // AcknowledgedWritesExistOnMajority :937-943
    for (int v = 0; v < P.n; v++) {
      if (s.acked[v] != 2) continue;
      int q = 0;
      for (int r = 1; r <= P.R; r++) q += ReplicaHasOp(s, r, v) ? 1 : 0;
      if (!(q >= P.R / 2 + 1)) bad |= 2;
    }
  }
  // bit2 NoLogDivergence compares rep_log[r1][n] with itself (VSR.tla:931) -> always TRUE
  // bit3 TestInv == TRUE (VSR.tla:952)
  return bad;
}

// =============================================================================================
// SYMMETRY: apply a permutation of Values to every place a value occurs (VSR.tla:151)
// =============================================================================================
static Log permute_log(const Log& l, const int* pi) {
  Log o = l;
  for (int i = l.lo; i <= l.hi; i++) o.e[i].op = pi[l.e[i].op];
  return o;
}
static Msg permute_msg(const Msg& m, const int* pi) {
  Msg o = m;
  if (m.type == T_PREPARE) o.entry.op = pi[m.entry.op];
  o.log = permute_log(m.log, pi);
  return o;
}
State permute(const Params& P, const State& s, const int* pi) {
  State t = s;
  for (int r = 1; r <= P.R; r++) {
    t.rep[r].log = permute_log(s.rep[r].log, pi);
    for (Msg& m : t.rep[r].svc_recv) m = permute_msg(m, pi);
    for (Msg& m : t.rep[r].dvc_recv) m = permute_msg(m, pi);
    std::sort(t.rep[r].svc_recv.begin(), t.rep[r].svc_recv.end());
    std::sort(t.rep[r].dvc_recv.begin(), t.rep[r].dvc_recv.end());
  }
  for (auto& mc : t.messages) mc.first = permute_msg(mc.first, pi);
  std::sort(t.messages.begin(), t.messages.end(),
            [](const std::pair<Msg, int>& a, const std::pair<Msg, int>& b) { return a.first < b.first; });
  for (int v = 0; v < P.n; v++) t.acked[pi[v]] = s.acked[v];
  return t;
}

// =============================================================================================
// Packed format v1 — the oracle's own codec (DESIGN.md "Packed record").
// =============================================================================================
int words_per_replica(const Params& P) { return 1 + (P.R + 1 + 1) / 2; }
int fixed_words(const Params& P) { return 1 + P.R * words_per_replica(P); }

static u32 enc_entry(const Entry& e) {
  if (e.view == 0) return 0;
  if (e.view < 1 || e.view > 7 || e.op < 0 || e.op > 3 || e.client < 1 || e.client > 2 || e.req < 1 || e.req > 3)
    throw RepError("log entry field out of packed range");
  return (u32)e.view | ((u32)e.op << 3) | ((u32)(e.client - 1) << 5) | ((u32)e.req << 6);
}
static Entry dec_entry(u32 b) {
  Entry e;
  if (b == 0) return e;
  e.view = b & 7;
  e.op = (b >> 3) & 3;
  e.client = ((b >> 5) & 1) + 1;
  e.req = (b >> 6) & 3;
  return e;
}
// entry with op number i sits in byte i-1
static u32 enc_log(const Log& l) {
  u32 w = 0;
  for (int i = l.lo; i <= l.hi; i++) {
    if (i < 1 || i > 3) throw RepError("log position out of packed range (1..3)");
    u32 b = enc_entry(l.e[i]);
    if (b == 0) throw RepError("hole inside a log");
    w |= b << (8 * (i - 1));
  }
  return w;
}
static Log dec_log(u32 w) {
  Log l;
  int lo = 0, hi = 0;
  for (int i = 1; i <= 3; i++) {
    u32 b = (w >> (8 * (i - 1))) & 0xFF;
    if (b) {
      if (!lo) lo = i;
      hi = i;
      l.e[i] = dec_entry(b);
    }
  }
  if (lo) { l.lo = lo; l.hi = hi; }
  return l;
}

u64 enc_msg(const Params& P, const Msg& m, int count) {
  (void)P;
  if (m.type < 1 || m.type > 7 || m.view < 1 || m.view > 7 || m.dest < 1 || m.dest > 5 || m.source < 1 || m.source > 5 ||
      m.op < 0 || m.op > 3 || m.commit < 0 || m.commit > 3 || m.lnv < 0 || m.lnv > 7 || m.first_op < 0 || m.first_op > 3)
    throw RepError("message field out of packed range");
  if (count < 0 || count > 3) throw RepError("delivery count out of packed range (0..3)");
  u32 lg = (m.type == T_PREPARE) ? enc_entry(m.entry) : enc_log(m.log);
  return (u64)m.type | ((u64)m.view << 3) | ((u64)m.dest << 6) | ((u64)m.source << 9) | ((u64)m.op << 12) |
         ((u64)m.commit << 14) | ((u64)m.lnv << 16) | ((u64)m.first_op << 19) | ((u64)count << 21) | ((u64)lg << 32);
}
Msg dec_msg(const Params& P, u64 w, int* count) {
  (void)P;
  Msg m;
  m.type = w & 7;
  m.view = (w >> 3) & 7;
  m.dest = (w >> 6) & 7;
  m.source = (w >> 9) & 7;
  m.op = (w >> 12) & 3;
  m.commit = (w >> 14) & 3;
  m.lnv = (w >> 16) & 7;
  m.first_op = (w >> 19) & 3;
  if (count) *count = (w >> 21) & 3;
  u32 lg = (u32)(w >> 32);
  if (m.type == T_PREPARE) m.entry = dec_entry(lg & 0xFF);
  else m.log = dec_log(lg);
  return m;
}

static void enc_replica(const Params& P, const State& s, int r, u64* out) {
  const Replica& me = s.rep[r];
  if (me.view < 1 || me.view > 7 || me.op < 0 || me.op > 3 || me.commit < 0 || me.commit > 3 || me.lnv < 0 || me.lnv > 7)
    throw RepError("replica scalar out of packed range");
  if (me.op != me.log.len() || (me.log.len() > 0 && me.log.lo != 1)) throw RepError("I3 broken: rep_op_number != Len(rep_log)");
  u64 A = (u64)me.status | ((u64)me.view << 2) | ((u64)me.op << 5) | ((u64)me.commit << 7) | ((u64)me.lnv << 9) |
          ((u64)(me.sent_dvc ? 1 : 0) << 12) | ((u64)(me.sent_sv ? 1 : 0) << 13);
  for (const Msg& m : me.svc_recv) {   // I1: every record has dest=r, view=View(r)
    if (m.type != T_SVC || m.dest != r || m.view != me.view) throw RepError("I1 broken: foreign SVC in rep_svc_recv");
    A |= (u64)1 << (14 + (m.source - 1));
  }
  for (int p = 1; p <= P.R; p++) {
    if (me.peer_op[p] < 0 || me.peer_op[p] > 3) throw RepError("peer op out of range");
    A |= (u64)me.peer_op[p] << (19 + 2 * (p - 1));
  }
  for (int c = 1; c <= P.C; c++) {
    if (me.ct[c].req < 0 || me.ct[c].req > 3 || me.ct[c].op < 0 || me.ct[c].op > 3) throw RepError("client row out of range");
    u64 row = (u64)me.ct[c].req | ((u64)me.ct[c].op << 2) | ((u64)(me.ct[c].exec ? 1 : 0) << 4);
    A |= row << (29 + 5 * (c - 1));
  }
  u32 x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  x[0] = enc_log(me.log);
  for (const Msg& m : me.dvc_recv) {   // I2: dest=r, view=View(r), at most one per source
    if (m.type != T_DVC || m.dest != r || m.view != me.view) throw RepError("I2 broken: foreign DVC in rep_dvc_recv");
    if (x[m.source]) throw RepError("I2 broken: two DVCs from one source");
    x[m.source] = 1u | ((u32)m.lnv << 1) | ((u32)m.op << 4) | ((u32)m.commit << 6) | (enc_log(m.log) << 8);
  }
  out[0] = A;
  int wpr = words_per_replica(P);
  for (int k = 0; k < wpr - 1; k++) out[1 + k] = (u64)x[2 * k] | ((u64)x[2 * k + 1] << 32);
}

void encode(const Params& P, const State& s, std::vector<u64>& out) {
  if (s.messages.size() > 255) throw RepError("bag larger than 255 entries");
  if (s.aux_svc < 0 || s.aux_svc > 7) throw RepError("aux_svc out of range");
  u64 hdr = (u64)s.messages.size() | ((u64)s.aux_svc << 8);
  for (int v = 0; v < P.n; v++) hdr |= (u64)s.acked[v] << (11 + 2 * v);
  out.push_back(hdr);
  int wpr = words_per_replica(P);
  size_t base = out.size();
  out.resize(base + (size_t)P.R * wpr);
  for (int r = 1; r <= P.R; r++) enc_replica(P, s, r, &out[base + (size_t)(r - 1) * wpr]);
  for (const auto& mc : s.messages) out.push_back(enc_msg(P, mc.first, mc.second));
}

State decode(const Params& P, const u64* rec, int* nwords) {
  State s;
  u64 hdr = rec[0];
  int nmsg = hdr & 0xFF;
  s.aux_svc = (hdr >> 8) & 7;
  for (int v = 0; v < P.n; v++) s.acked[v] = (hdr >> (11 + 2 * v)) & 3;
  int wpr = words_per_replica(P);
  for (int r = 1; r <= P.R; r++) {
    const u64* b = rec + 1 + (r - 1) * wpr;
    u64 A = b[0];
    Replica& me = s.rep[r];
    me.status = A & 3;
    me.view = (A >> 2) & 7;
    me.op = (A >> 5) & 3;
    me.commit = (A >> 7) & 3;
    me.lnv = (A >> 9) & 7;
    me.sent_dvc = (A >> 12) & 1;
    me.sent_sv = (A >> 13) & 1;
    for (int src = 1; src <= P.R; src++)
      if ((A >> (14 + src - 1)) & 1) {
        Msg m = NewSVCMessage(src, me.view);
        m.dest = r;
        me.svc_recv.push_back(m);
      }
    std::sort(me.svc_recv.begin(), me.svc_recv.end());
    for (int p = 1; p <= P.R; p++) me.peer_op[p] = (A >> (19 + 2 * (p - 1))) & 3;
    for (int c = 1; c <= P.C; c++) {
      u64 row = (A >> (29 + 5 * (c - 1))) & 31;
      me.ct[c].req = row & 3;
      me.ct[c].op = (row >> 2) & 3;
      me.ct[c].exec = (row >> 4) & 1;
    }
    u32 x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int k = 0; k < wpr - 1; k++) {
      x[2 * k] = (u32)b[1 + k];
      x[2 * k + 1] = (u32)(b[1 + k] >> 32);
    }
    me.log = dec_log(x[0]);
    for (int src = 1; src <= P.R; src++)
      if (x[src] & 1) {
        Msg m;
        m.type = T_DVC;
        m.view = me.view;
        m.dest = r;
        m.source = src;
        m.lnv = (x[src] >> 1) & 7;
        m.op = (x[src] >> 4) & 3;
        m.commit = (x[src] >> 6) & 3;
        m.log = dec_log(x[src] >> 8);
        me.dvc_recv.push_back(m);
      }
    std::sort(me.dvc_recv.begin(), me.dvc_recv.end());
  }
  const u64* mw = rec + fixed_words(P);
  for (int j = 0; j < nmsg; j++) {
    int cnt = 0;
    Msg m = dec_msg(P, mw[j], &cnt);
    s.messages.push_back(std::make_pair(m, cnt));
  }
  std::sort(s.messages.begin(), s.messages.end(),
            [](const std::pair<Msg, int>& a, const std::pair<Msg, int>& b) { return a.first < b.first; });
  if (nwords) *nwords = fixed_words(P) + nmsg;
  return s;
}

// =============================================================================================
// Fingerprint of the canonical VIEW (DESIGN.md "Fingerprint").
//   view (VSR.tla:149-150) = every variable except aux_svc, aux_restart, aux_client_acked;
//   symmetry (VSR.tla:151)  = min over all permutations pi of Values of the hash of pi(view).
// The hash is a SUM over components (one per word of a replica column, salted with its position, and one per bag entry) so
// that it does not depend on the storage order of the bag and so that the HIP path can update it incrementally.
// =============================================================================================
u64 fmix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
static const u64 SALT_MSG = 0x9E3779B97F4A7C15ULL;
// Second-hash audit (vsrmc_model_set_fp_seed in the product): a seed xor-ed into every salt.  0 = the function of the committed fixtures.
// Process-global; the stand-alone drivers read VSR_ORACLE_FP_SEED (hex) once.
static u64 g_fp_seed = [] { const char* e = std::getenv("VSR_ORACLE_FP_SEED"); return e ? (u64)std::strtoull(e, nullptr, 16) : (u64)0; }();
void set_fp_seed(u64 seed) { g_fp_seed = seed; }
u64 fp_seed() { return g_fp_seed; }
// position salt of word k of replica r's column (Zobrist-style: every (position, word) pair contributes one independent term)
static u64 salt_word(int r, int k) { return fmix64(0xA0761D6478BD642FULL + (u64)(8 * r + k)); }

static u64 view_hash(const Params& P, const State& s) {
  std::vector<u64> rec;
  encode(P, s, rec);
  int wpr = words_per_replica(P);
  u64 sum = 0;
  for (int r = 1; r <= P.R; r++) {
    const u64* b = &rec[1 + (size_t)(r - 1) * wpr];
    for (int k = 0; k < wpr; k++) sum += fmix64(b[k] ^ (salt_word(r, k) ^ g_fp_seed));
  }
  for (size_t j = fixed_words(P); j < rec.size(); j++) sum += fmix64(rec[j] ^ (SALT_MSG ^ g_fp_seed));
  return sum;
}

Fp fingerprint(const Params& P, const State& s) {
  int pi[4] = {0, 1, 2, 3};
  Fp best;
  best.fp = 0;
  best.auxkey = 0;
  best.argmin = -1;
  int idx = 0;
  do {
    State t = permute(P, s, pi);
    u64 h = view_hash(P, t);
    u32 ak = (u32)t.aux_svc;
    for (int v = 0; v < P.n; v++) ak |= (u32)t.acked[v] << (3 + 2 * v);
    if (best.argmin < 0 || h < best.fp || (h == best.fp && ak < best.auxkey)) {
      best.fp = h;
      best.auxkey = ak;
      best.argmin = idx;
    }
    idx++;
  } while (P.symmetry && std::next_permutation(pi, pi + P.n));
  if (best.fp == 0) best.fp = 1;   // 0 is the empty-slot sentinel of the seen-set
  return best;
}

}  // namespace vsr_oracle

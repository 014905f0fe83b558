// oracle/vrst_oracle.cpp — CPU ORACLE for VR_STATE_TRANSFER.tla (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See vrst_oracle.hpp.
// Every function cites the lines of /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.tla it
// restates (VRST.tla:NNN).  Unpacked structs, sorted bag, whole-state copies: nothing here is shared with the HIP path.
#include <cstdlib>
#include "vrst_oracle.hpp"

#include <algorithm>

namespace vrst_oracle {

const char* const ACTION_NAMES[16] = {"Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC",
                                      "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest",
                                      "ReceivePrepareMsg", "ReceivePrepareOkMsg", "ExecuteOp", "SendGetState", "ReceiveGetState",
                                      "ReceiveNewState"};

Params params_from_array(const int* p) {
  Params P;
  P.R = p[0];
  P.C = p[1];
  P.n = p[2];
  P.L = p[3];
  P.no_progress_limit = p[4];
  P.symmetry = p[6] != 0;
  P.invariant_mask = p[7];
  return P;
}

static void check_params(const Params& P) {
  if (P.R < 2 || P.R > 5 || P.n < 1 || P.n > 3 || P.L < 0 || P.L > 6) throw RepError("model constants outside the supported bounds");
  if (P.no_progress_limit != 0) throw RepError("NoProgressChangeLimit > 0 is not supported (NoProgressChange, VRST.tla:765-776, is not restated)");
  if (P.symmetry) throw RepError("SYMMETRY is not supported for VR_STATE_TRANSFER (VRST.cfg:25-27 keeps it commented out)");
}

// ---- bag algebra (VRST.tla:165-211) -----------------------------------------------------------------------------------
typedef std::vector<std::pair<Msg, int>> Bag;

static int bag_find(const Bag& b, const Msg& m) {
  for (size_t i = 0; i < b.size(); i++)
    if (b[i].first == m) return (int)i;
  return -1;
}
static void bag_insert(Bag& b, const Msg& m, int count) {
  b.push_back(std::make_pair(m, count));
  std::sort(b.begin(), b.end(), [](const std::pair<Msg, int>& x, const std::pair<Msg, int>& y) { return x.first < y.first; });
}
// SendFunc(m, msgs, deliver_count), VRST.tla:165-168: an existing key gets count + 1 (whatever deliver_count is), a new key
// starts at deliver_count
static void send_func(Bag& b, const Msg& m, int deliver_count) {
  int i = bag_find(b, m);
  if (i >= 0) {
    if (b[i].second + 1 > 3) throw RepError("delivery count > 3");
    b[i].second += 1;
  } else {
    bag_insert(b, m, deliver_count);
  }
}
// BroadcastFunc, VRST.tla:170-177: one copy per replica other than the source, dest overwritten
static void broadcast_func(const Params& P, Bag& b, Msg msg, int source) {
  for (int r = 1; r <= P.R; r++) {
    if (r == source) continue;
    msg.dest = r;
    send_func(b, msg, 1);
  }
}
// DiscardFunc, VRST.tla:181-182: count - 1, the key stays
static void discard_func(Bag& b, const Msg& m) {
  int i = bag_find(b, m);
  if (i < 0 || b[i].second <= 0) throw RepError("discard of a message that is not receivable");
  b[i].second -= 1;
}
// ReceivableMsg(m, type, r), VRST.tla:213-218
static bool receivable(const std::pair<Msg, int>& mc, int type, int r) {
  const Msg& m = mc.first;
  return m.type == type && (m.dest == r || (m.dest == AnyDest && m.source != r)) && mc.second > 0;
}

// ---- helpers (VRST.tla:224-257) ---------------------------------------------------------------------------------------
static int primary(const Params& P, int v) { return 1 + ((v - 1) % P.R); }                          // :233-234
static bool is_normal_primary(const Params& P, const State& s, int r) {                             // :236-238
  return primary(P, s.rep[r].view) == r && s.rep[r].status == Normal;
}
static bool is_normal_backup(const Params& P, const State& s, int r) {                              // :240-242
  return !(primary(P, s.rep[r].view) == r) && s.rep[r].status == Normal;
}
static Msg svc_msg(int r, int view) {                                                               // NewSVCMessage :244-248
  Msg m;
  m.type = T_SVC;
  m.view = view;
  m.dest = 0;
  m.source = r;
  return m;
}
static void reset_sent(Replica& x) { x.sent_dvc = false; x.sent_sv = false; }                       // ResetSentVars :250-252
static bool can_progress(const State& s, int r) { return !s.rep[r].no_progress; }                   // :257

State init_state(const Params& P) {                                                                 // Init :267-283
  check_params(P);
  State s;
  for (int r = 1; r <= P.R; r++) {
    Replica& x = s.rep[r];
    x.status = Normal;
    x.view = 1;
    x.op = 0;
    x.commit = 0;
    x.lnv = 1;                       // :279 rep_last_normal_view = 1 (VSR.tla starts it at 0)
    x.sent_dvc = x.sent_sv = x.no_progress = false;
  }
  return s;
}

// ---- the 15 live actions, in Next order (VRST.tla:779-799) --------------------------------------------------------------
static void emit(std::vector<Succ>& out, int action, State&& t) {
  Succ sc;
  sc.action = action;
  sc.st = std::move(t);
  out.push_back(std::move(sc));
}

static void TimerSendSVC(const Params& P, const State& s, std::vector<Succ>& out) {                  // :522-535
  if (!(s.aux_svc < P.L)) return;                                                                    // :524
  for (int r = 1; r <= P.R; r++) {
    if (!can_progress(s, r)) continue;                                                               // :526
    if (is_normal_primary(P, s, r)) continue;                                                        // :527
    State t = s;
    if (s.rep[r].view + 1 > 7) throw RepError("view number > 7");
    t.rep[r].view = s.rep[r].view + 1;                                                               // :529
    t.rep[r].status = ViewChange;                                                                    // :530
    reset_sent(t.rep[r]);                                                                            // :531
    t.aux_svc = s.aux_svc + 1;                                                                       // :532
    broadcast_func(P, t.messages, svc_msg(r, s.rep[r].view + 1), r);                                 // :533
    emit(out, A_TimerSendSVC, std::move(t));
  }
}

static void ReceiveHigher(const Params& P, const State& s, std::vector<Succ>& out, int type, int action) {
  // ReceiveHigherSVC :545-556 / ReceiveHigherDVC :623-634 (identical but for the message type); \E m, r: m is the outer variable
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], type, r)) continue;
      if (!(m.view > s.rep[r].view)) continue;
      State t = s;
      t.rep[r].view = m.view;
      t.rep[r].status = ViewChange;
      reset_sent(t.rep[r]);
      discard_func(t.messages, m);                                                                   // DiscardAndBroadcast :200-206
      broadcast_func(P, t.messages, svc_msg(r, m.view), r);
      emit(out, action, std::move(t));
    }
}

static void ReceiveMatching(const Params& P, const State& s, std::vector<Succ>& out, int type, int action) {
  // ReceiveMatchingSVC :565-574 / ReceiveMatchingDVC :643-652: the message is only counted, i.e. its key drops to count 0
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      if (!can_progress(s, r)) continue;
      if (s.rep[r].status != ViewChange) continue;
      if (!receivable(s.messages[j], type, r)) continue;
      if (m.view != s.rep[r].view) continue;
      State t = s;
      discard_func(t.messages, m);
      emit(out, action, std::move(t));
    }
}

static void SendDVC(const Params& P, const State& s, std::vector<Succ>& out) {                       // :588-614
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (x.status != ViewChange) continue;                                                            // :592
    if (x.sent_dvc) continue;                                                                        // :593
    int q = 0;                                                                                       // :594-598 received (count 0) SVCs of this view
    for (const auto& mc : s.messages)
      if (mc.first.type == T_SVC && mc.first.dest == r && mc.first.view == x.view && mc.second == 0) q++;
    if (!(q >= P.R / 2)) continue;
    State t = s;
    t.rep[r].sent_dvc = true;                                                                        // :600
    Msg m;                                                                                           // :601-608
    m.type = T_DVC;
    m.view = x.view;
    m.log = x.log;
    m.lnv = x.lnv;
    m.op = x.op;
    m.commit = x.commit;
    m.dest = primary(P, x.view);
    m.source = r;
    if (primary(P, x.view) == r) send_func(t.messages, m, 0);                                        // SendAsReceived :609-610, :187-188
    else send_func(t.messages, m, 1);                                                                // Send :611-612
    emit(out, A_SendDVC, std::move(t));
  }
}

static bool valid_dvc(const State& s, int r, const std::pair<Msg, int>& mc) {                        // ValidDvc :666-670
  return mc.first.view == s.rep[r].view && mc.first.type == T_DVC && mc.first.dest == r && mc.second == 0;
}

static void SendSV(const Params& P, const State& s, std::vector<Succ>& out) {                        // :695-721
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (x.status != ViewChange) continue;                                                            // :699
    if (x.sent_sv) continue;                                                                         // :700
    int q = 0;
    for (const auto& mc : s.messages) q += valid_dvc(s, r, mc) ? 1 : 0;
    if (!(q >= P.R / 2 + 1)) continue;                                                               // :701
    // HighestLog (:672-680): CHOOSE among the valid DVCs maximal in (last_normal_vn, op_number); HighestCommitNumber (:687-693):
    // the largest commit_number.  CHOOSE picks the first such record in TLC's value order [TLC-RECALLED, as in vsr_oracle.cpp:
    // field order view_number, type, op_number, commit_number, dest, source, log, last_normal_vn] = smallest (commit, source).
    const Msg* best = nullptr;
    int max_commit = -1;
    for (const auto& mc : s.messages) {
      if (!valid_dvc(s, r, mc)) continue;
      const Msg& m = mc.first;
      max_commit = std::max(max_commit, m.commit);
      bool better = !best || m.lnv > best->lnv || (m.lnv == best->lnv && m.op > best->op) ||
                    (m.lnv == best->lnv && m.op == best->op && (m.commit < best->commit || (m.commit == best->commit && m.source < best->source)));
      if (better) best = &m;
    }
    if (!best) throw EvalError("CHOOSE over an empty set (HighestLog)");
    State t = s;
    Replica& y = t.rep[r];
    y.status = Normal;                                                                               // :707
    y.log = best->log;                                                                               // :708
    y.op = best->log.len();                                                                          // :709, HighestOpNumber :682-685
    for (int p = 1; p <= P.R; p++) y.peer_op[p] = 0;                                                 // :710
    y.commit = max_commit;                                                                           // :711
    y.sent_sv = true;                                                                                // :712
    y.lnv = x.view;                                                                                  // :713
    Msg m;                                                                                           // :714-720
    m.type = T_SV;
    m.view = x.view;
    m.log = best->log;
    m.op = best->log.len();
    m.commit = max_commit;
    m.source = r;
    broadcast_func(P, t.messages, m, r);
    emit(out, A_SendSV, std::move(t));
  }
}

static Msg prepare_ok(int view, int op, int dest, int source) {
  Msg m;
  m.type = T_PREPAREOK;
  m.view = view;
  m.op = op;
  m.dest = dest;
  m.source = source;
  return m;
}

static void ReceiveSV(const Params& P, const State& s, std::vector<Succ>& out) {                     // :733-756
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_SV, r)) continue;
      if (!((m.view == x.view && x.status == ViewChange) || m.view > x.view)) continue;              // :738-740
      State t = s;
      Replica& y = t.rep[r];
      y.status = Normal;                                                                             // :742
      y.view = m.view;                                                                               // :743
      y.log = m.log;                                                                                 // :744
      y.op = m.op;                                                                                   // :745
      y.commit = m.commit;                                                                           // :746
      y.lnv = m.view;                                                                                // :747
      reset_sent(y);                                                                                 // :748
      discard_func(t.messages, m);
      if (x.commit < m.op) send_func(t.messages, prepare_ok(m.view, m.op, primary(P, m.view), r), 1);   // :749-755 (old commit number)
      emit(out, A_ReceiveSV, std::move(t));
    }
}

static void ReceiveClientRequest(const Params& P, const State& s, std::vector<Succ>& out) {          // :298-318
  for (int r = 1; r <= P.R; r++)
    for (int v = 0; v < P.n; v++) {
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;                                                             // :301
      if (!is_normal_primary(P, s, r)) continue;                                                     // :302
      if (s.acked[v] != 0) continue;                                                                 // :303
      if (x.log.len() + 1 > 3) throw RepError("log longer than 3 entries");
      State t = s;
      Replica& y = t.rep[r];
      int opn = x.log.len() + 1;                                                                     // :305
      y.log.lo = 1;
      y.log.hi = opn;
      y.log.v[opn] = v;                                                                              // :308
      y.op = opn;                                                                                    // :309
      Msg m;                                                                                         // :310-316
      m.type = T_PREPARE;
      m.view = x.view;
      m.entry = v;
      m.op = opn;
      m.commit = x.commit;
      m.source = r;
      broadcast_func(P, t.messages, m, r);
      t.acked[v] = 1;                                                                                // :317
      emit(out, A_ReceiveClientRequest, std::move(t));
    }
}

static void ReceivePrepareMsg(const Params& P, const State& s, std::vector<Succ>& out) {             // :330-349
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!is_normal_backup(P, s, r)) continue;                                                      // :334
      if (!receivable(s.messages[j], T_PREPARE, r)) continue;
      if (m.view != x.view) continue;                                                                // :336
      if (m.op != x.op + 1) continue;                                                                // :337
      if (x.log.len() + 1 > 3) throw RepError("log longer than 3 entries");
      State t = s;
      Replica& y = t.rep[r];
      int pos = x.log.len() + 1;                                                                     // Append :339
      y.log.lo = 1;
      y.log.hi = pos;
      y.log.v[pos] = m.entry;
      y.op = m.op;                                                                                   // :340
      y.commit = m.commit;                                                                           // :341
      discard_func(t.messages, m);
      send_func(t.messages, prepare_ok(x.view, m.op, m.source, r), 1);                               // :342-346
      emit(out, A_ReceivePrepareMsg, std::move(t));
    }
}

static void ReceivePrepareOkMsg(const Params& P, const State& s, std::vector<Succ>& out) {           // :361-372
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!is_normal_primary(P, s, r)) continue;                                                     // :365
      if (!receivable(s.messages[j], T_PREPAREOK, r)) continue;
      if (m.view != x.view) continue;                                                                // :367
      if (!(m.op > x.peer_op[m.source])) continue;                                                   // :368
      State t = s;
      t.rep[r].peer_op[m.source] = m.op;                                                             // :370
      discard_func(t.messages, m);                                                                   // :371
      emit(out, A_ReceivePrepareOkMsg, std::move(t));
    }
}

static void ExecuteOp(const Params& P, const State& s, std::vector<Succ>& out) {                     // :389-405
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (!is_normal_primary(P, s, r)) continue;                                                       // :393
    if (!(x.commit < x.op)) continue;                                                                // :394
    int q = 0;                                                                                       // IsCommitted :384-387
    for (int p = 1; p <= P.R; p++) q += x.peer_op[p] >= x.commit + 1 ? 1 : 0;
    if (!(q >= P.R / 2)) continue;                                                                   // :395
    int opn = x.commit + 1;                                                                          // :397
    if (opn < x.log.lo || opn > x.log.hi) throw EvalError("rep_log[r][op_number] outside the log (ExecuteOp, VRST.tla:398)");
    int v = x.log.v[opn];
    State t = s;
    t.rep[r].commit = opn;                                                                           // :400
    if (s.acked[v] == 0) throw RepError("aux_client_acked EXCEPT on a key outside its domain");
    t.acked[v] = 2;                                                                                  // :401
    emit(out, A_ExecuteOp, std::move(t));
  }
}

static void SendGetState(const Params& P, const State& s, std::vector<Succ>& out) {                  // :431-447
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!is_normal_backup(P, s, r)) continue;                                                      // :435
      if (!receivable(s.messages[j], T_PREPARE, r)) continue;
      if (!(m.view > x.view)) continue;                                                              // :437
      if (!(m.op > x.op + 1)) continue;                                                              // :438
      Msg gs;                                                                                        // :441-445
      gs.type = T_GETSTATE;
      gs.view = m.view;
      gs.op = x.commit;
      gs.dest = AnyDest;
      gs.source = r;
      if (bag_find(s.messages, gs) >= 0) continue;                                                   // SendOnce :190-192
      State t = s;
      t.rep[r].status = StateTransfer;                                                               // :440
      send_func(t.messages, gs, 1);
      emit(out, A_SendGetState, std::move(t));
    }
}

static void ReceiveGetState(const Params& P, const State& s, std::vector<Succ>& out) {               // :460-478
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (x.status != Normal) continue;                                                              // :464
      if (!receivable(s.messages[j], T_GETSTATE, r)) continue;                                       // AnyDest: any replica but the sender
      if (x.view != m.view) continue;                                                                // :466
      if (!(x.op > m.op)) continue;                                                                  // :467
      Msg ns;                                                                                        // :470-477
      ns.type = T_NEWSTATE;
      ns.view = x.view;
      ns.log.lo = m.op + 1;
      ns.log.hi = x.op;
      for (int on = m.op + 1; on <= x.op; on++) {
        if (on < x.log.lo || on > x.log.hi) throw EvalError("rep_log[r][on] outside the log (ReceiveGetState, VRST.tla:472-473)");
        ns.log.v[on] = x.log.v[on];
      }
      ns.first_op = m.op + 1;
      ns.op = x.op;
      ns.commit = x.commit;
      ns.dest = m.source;
      ns.source = r;
      State t = s;
      discard_func(t.messages, m);
      send_func(t.messages, ns, 1);
      emit(out, A_ReceiveGetState, std::move(t));
    }
}

static void ReceiveNewState(const Params& P, const State& s, std::vector<Succ>& out) {               // :488-508
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (x.status != StateTransfer) continue;                                                       // :491
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_NEWSTATE, r)) continue;
      if (!(m.view > x.view)) continue;                                                              // :494
      State t = s;
      Replica& y = t.rep[r];
      y.status = Normal;                                                                             // :496
      y.view = m.view;                                                                               // :497
      y.lnv = m.view;                                                                                // :498
      Log nl;                                                                                        // :499-503
      nl.lo = 1;
      nl.hi = m.op;
      for (int on = 1; on <= m.op; on++) {
        if (on < m.first_op) {
          if (on < x.log.lo || on > x.log.hi) throw EvalError("rep_log[r][on] outside the log (ReceiveNewState, VRST.tla:501-502)");
          nl.v[on] = x.log.v[on];
        } else {
          if (on < m.log.lo || on > m.log.hi) throw EvalError("m.log[on] outside the message log (ReceiveNewState, VRST.tla:503)");
          nl.v[on] = m.log.v[on];
        }
      }
      y.log = nl;
      y.op = m.op;                                                                                   // :504
      y.commit = m.commit;                                                                           // :505
      discard_func(t.messages, m);                                                                   // :506
      emit(out, A_ReceiveNewState, std::move(t));
    }
}

void successors(const Params& P, const State& s, std::vector<Succ>& out) {                           // Next :779-799
  check_params(P);
  TimerSendSVC(P, s, out);
  ReceiveHigher(P, s, out, T_SVC, A_ReceiveHigherSVC);
  ReceiveMatching(P, s, out, T_SVC, A_ReceiveMatchingSVC);
  SendDVC(P, s, out);
  ReceiveHigher(P, s, out, T_DVC, A_ReceiveHigherDVC);
  ReceiveMatching(P, s, out, T_DVC, A_ReceiveMatchingDVC);
  SendSV(P, s, out);
  ReceiveSV(P, s, out);
  ReceiveClientRequest(P, s, out);
  ReceivePrepareMsg(P, s, out);
  ReceivePrepareOkMsg(P, s, out);
  ExecuteOp(P, s, out);
  SendGetState(P, s, out);
  ReceiveGetState(P, s, out);
  ReceiveNewState(P, s, out);
  // NoProgressChange (:765-776): no_progress_ctr < NoProgressChangeLimit = 0 is never true
}

// ---- invariants (VRST.tla:806-847) ------------------------------------------------------------------------------------
static bool replica_has_op(const State& s, int r, int v) {                                           // ReplicaHasOp :814-816
  const Log& l = s.rep[r].log;
  for (int i = l.lo; i <= l.hi; i++)
    if (l.v[i] == v) return true;
  return false;
}
int check_invariants(const Params& P, const State& s) {
  int bad = 0;
  for (int v = 0; v < P.n; v++) {
    if (s.acked[v] != 2) continue;
    int holders = 0;
    for (int r = 1; r <= P.R; r++) holders += replica_has_op(s, r, v) ? 1 : 0;
    if ((P.invariant_mask & 1) && holders == 0) bad |= 1;                                            // AcknowledgedWriteNotLost :830-835
    if ((P.invariant_mask & 2) && !(holders >= P.R / 2 + 1)) bad |= 2;                               // AcknowledgedWritesExistOnMajority :818-824
  }
  if (P.invariant_mask & 4) {                                                                        // NoLogDivergence :806-811
    for (int opn = 1; opn <= P.n; opn++)
      for (int r1 = 1; r1 <= P.R; r1++)
        for (int r2 = 1; r2 <= P.R; r2++) {
          if (!(opn <= s.rep[r1].commit && opn <= s.rep[r2].commit)) continue;
          const Log &a = s.rep[r1].log, &b = s.rep[r2].log;
          if (opn > a.hi || opn > b.hi) throw EvalError("rep_log[r][op_number] outside the log (NoLogDivergence, VRST.tla:811)");
          if (a.v[opn] != b.v[opn]) bad |= 4;
        }
  }
  if (P.invariant_mask & 8)                                                                          // CommitNumberNeverHigherThanOpNumber :845-847
    for (int r = 1; r <= P.R; r++)
      if (!(s.rep[r].commit <= s.rep[r].op)) bad |= 8;
  return bad;
}

// ---- packed format (see DESIGN.md "Second model") --------------------------------------------------------------------
//   [0]      header: nmsg(8) | aux_svc(3)<<8 | acked[v](2)<<(11+2v) | no_progress_ctr(3)<<20
//   [1..R]   replica word: status(2) view(3)<<2 op(2)<<5 commit(2)<<7 lnv(3)<<9 sent_dvc<<12 sent_sv<<13 no_progress<<14
//            peer_op[p](2)<<(15+2(p-1)) | log entry i (1..3): (1 | value<<1) << (25+3(i-1))
//   [1+R..)  bag: type(3) view(3)<<3 dest(3)<<6 source(3)<<9 op(2)<<12 commit(2)<<14 lnv(3)<<16 first_op(2)<<19 count(2)<<21 |
//            entries << 32, one byte per op number (byte on-1): 1 | value<<3     (Prepare: byte 0 = the entry)
int words_per_replica(const Params&) { return 1; }
int fixed_words(const Params& P) { return 1 + P.R; }

static u64 enc_log_bits(const Log& l) {
  u64 w = 0;
  if (l.len() && l.lo != 1) throw RepError("a replica log that does not start at op 1");
  for (int i = l.lo; i <= l.hi; i++) w |= (u64)(1 | (l.v[i] << 1)) << (3 * (i - 1));
  return w;
}
static u64 enc_msg_word(const Msg& m, int count) {
  u64 lg = 0;
  if (m.type == T_PREPARE) lg = (u64)(1 | (m.entry << 3));
  else for (int i = m.log.lo; i <= m.log.hi; i++) lg |= (u64)(1 | (m.log.v[i] << 3)) << (8 * (i - 1));
  if (count < 0 || count > 3) throw RepError("delivery count outside 0..3");
  return (u64)m.type | ((u64)m.view << 3) | ((u64)m.dest << 6) | ((u64)m.source << 9) | ((u64)m.op << 12) | ((u64)m.commit << 14) |
         ((u64)m.lnv << 16) | ((u64)m.first_op << 19) | ((u64)count << 21) | (lg << 32);
}

void encode(const Params& P, const State& s, std::vector<u64>& out) {
  if (s.messages.size() > 255) throw RepError("bag larger than 255 entries");
  u64 hdr = (u64)s.messages.size() | ((u64)s.aux_svc << 8) | ((u64)s.no_progress_ctr << 20);
  for (int v = 0; v < P.n; v++) hdr |= (u64)s.acked[v] << (11 + 2 * v);
  out.push_back(hdr);
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (x.view > 7 || x.op > 3 || x.commit > 3 || x.lnv > 7) throw RepError("replica field outside its packed range");
    if (x.op != x.log.len()) throw RepError("rep_op_number differs from Len(rep_log)");
    u64 A = (u64)x.status | ((u64)x.view << 2) | ((u64)x.op << 5) | ((u64)x.commit << 7) | ((u64)x.lnv << 9) |
            ((u64)x.sent_dvc << 12) | ((u64)x.sent_sv << 13) | ((u64)x.no_progress << 14);
    for (int p = 1; p <= P.R; p++) A |= (u64)x.peer_op[p] << (15 + 2 * (p - 1));
    A |= enc_log_bits(x.log) << 25;
    out.push_back(A);
  }
  for (const auto& mc : s.messages) out.push_back(enc_msg_word(mc.first, mc.second));
}

State decode(const Params& P, const u64* rec, int* nwords) {
  State s;
  const u64 hdr = rec[0];
  const int nmsg = (int)(hdr & 0xFF);
  s.aux_svc = (int)((hdr >> 8) & 7);
  s.no_progress_ctr = (int)((hdr >> 20) & 7);
  for (int v = 0; v < P.n; v++) s.acked[v] = (int)((hdr >> (11 + 2 * v)) & 3);
  for (int r = 1; r <= P.R; r++) {
    const u64 A = rec[r];
    Replica& x = s.rep[r];
    x.status = (int)(A & 3);
    x.view = (int)((A >> 2) & 7);
    x.op = (int)((A >> 5) & 3);
    x.commit = (int)((A >> 7) & 3);
    x.lnv = (int)((A >> 9) & 7);
    x.sent_dvc = (A >> 12) & 1;
    x.sent_sv = (A >> 13) & 1;
    x.no_progress = (A >> 14) & 1;
    for (int p = 1; p <= P.R; p++) x.peer_op[p] = (int)((A >> (15 + 2 * (p - 1))) & 3);
    x.log.lo = 1;
    x.log.hi = 0;
    for (int i = 1; i <= 3; i++) {
      const int e = (int)((A >> (25 + 3 * (i - 1))) & 7);
      if (e & 1) { x.log.hi = i; x.log.v[i] = e >> 1; }
    }
  }
  const u64* mw = rec + fixed_words(P);
  for (int j = 0; j < nmsg; j++) {
    const u64 w = mw[j];
    Msg m;
    m.type = (int)(w & 7);
    m.view = (int)((w >> 3) & 7);
    m.dest = (int)((w >> 6) & 7);
    m.source = (int)((w >> 9) & 7);
    m.op = (int)((w >> 12) & 3);
    m.commit = (int)((w >> 14) & 3);
    m.lnv = (int)((w >> 16) & 7);
    m.first_op = (int)((w >> 19) & 3);
    const int count = (int)((w >> 21) & 3);
    const u32 lg = (u32)(w >> 32);
    if (m.type == T_PREPARE) {
      m.entry = (int)((lg >> 3) & 3);
    } else if (m.type == T_DVC || m.type == T_SV || m.type == T_NEWSTATE) {
      m.log.lo = m.type == T_NEWSTATE ? m.first_op : 1;
      m.log.hi = m.log.lo - 1;
      for (int i = 1; i <= 3; i++) {
        const int e = (int)((lg >> (8 * (i - 1))) & 0xFF);
        if (e & 7) { m.log.hi = i; m.log.v[i] = (e >> 3) & 3; }
      }
      if (m.type != T_NEWSTATE && m.log.hi < 1) { m.log.lo = 1; m.log.hi = 0; }
    }
    s.messages.push_back(std::make_pair(m, count));
  }
  std::sort(s.messages.begin(), s.messages.end(),
            [](const std::pair<Msg, int>& a, const std::pair<Msg, int>& b) { return a.first < b.first; });
  if (nwords) *nwords = fixed_words(P) + nmsg;
  return s;
}

// ---- fingerprint of the VIEW (VRST.tla:97; VRST.cfg:23): everything but aux_svc and aux_client_acked ---------------------
// The same function family as the first model's (version 2): one salted term per replica word, one term per bag entry.
// no_progress_ctr is in the view but constant (limit 0); it lives in the header word, which is not hashed.
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
static u64 salt_word(int r, int k) { return fmix64(0xA0761D6478BD642FULL + (u64)(8 * r + k)); }

Fp fingerprint(const Params& P, const State& s) { return fingerprint_with_seed(P, s, g_fp_seed); }

Fp fingerprint_with_seed(const Params& P, const State& s, u64 seed) {
  std::vector<u64> rec;
  encode(P, s, rec);
  u64 sum = 0;
  for (int r = 1; r <= P.R; r++) sum += fmix64(rec[r] ^ (salt_word(r, 0) ^ seed));
  for (size_t j = fixed_words(P); j < rec.size(); j++) sum += fmix64(rec[j] ^ (SALT_MSG ^ seed));
  Fp f;
  f.fp = sum ? sum : 1;
  f.auxkey = (u32)s.aux_svc;
  for (int v = 0; v < P.n; v++) f.auxkey |= (u32)s.acked[v] << (3 + 2 * v);
  f.argmin = 0;
  return f;
}

}  // namespace vrst_oracle

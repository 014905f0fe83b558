// oracle/vras_oracle.cpp — CPU ORACLE for VR_APP_STATE.tla (TEST INFRASTRUCTURE, NOT PRODUCT CODE).  See vras_oracle.hpp.
// Every function cites the lines of /root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.tla it
// restates (VRAS.tla:NNN).  Unpacked structs, sorted bag, whole-state copies: nothing here is shared with the HIP path.
#include <cstdlib>
#include "vras_oracle.hpp"

#include <algorithm>

namespace vras_oracle {

const char* const ACTION_NAMES[16] = {"Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC",
                                      "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest",
                                      "ReceivePrepareMsg", "ReceivePrepareOkMsg", "PrimaryExecuteOp", "SendGetState", "ReceiveGetState",
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
  if (P.R < 2 || P.R > 3 || P.n < 1 || P.n > 3 || P.L < 0 || P.L > 6) throw RepError("model constants outside the supported bounds");
  if (P.no_progress_limit != 0) throw RepError("NoProgressChangeLimit > 0 is not supported (NoProgressChange, VRAS.tla:797-807, is not restated)");
  if (P.symmetry) throw RepError("SYMMETRY is not supported for VR_APP_STATE (VRAS.cfg:26-28 keeps it commented out)");
}

// ---- bag algebra (VRAS.tla:170-224) -----------------------------------------------------------------------------------
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
// SendFunc(m, msgs, deliver_count), VRAS.tla:170-173: an existing key gets count + 1 (whatever deliver_count is), a new key
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
// BroadcastFunc, VRAS.tla:175-182: one copy per replica other than the source, dest overwritten
static void broadcast_func(const Params& P, Bag& b, Msg msg, int source) {
  for (int r = 1; r <= P.R; r++) {
    if (r == source) continue;
    msg.dest = r;
    send_func(b, msg, 1);
  }
}
// DiscardFunc, VRAS.tla:186-187: count - 1, the key stays
static void discard_func(Bag& b, const Msg& m) {
  int i = bag_find(b, m);
  if (i < 0 || b[i].second <= 0) throw RepError("discard of a message that is not receivable");
  b[i].second -= 1;
}
// ReceivableMsg(m, type, r), VRAS.tla:218-223
static bool receivable(const std::pair<Msg, int>& mc, int type, int r) {
  const Msg& m = mc.first;
  return m.type == type && (m.dest == r || (m.dest == AnyDest && m.source != r)) && mc.second > 0;
}

// ---- helpers (VRAS.tla:229-283) ---------------------------------------------------------------------------------------
static int primary(const Params& P, int v) { return 1 + ((v - 1) % P.R); }                          // :238-239
static bool is_normal_primary(const Params& P, const State& s, int r) {                             // :241-243
  return primary(P, s.rep[r].view) == r && s.rep[r].status == Normal;
}
static bool is_normal_backup(const Params& P, const State& s, int r) {                              // :245-247
  return !(primary(P, s.rep[r].view) == r) && s.rep[r].status == Normal;
}
static Msg svc_msg(int r, int view) {                                                               // NewSVCMessage :249-253
  Msg m;
  m.type = T_SVC;
  m.view = view;
  m.dest = 0;
  m.source = r;
  return m;
}
static void set_add(std::vector<Msg>& set, const Msg& m) {                                          // @ \union {m}
  for (const Msg& x : set)
    if (x == m) return;
  set.push_back(m);
  std::sort(set.begin(), set.end());
}
static void reset_vc(Replica& x, const std::vector<Msg>& dvcs) {                                    // ResetVcVars :255-258
  x.sent_dvc = false;
  x.sent_sv = false;
  x.recv_dvc = dvcs;
}
static bool can_progress(const State& s, int r) { return !s.rep[r].no_progress; }                   // :263
// MaybeExecuteOps(r, log, old_commit, new_commit) :277-283 with AppendOps :270-275: the operations old_commit+1 .. new_commit of
// `log` are appended to the application state; log[op] outside the log is a TLC evaluation error
static void maybe_execute_ops(Replica& y, const Log& log, int old_commit, int new_commit) {
  if (!(new_commit > old_commit)) return;                                                           // :278, :283
  for (int op = old_commit + 1; op <= new_commit; op++) {                                           // :271-275
    if (op < log.lo || op > log.hi) throw EvalError("log[op] outside the log (AppendOps, VRAS.tla:274)");
    y.app_state.push_back(log.v[op]);
  }
  y.commit = new_commit;                                                                            // :281
}

State init_state(const Params& P) {                                                                 // Init :292-315
  check_params(P);
  State s;
  for (int r = 1; r <= P.R; r++) {
    Replica& x = s.rep[r];
    x.status = Normal;
    x.view = 1;
    x.op = 0;
    x.commit = 0;
    x.lnv = 1;                       // :307
    x.sent_dvc = x.sent_sv = x.no_progress = false;
  }
  return s;
}

// ---- the 15 live actions, in Next order (VRAS.tla:811-831) --------------------------------------------------------------
static void emit(std::vector<Succ>& out, int action, State&& t) {
  Succ sc;
  sc.action = action;
  sc.st = std::move(t);
  out.push_back(std::move(sc));
}

static void TimerSendSVC(const Params& P, const State& s, std::vector<Succ>& out) {                  // :551-565
  if (!(s.aux_svc < P.L)) return;                                                                    // :553
  for (int r = 1; r <= P.R; r++) {
    if (!can_progress(s, r)) continue;                                                               // :555
    if (is_normal_primary(P, s, r)) continue;                                                        // :556
    State t = s;
    if (s.rep[r].view + 1 > 7) throw RepError("view number > 7");
    t.rep[r].view = s.rep[r].view + 1;                                                               // :558
    t.rep[r].status = ViewChange;                                                                    // :559
    reset_vc(t.rep[r], {});                                                                          // :560
    t.aux_svc = s.aux_svc + 1;                                                                       // :561
    broadcast_func(P, t.messages, svc_msg(r, s.rep[r].view + 1), r);                                 // :562
    emit(out, A_TimerSendSVC, std::move(t));
  }
}

static void ReceiveHigher(const Params& P, const State& s, std::vector<Succ>& out, int type, int action) {
  // ReceiveHigherSVC :575-587 / ReceiveHigherDVC :656-668: the same but for the message type and what ResetVcVars keeps — nothing
  // (:584) or the DoViewChange just received (:665); \E m, r: m is the outer variable
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], type, r)) continue;
      if (!(m.view > s.rep[r].view)) continue;
      State t = s;
      t.rep[r].view = m.view;
      t.rep[r].status = ViewChange;
      reset_vc(t.rep[r], type == T_DVC ? std::vector<Msg>{m} : std::vector<Msg>{});
      discard_func(t.messages, m);                                                                   // DiscardAndBroadcast :205-211
      broadcast_func(P, t.messages, svc_msg(r, m.view), r);
      emit(out, action, std::move(t));
    }
}

static void ReceiveMatchingSVC(const Params& P, const State& s, std::vector<Succ>& out) {            // :595-606
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      if (!can_progress(s, r)) continue;
      if (s.rep[r].status != ViewChange) continue;                                                   // :599
      if (!receivable(s.messages[j], T_SVC, r)) continue;
      if (m.view != s.rep[r].view) continue;                                                         // :601
      if (s.rep[r].sent_dvc) continue;                                                               // :602 "reduce state space"
      State t = s;
      discard_func(t.messages, m);                                                                   // :604: the key stays with count 0
      emit(out, A_ReceiveMatchingSVC, std::move(t));
    }
}

static void SendDVC(const Params& P, const State& s, std::vector<Succ>& out) {                       // :619-647
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (x.status != ViewChange) continue;                                                            // :623
    if (x.sent_dvc) continue;                                                                        // :624
    int q = 0;                                                                                       // :625-629 received (count 0) SVCs of this view
    for (const auto& mc : s.messages)
      if (mc.first.type == T_SVC && mc.first.dest == r && mc.first.view == x.view && mc.second == 0) q++;
    if (!(q >= P.R / 2)) continue;
    State t = s;
    t.rep[r].sent_dvc = true;                                                                        // :631
    Msg m;                                                                                           // :632-639
    m.type = T_DVC;
    m.view = x.view;
    m.log = x.log;
    m.lnv = x.lnv;
    m.op = x.op;
    m.commit = x.commit;
    m.dest = primary(P, x.view);
    m.source = r;
    if (primary(P, x.view) == r) {
      send_func(t.messages, m, 0);                                                                   // SendAsReceived :640-641, :192-193
      set_add(t.rep[r].recv_dvc, m);                                                                 // :642
    } else {
      send_func(t.messages, m, 1);                                                                   // Send :643-645
    }
    emit(out, A_SendDVC, std::move(t));
  }
}

static void ReceiveMatchingDVC(const Params& P, const State& s, std::vector<Succ>& out) {            // :676-687
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      if (!can_progress(s, r)) continue;
      if (s.rep[r].status != ViewChange) continue;                                                   // :680
      if (!receivable(s.messages[j], T_DVC, r)) continue;
      if (m.view != s.rep[r].view) continue;                                                         // :682
      State t = s;
      discard_func(t.messages, m);                                                                   // :684
      set_add(t.rep[r].recv_dvc, m);                                                                 // :685
      emit(out, A_ReceiveMatchingDVC, std::move(t));
    }
}

static bool valid_dvc(const State& s, int r, const Msg& m) { return m.view == s.rep[r].view; }       // ValidDvc :700-701

static void SendSV(const Params& P, const State& s, std::vector<Succ>& out) {                        // :726-754
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (x.status != ViewChange) continue;                                                            // :730
    if (x.sent_sv) continue;                                                                         // :731
    int q = 0;
    for (const Msg& m : x.recv_dvc) q += valid_dvc(s, r, m) ? 1 : 0;
    if (!(q >= P.R / 2 + 1)) continue;                                                               // :732
    // HighestLog (:703-711): CHOOSE among the valid DVCs maximal in (last_normal_vn, op_number); HighestCommitNumber (:718-724):
    // the largest commit_number.  CHOOSE picks the first such record in TLC's value order [TLC-RECALLED, as in vrst_oracle.cpp]
    // = smallest (commit, source).
    const Msg* best = nullptr;
    int max_commit = -1;
    for (const Msg& m : x.recv_dvc) {
      if (!valid_dvc(s, r, m)) continue;
      max_commit = std::max(max_commit, m.commit);
      bool better = !best || m.lnv > best->lnv || (m.lnv == best->lnv && m.op > best->op) ||
                    (m.lnv == best->lnv && m.op == best->op && (m.commit < best->commit || (m.commit == best->commit && m.source < best->source)));
      if (better) best = &m;
    }
    if (!best) throw EvalError("CHOOSE over an empty set (HighestLog)");
    const Log new_log = best->log;                                                                   // :734
    const int new_on = new_log.len();                                                                // :735, HighestOpNumber :713-716
    State t = s;
    Replica& y = t.rep[r];
    y.status = Normal;                                                                               // :738
    y.log = new_log;                                                                                 // :739
    maybe_execute_ops(y, new_log, x.commit, max_commit);                                             // :740
    y.op = new_on;                                                                                   // :741
    for (int p = 1; p <= P.R; p++) y.peer_op[p] = 0;                                                 // :742
    y.sent_sv = true;                                                                                // :743
    y.recv_dvc.clear();                                                                              // :744
    y.lnv = x.view;                                                                                  // :745
    Msg m;                                                                                           // :746-752
    m.type = T_SV;
    m.view = x.view;
    m.log = new_log;
    m.op = new_on;
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

static void ReceiveSV(const Params& P, const State& s, std::vector<Succ>& out) {                     // :765-788
  for (size_t j = 0; j < s.messages.size(); j++)
    for (int r = 1; r <= P.R; r++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_SV, r)) continue;
      if (!((m.view == x.view && x.status == ViewChange) || m.view > x.view)) continue;              // :770-772
      State t = s;
      Replica& y = t.rep[r];
      y.status = Normal;                                                                             // :774
      y.view = m.view;                                                                               // :775
      y.log = m.log;                                                                                 // :776
      maybe_execute_ops(y, m.log, x.commit, m.commit);                                               // :777
      y.op = m.op;                                                                                   // :778
      y.lnv = m.view;                                                                                // :779
      reset_vc(y, {});                                                                               // :780
      discard_func(t.messages, m);
      if (x.commit < m.op) send_func(t.messages, prepare_ok(m.view, m.op, primary(P, m.view), r), 1);   // :781-787 (old commit number)
      emit(out, A_ReceiveSV, std::move(t));
    }
}

static void ReceiveClientRequest(const Params& P, const State& s, std::vector<Succ>& out) {          // :328-349
  for (int r = 1; r <= P.R; r++)
    for (int v = 0; v < P.n; v++) {
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;                                                             // :331
      if (!is_normal_primary(P, s, r)) continue;                                                     // :332
      if (s.acked[v] != 0) continue;                                                                 // :333
      if (x.log.len() + 1 > 3) throw RepError("log longer than 3 entries");
      State t = s;
      Replica& y = t.rep[r];
      int opn = x.log.len() + 1;                                                                     // :335
      y.log.lo = 1;
      y.log.hi = opn;
      y.log.v[opn] = v;                                                                              // :338
      y.op = opn;                                                                                    // :339
      Msg m;                                                                                         // :340-346
      m.type = T_PREPARE;
      m.view = x.view;
      m.entry = v;
      m.op = opn;
      m.commit = x.commit;
      m.source = r;
      broadcast_func(P, t.messages, m, r);
      t.acked[v] = 1;                                                                                // :347
      emit(out, A_ReceiveClientRequest, std::move(t));
    }
}

static void ReceivePrepareMsg(const Params& P, const State& s, std::vector<Succ>& out) {             // :360-380
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_PREPARE, r)) continue;
      if (!is_normal_backup(P, s, r)) continue;                                                      // :365
      if (m.view != x.view) continue;                                                                // :366
      if (m.op != x.op + 1) continue;                                                                // :367
      if (x.log.len() + 1 > 3) throw RepError("log longer than 3 entries");
      State t = s;
      Replica& y = t.rep[r];
      int pos = x.log.len() + 1;                                                                     // Append :369
      y.log.lo = 1;
      y.log.hi = pos;
      y.log.v[pos] = m.entry;                                                                        // :371
      y.op = m.op;                                                                                   // :372
      maybe_execute_ops(y, y.log, x.commit, m.commit);                                               // :373
      discard_func(t.messages, m);
      send_func(t.messages, prepare_ok(x.view, m.op, m.source, r), 1);                               // :374-378
      emit(out, A_ReceivePrepareMsg, std::move(t));
    }
}

static void ReceivePrepareOkMsg(const Params& P, const State& s, std::vector<Succ>& out) {           // :393-405
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_PREPAREOK, r)) continue;
      if (!is_normal_primary(P, s, r)) continue;                                                     // :398
      if (m.view != x.view) continue;                                                                // :399
      if (!(m.op > x.peer_op[m.source])) continue;                                                   // :400
      State t = s;
      t.rep[r].peer_op[m.source] = m.op;                                                             // :402
      discard_func(t.messages, m);                                                                   // :403
      emit(out, A_ReceivePrepareOkMsg, std::move(t));
    }
}

static void PrimaryExecuteOp(const Params& P, const State& s, std::vector<Succ>& out) {              // :420-435
  for (int r = 1; r <= P.R; r++) {
    const Replica& x = s.rep[r];
    if (!can_progress(s, r)) continue;
    if (!is_normal_primary(P, s, r)) continue;                                                       // :424
    if (!(x.commit < x.op)) continue;                                                                // :425
    int q = 0;                                                                                       // IsCommitted :415-418
    for (int p = 1; p <= P.R; p++) q += x.peer_op[p] >= x.commit + 1 ? 1 : 0;
    if (!(q >= P.R / 2)) continue;                                                                   // :426
    int new_commit = x.commit + 1;                                                                   // :428
    if (new_commit < x.log.lo || new_commit > x.log.hi) throw EvalError("rep_log[r][new_commit] outside the log (PrimaryExecuteOp, VRAS.tla:429)");
    int v = x.log.v[new_commit];
    State t = s;
    maybe_execute_ops(t.rep[r], x.log, x.commit, new_commit);                                        // :431
    if (s.acked[v] == 0) throw RepError("aux_client_acked EXCEPT on a key outside its domain");
    t.acked[v] = 2;                                                                                  // :432
    emit(out, A_PrimaryExecuteOp, std::move(t));
  }
}

static void SendGetState(const Params& P, const State& s, std::vector<Succ>& out) {                  // :461-476
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!is_normal_backup(P, s, r)) continue;                                                      // :464
      if (!receivable(s.messages[j], T_PREPARE, r)) continue;
      if (!(m.view > x.view)) continue;                                                              // :466
      if (!(m.op > x.op + 1)) continue;                                                              // :467
      Msg gs;                                                                                        // :469-473
      gs.type = T_GETSTATE;
      gs.view = m.view;
      gs.op = x.commit;
      gs.dest = AnyDest;
      gs.source = r;
      if (bag_find(s.messages, gs) >= 0) continue;                                                   // SendOnce :195-197
      State t = s;
      t.rep[r].status = StateTransfer;                                                               // :468
      send_func(t.messages, gs, 1);
      emit(out, A_SendGetState, std::move(t));
    }
}

static void ReceiveGetState(const Params& P, const State& s, std::vector<Succ>& out) {               // :490-507
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_GETSTATE, r)) continue;                                       // AnyDest: any replica but the sender
      if (x.view != m.view) continue;                                                                // :494
      if (x.status != Normal) continue;                                                              // :495
      if (!(x.op > m.op)) continue;                                                                  // :496
      Msg ns;                                                                                        // :498-505
      ns.type = T_NEWSTATE;
      ns.view = x.view;
      // LogSuffix(r, rep_log[r], m.op_number) :265-268: <<>> when Len(log) <= op_number, else [op \in op_number+1..Len(log) |-> log[op]]
      if (x.log.len() <= m.op) {
        ns.log.lo = 1;
        ns.log.hi = 0;
      } else {
        ns.log.lo = m.op + 1;
        ns.log.hi = x.log.len();
        for (int on = m.op + 1; on <= x.log.len(); on++) ns.log.v[on] = x.log.v[on];
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

static void ReceiveNewState(const Params& P, const State& s, std::vector<Succ>& out) {               // :516-537
  for (int r = 1; r <= P.R; r++)
    for (size_t j = 0; j < s.messages.size(); j++) {
      const Msg& m = s.messages[j].first;
      const Replica& x = s.rep[r];
      if (x.status != StateTransfer) continue;                                                       // :519
      if (!can_progress(s, r)) continue;
      if (!receivable(s.messages[j], T_NEWSTATE, r)) continue;
      if (!(m.view > x.view)) continue;                                                              // :522
      Log nl;                                                                                        // :524-527
      nl.lo = 1;
      nl.hi = m.op;
      for (int on = 1; on <= m.op; on++) {
        if (on < m.first_op) {
          if (on < x.log.lo || on > x.log.hi) throw EvalError("rep_log[r][op] outside the log (ReceiveNewState, VRAS.tla:526)");
          nl.v[on] = x.log.v[on];
        } else {
          if (on < m.log.lo || on > m.log.hi) throw EvalError("m.log[op] outside the message log (ReceiveNewState, VRAS.tla:527)");
          nl.v[on] = m.log.v[on];
        }
      }
      State t = s;
      Replica& y = t.rep[r];
      y.status = Normal;                                                                             // :529
      y.view = m.view;                                                                               // :530
      y.lnv = m.view;                                                                                // :531
      y.log = nl;                                                                                    // :532
      maybe_execute_ops(y, nl, x.commit, m.commit);                                                  // :533
      y.op = m.op;                                                                                   // :534
      discard_func(t.messages, m);                                                                   // :535
      emit(out, A_ReceiveNewState, std::move(t));
    }
}

void successors(const Params& P, const State& s, std::vector<Succ>& out) {                           // Next :811-831
  check_params(P);
  TimerSendSVC(P, s, out);
  ReceiveHigher(P, s, out, T_SVC, A_ReceiveHigherSVC);
  ReceiveMatchingSVC(P, s, out);
  SendDVC(P, s, out);
  ReceiveHigher(P, s, out, T_DVC, A_ReceiveHigherDVC);
  ReceiveMatchingDVC(P, s, out);
  SendSV(P, s, out);
  ReceiveSV(P, s, out);
  ReceiveClientRequest(P, s, out);
  ReceivePrepareMsg(P, s, out);
  ReceivePrepareOkMsg(P, s, out);
  PrimaryExecuteOp(P, s, out);
  SendGetState(P, s, out);
  ReceiveGetState(P, s, out);
  ReceiveNewState(P, s, out);
  // NoProgressChange (:797-807): no_progress_ctr < NoProgressChangeLimit = 0 is never true
}

// ---- invariants (VRAS.tla:840-894) ------------------------------------------------------------------------------------
static bool replica_has_op(const State& s, int r, int v) {                                           // ReplicaHasOp :861-863
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
    if ((P.invariant_mask & 1) && holders == 0) bad |= 1;                                            // AcknowledgedWriteNotLost :877-882
    if ((P.invariant_mask & 2) && !(holders >= P.R / 2 + 1)) bad |= 2;                               // AcknowledgedWritesExistOnMajority :865-871
  }
  if (P.invariant_mask & (4 | 16)) {
    for (int opn = 1; opn <= P.n; opn++)
      for (int r1 = 1; r1 <= P.R; r1++)
        for (int r2 = 1; r2 <= P.R; r2++) {
          if (!(opn <= s.rep[r1].commit && opn <= s.rep[r2].commit)) continue;                       // :843-844 / :855-856
          const Log &a = s.rep[r1].log, &b = s.rep[r2].log;
          if (P.invariant_mask & 4) {                                                                // NoLogDivergence :840-845
            if (opn > a.hi || opn > b.hi) throw EvalError("rep_log[r][op_number] outside the log (NoLogDivergence, VRAS.tla:845)");
            if (a.v[opn] != b.v[opn]) bad |= 4;
          }
          if (P.invariant_mask & 16) {                                                               // NoAppStateDivergence :852-858
            const std::vector<int>&x = s.rep[r1].app_state, &y = s.rep[r2].app_state;
            if (opn > (int)x.size() || opn > (int)y.size()) throw EvalError("rep_app_state[r][op_number] outside the sequence (NoAppStateDivergence, VRAS.tla:857)");
            if (x[opn - 1] != y[opn - 1]) {                                                          // :857
              if (opn > a.hi) throw EvalError("rep_log[r1][op_number] outside the log (NoAppStateDivergence, VRAS.tla:858)");
              if (a.v[opn] == x[opn - 1]) bad |= 16;                                                 // :858
            }
          }
        }
  }
  if (P.invariant_mask & 8)                                                                          // CommitNumberNeverHigherThanOpNumber :892-894
    for (int r = 1; r <= P.R; r++)
      if (!(s.rep[r].commit <= s.rep[r].op)) bad |= 8;
  return bad;
}

// ---- packed format (see DESIGN.md "Third model") ---------------------------------------------------------------------
//   [0]      header: nmsg(8) | aux_svc(3)<<8 | acked[v](2)<<(11+2v) | no_progress_ctr(3)<<20
//   per replica two words (r = 1..R at [1+2(r-1)], [2+2(r-1)]):
//     A: status(2) view(3)<<2 op(2)<<5 commit(2)<<7 lnv(3)<<9 sent_dvc<<12 sent_sv<<13 no_progress<<14
//        peer_op[p](2)<<(15+2(p-1)) | log entry i (1..3): (1 | value<<1) << (25+3(i-1)) | app_state entry i (1..commit): value << (34+2(i-1))
//     B: rep_recv_dvc[r]: the view its members share (3 bits, 0 when the set is empty) | for every source s a 17-bit slot at
//        3+17(s-1): present(1) | last_normal_vn(3)<<1 | op_number(2)<<4 | commit_number(2)<<6 | log bits (as in A)<<8
//        (members have dest = r, type DoViewChange, one per source, one common view: asserted by encode)
//   [1+2R..) bag: type(3) view(3)<<3 dest(3)<<6 source(3)<<9 op(2)<<12 commit(2)<<14 lnv(3)<<16 first_op(2)<<19 count(2)<<21 |
//            entries << 32, one byte per op number (byte on-1): 1 | value<<3     (Prepare: byte 0 = the entry)
int words_per_replica(const Params&) { return 2; }
int fixed_words(const Params& P) { return 1 + 2 * P.R; }

static u64 enc_log_bits(const Log& l) {
  u64 w = 0;
  if (l.len() && l.lo != 1) throw RepError("a replica log that does not start at op 1");
  for (int i = l.lo; i <= l.hi; i++) w |= (u64)(1 | (l.v[i] << 1)) << (3 * (i - 1));
  return w;
}
static Log dec_log_bits(u64 bits) {
  Log l;
  l.lo = 1;
  l.hi = 0;
  for (int i = 1; i <= 3; i++) {
    const int e = (int)((bits >> (3 * (i - 1))) & 7);
    if (e & 1) { l.hi = i; l.v[i] = e >> 1; }
  }
  return l;
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
    if ((int)x.app_state.size() != x.commit) throw RepError("Len(rep_app_state) differs from rep_commit_number");
    u64 A = (u64)x.status | ((u64)x.view << 2) | ((u64)x.op << 5) | ((u64)x.commit << 7) | ((u64)x.lnv << 9) |
            ((u64)x.sent_dvc << 12) | ((u64)x.sent_sv << 13) | ((u64)x.no_progress << 14);
    for (int p = 1; p <= P.R; p++) A |= (u64)x.peer_op[p] << (15 + 2 * (p - 1));
    A |= enc_log_bits(x.log) << 25;
    for (size_t i = 0; i < x.app_state.size(); i++) A |= (u64)x.app_state[i] << (34 + 2 * i);
    out.push_back(A);
    u64 B = 0;
    for (const Msg& m : x.recv_dvc) {
      if (m.type != T_DVC || m.dest != r || m.source < 1 || m.source > P.R) throw RepError("rep_recv_dvc member that is not a DoViewChange to this replica");
      if (m.op != m.log.len()) throw RepError("DoViewChange whose op_number differs from Len(log)");
      if ((B & 7) && (int)(B & 7) != m.view) throw RepError("rep_recv_dvc members of two views");
      const int sh = 3 + 17 * (m.source - 1);
      if ((B >> sh) & 1) throw RepError("two rep_recv_dvc members from one source");
      B |= (u64)m.view;
      B |= ((u64)1 | ((u64)m.lnv << 1) | ((u64)m.op << 4) | ((u64)m.commit << 6) | (enc_log_bits(m.log) << 8)) << sh;
    }
    out.push_back(B);
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
    const u64 A = rec[1 + 2 * (r - 1)], B = rec[2 + 2 * (r - 1)];
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
    x.log = dec_log_bits((A >> 25) & 0x1FF);
    for (int i = 0; i < x.commit; i++) x.app_state.push_back((int)((A >> (34 + 2 * i)) & 3));
    for (int src = 1; src <= P.R; src++) {
      const u64 slot = (B >> (3 + 17 * (src - 1))) & 0x1FFFF;
      if (!(slot & 1)) continue;
      Msg m;
      m.type = T_DVC;
      m.view = (int)(B & 7);
      m.dest = r;
      m.source = src;
      m.lnv = (int)((slot >> 1) & 7);
      m.op = (int)((slot >> 4) & 3);
      m.commit = (int)((slot >> 6) & 3);
      m.log = dec_log_bits((slot >> 8) & 0x1FF);
      x.recv_dvc.push_back(m);
    }
    std::sort(x.recv_dvc.begin(), x.recv_dvc.end());
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
      if (m.log.hi < m.log.lo) { m.log.lo = 1; m.log.hi = 0; }
    }
    s.messages.push_back(std::make_pair(m, count));
  }
  std::sort(s.messages.begin(), s.messages.end(),
            [](const std::pair<Msg, int>& a, const std::pair<Msg, int>& b) { return a.first < b.first; });
  if (nwords) *nwords = fixed_words(P) + nmsg;
  return s;
}

// ---- fingerprint of the VIEW (VRAS.tla:102; VRAS.cfg:24): everything but aux_svc, aux_client_acked, aux_restart ----------
// The same function family as the other models' (version 2): one salted term per replica word (two words here), one term per
// bag entry.  no_progress_ctr, rep_rec_number and rep_rec_recv are in the view but constant; the header word is not hashed.
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

Fp fingerprint(const Params& P, const State& s) {
  std::vector<u64> rec;
  encode(P, s, rec);
  u64 sum = 0;
  for (int r = 1; r <= P.R; r++)
    for (int k = 0; k < 2; k++) sum += fmix64(rec[1 + 2 * (r - 1) + k] ^ (salt_word(r, k) ^ g_fp_seed));
  for (size_t j = fixed_words(P); j < rec.size(); j++) sum += fmix64(rec[j] ^ (SALT_MSG ^ g_fp_seed));
  Fp f;
  f.fp = sum ? sum : 1;
  f.auxkey = (u32)s.aux_svc;
  for (int v = 0; v < P.n; v++) f.auxkey |= (u32)s.acked[v] << (3 + 2 * v);
  f.argmin = 0;
  return f;
}

}  // namespace vras_oracle

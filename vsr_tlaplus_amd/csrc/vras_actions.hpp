// vras_actions.hpp — the guarded-update action table of the THIRD model, lowered onto its packed record:
//   /root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.tla   (cited as VRAS.tla:NNN)
// under VR_APP_STATE.cfg (VIEW view, no SYMMETRY, NoProgressChangeLimit = 0).  SURVEY.md §8(f) rank 2 ("then
// 04-application-state").  Same role as vsr_actions.hpp / vrst_actions.hpp (Tool.getNextStates for one model); the kernels,
// seen-set, frontier and trace machinery are shared, selected at compile time by the model id of the kernel instantiation
// (vsr_kernels.hpp, ModelOps<2>).
//
// What differs from VR_STATE_TRANSFER.tla for the lowering: rep_app_state (VRAS.tla:74) — the executed operations, appended by
// MaybeExecuteOps (:277-283) wherever the commit number rises (the commit number never falls any more, and Len(rep_app_state) =
// rep_commit_number: the values live in the replica's A word); rep_recv_dvc (:82) — the DoViewChange records a replica counts are an
// explicit set now (a second replica word, one slot per source), no longer the bag keys with delivery count 0; one guard more on
// ReceiveMatchingSVC (:602); the invariant NoAppStateDivergence (:852-858).
//
// Record (device layout), 64-bit words:
//   [0]        header: nmsg(8) | aux_svc(3)<<8 | acked[v](2)<<(11+2v) | no_progress_ctr(3)<<20
//   per replica r two words at [1+2(r-1)], [2+2(r-1)]:
//     A: status(2) view(3)<<2 op(2)<<5 commit(2)<<7 last_normal_view(3)<<9 sent_dvc<<12 sent_sv<<13 no_progress<<14
//        peer_op[p](2)<<(15+2(p-1)) | log entry i = (1 | value<<1) << (25+3(i-1)) | app_state entry i (1..commit) = value << (34+2(i-1))
//     B: rep_recv_dvc[r]: the view its members share (3 bits; 0 = empty set) | per source s a 17-bit slot at 3+17(s-1):
//        present(1) | last_normal_vn(3)<<1 | op_number(2)<<4 | commit_number(2)<<6 | log bits (as in A)<<8
//   [1+2R]     H[0] (view hash; one permutation: the identity)
//   [2+2R ..)  bag: the message word of vsr_model.hpp (type, view, dest, source, op, commit, lnv, first_op, count | entries<<32),
//              entry byte = 1 | value<<3, dest 7 = AnyDest
// Ordinals: as in vrst_actions.hpp.  ReplicaCount <= 3 (three slots fit the B word).
#pragma once
#include "vrst_actions.hpp"

namespace vsr {
namespace vras {

using vrst::ANYDEST;
using vrst::ST2_NORMAL;
using vrst::ST2_STATETRANSFER;
using vrst::ST2_VIEWCHANGE;
using vrst::b_log;
using vrst::b_noprog;
using vrst::b_peer;
using vrst::b_set_log;
using vrst::b_set_peer;
using vrst::blog_entry;
using vrst::blog_len;
using vrst::blog_to_bytes;
using vrst::bytes_to_blog;

enum { A_PrimaryExecuteOp = A_ExecuteOp };                                                            // VRAS.tla:420: the same place in Next (:811-831)

VSR_HD int c_ia(int r) { return 1 + 2 * (r - 1); }                                                    // index of replica r's A word
VSR_HD int c_app(u64 A, int i) { return (int)((A >> (34 + 2 * (i - 1))) & 3); }                       // rep_app_state[r][i] (value index)
VSR_HD int d_view(u64 B) { return (int)(B & 7); }
VSR_HD u32 d_slot(u64 B, int s) { return (u32)((B >> (3 + 17 * (s - 1))) & 0x1FFFF); }
VSR_HD u32 d_make_slot(int lnv, int op, int commit, u32 logbits) {
  return 1u | ((u32)lnv << 1) | ((u32)op << 4) | ((u32)commit << 6) | (logbits << 8);
}
// rep_recv_dvc[r] \union {DoViewChange from `src` of view `view`}; a member of another view or a second, different record from
// one source cannot be expressed (and cannot happen: the set is emptied whenever the replica's view changes — ResetVcVars,
// VRAS.tla:255-258 — and a replica sends one DoViewChange per view, :624)
VSR_HD u64 d_add(u64 B, int view, int src, u32 slot, int* err) {
  if (d_view(B) != 0 && d_view(B) != view) { *err = ERR_REP_I1; return B; }
  const u32 old = d_slot(B, src);
  if ((old & 1) && old != slot) { *err = ERR_REP_I2; return B; }
  return (B & ~(u64)7) | (u64)view | ((u64)slot << (3 + 17 * (src - 1)));
}

// MaybeExecuteOps(r, log, old_commit, new_commit), VRAS.tla:277-283 with AppendOps :270-275: operations old_commit+1 .. new_commit
// of `log` are appended to the application state and the commit number rises; log[op] outside the log is a TLC evaluation error
VSR_HD u64 maybe_execute_ops(u64 A, u32 logbits, int old_commit, int new_commit, int* err) {
  if (!(new_commit > old_commit)) return A;                      // :278, :283
  for (int op = old_commit + 1; op <= new_commit; op++) {        // :271-275
    const int e = op <= 3 ? blog_entry(logbits, op) : 0;
    if (!(e & 1)) { *err = ERR_EVAL_DOMAIN; return A; }
    A = a_set(A, 34 + 2 * (op - 1), 2, e >> 1);
  }
  return a_set_commit(A, new_commit);                            // :281
}

// The action table.  D.action is set before the guards return (also in GUARD_ONLY mode), as in vrst::gen.
template <bool GUARD_ONLY, typename PTR>
VSR_HD bool gen(const Model& M, PTR rec, int ord, Delta& D) {
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  PTR bag = rec + M.fixed;
  const vrst::Ord2 o = vrst::ord_decode2(M, ord);
  int r = o.r;
  u64 mw = 0;
  if (o.group == 5) {
    if (o.j >= nmsg) return false;
    mw = bag[o.j];
    if (m_count(mw) == 0) return false;                          // ReceivableMsg: messages[m] > 0          VRAS.tla:223
    const int dest = m_dest(mw);
    if (o.k == 0) {                                              //   m.dest = r                            :220
      if (dest == ANYDEST || dest < 1 || dest > M.R) return false;
      r = dest;
    } else {                                                     //   m.dest = AnyDest /\ m.source # r      :221-222
      if (dest != ANYDEST || o.k == m_source(mw)) return false;
      r = o.k;
    }
  }
  const u64 A = rec[c_ia(r)], B = rec[c_ia(r) + 1];
  const int view = a_view(A), status = a_status(A), op = a_op(A), commit = a_commit(A);
  const bool prim = primary_of(M, view) == r;
  if (b_noprog(A)) return false;                                 // CanProgress(r)                          :263
  D.action = 0;
  if (!GUARD_ONLY) {
    D.hdr = hdr;
    D.r = r;
    D.used = 0;
    D.clear_pj();
    D.err = 0;
    D.rep[0] = A;
    D.rep[1] = B;
    D.rep[2] = D.rep[3] = 0;
  }
  u64& nA = D.rep[0];
  u64& nB = D.rep[1];

  switch (o.group) {
    case 0: {  // ---- TimerSendSVC (VRAS.tla:551-565)
      D.action = A_TimerSendSVC;
      if (!(hdr_aux_svc(hdr) < M.L)) return false;               // :553
      if (prim && status == ST2_NORMAL) return false;            // ~IsNormalPrimary(r) :556
      if (GUARD_ONLY) return true;
      if (view + 1 > 7) { D.err = ERR_REP_RANGE; return true; }
      nA = a_set_view(nA, view + 1);                             // :558
      nA = a_set_status(nA, ST2_VIEWCHANGE);                     // :559
      nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);              // ResetVcVars(r, {}) :560
      nB = 0;
      D.hdr = (hdr & ~((u64)7 << 8)) | ((u64)(hdr_aux_svc(hdr) + 1) << 8);   // :561
      bag_broadcast(M, bag, nmsg, D, m_make(T_SVC, view + 1, 0, r, 0, 0, 0, 0, 0), r);   // :562
      break;
    }
    case 1: {  // ---- SendDVC (VRAS.tla:619-647)
      D.action = A_SendDVC;
      if (status != ST2_VIEWCHANGE) return false;                // :623
      if (a_sent_dvc(A)) return false;                           // :624
      int q = 0;                                                 // :625-629: SVCs of this view addressed to r that were received
      for (int j = 0; j < nmsg; j++) {
        const u64 w = bag[j];
        q += (m_type(w) == T_SVC && m_dest(w) == r && m_view(w) == view && m_count(w) == 0) ? 1 : 0;
      }
      if (!(q >= M.R / 2)) return false;
      if (GUARD_ONLY) return true;
      nA = a_set_sent_dvc(nA, 1);                                // :631
      const int p = primary_of(M, view);
      const u32 lg = b_log(A);
      const u64 key = m_make(T_DVC, view, p, r, op, commit, a_lnv(A), 0, blog_to_bytes(lg));   // :632-639
      if (p == r) {
        vrst::bag_send_cnt(M, bag, nmsg, D, key, 0);             // SendAsReceived :640-641
        nB = d_add(B, view, r, d_make_slot(a_lnv(A), op, commit, lg), &D.err);   // :642
      } else {
        vrst::bag_send_cnt(M, bag, nmsg, D, key, 1);             // Send :643-645
      }
      break;
    }
    case 2: {  // ---- SendSV (VRAS.tla:726-754)
      D.action = A_SendSV;
      if (status != ST2_VIEWCHANGE) return false;                // :730
      if (a_sent_sv(A)) return false;                            // :731
      // ValidDvc (:700-701): a member of rep_recv_dvc[r] of the replica's view; HighestLog (:703-711): CHOOSE among the valid ones
      // maximal in (last_normal_vn, op_number), first in TLC's value order = smallest (commit_number, source) [TLC-RECALLED, as in
      // vrst_actions.hpp]; HighestCommitNumber :718-724
      int q = 0, best_lnv = -1, best_op = -1, best_commit = 0, best_src = 0, max_commit = -1;
      u32 best_log = 0;
      if (d_view(B) == view)
        for (int s2 = 1; s2 <= 3; s2++) {
          if (s2 > M.R) break;
          const u32 sl = d_slot(B, s2);
          if (!(sl & 1)) continue;
          q++;
          const int l = (int)((sl >> 1) & 7), o2 = (int)((sl >> 4) & 3), c2 = (int)((sl >> 6) & 3);
          if (c2 > max_commit) max_commit = c2;
          const bool better = best_src == 0 || l > best_lnv || (l == best_lnv && o2 > best_op) ||
                              (l == best_lnv && o2 == best_op && (c2 < best_commit || (c2 == best_commit && s2 < best_src)));
          if (better) { best_lnv = l; best_op = o2; best_commit = c2; best_src = s2; best_log = (sl >> 8) & 0x1FF; }
        }
      if (!(q >= M.R / 2 + 1)) return false;                     // :732
      if (GUARD_ONLY) return true;
      const int new_on = blog_len(best_log);                     // :713-716
      nA = a_set_status(nA, ST2_NORMAL);                         // :738
      nA = b_set_log(nA, best_log);                              // :739
      nA = maybe_execute_ops(nA, best_log, commit, max_commit, &D.err);   // :740
      nA = a_set_op(nA, new_on);                                 // :741
      for (int p = 1; p <= M.R; p++) nA = b_set_peer(nA, p, 0);  // :742
      nA = a_set_sent_sv(nA, 1);                                 // :743
      nB = 0;                                                    // :744
      nA = a_set_lnv(nA, view);                                  // :745
      bag_broadcast(M, bag, nmsg, D, m_make(T_SV, view, 0, r, new_on, max_commit, 0, 0, blog_to_bytes(best_log)), r);   // :746-752
      break;
    }
    case 3: {  // ---- PrimaryExecuteOp (VRAS.tla:420-435)
      D.action = A_PrimaryExecuteOp;
      if (!(prim && status == ST2_NORMAL)) return false;         // :424
      if (!(commit < op)) return false;                          // :425
      int q = 0;                                                 // IsCommitted :415-418
      for (int p = 1; p <= M.R; p++) q += b_peer(A, p) >= commit + 1 ? 1 : 0;
      if (!(q >= M.R / 2)) return false;                         // :426
      if (GUARD_ONLY) return true;
      const int e = blog_entry(b_log(A), commit + 1);            // :429
      if (!(e & 1)) { D.err = ERR_EVAL_DOMAIN; return true; }
      nA = maybe_execute_ops(nA, b_log(A), commit, commit + 1, &D.err);   // :431
      if (hdr_acked(hdr, e >> 1) == 0) { D.err = ERR_EVAL_DOMAIN; return true; }
      D.hdr = hdr_set_acked(hdr, e >> 1, 2);                     // :432
      break;
    }
    case 4: {  // ---- ReceiveClientRequest (VRAS.tla:328-349)
      D.action = A_ReceiveClientRequest;
      if (!(prim && status == ST2_NORMAL)) return false;         // :332
      if (hdr_acked(hdr, o.v) != 0) return false;                // :333
      if (GUARD_ONLY) return true;
      const u32 lg = b_log(A);
      const int opn = blog_len(lg) + 1;                          // :335
      if (opn > 3) { D.err = ERR_REP_RANGE; return true; }
      nA = b_set_log(nA, lg | ((1u | ((u32)o.v << 1)) << (3 * (opn - 1))));   // :338
      nA = a_set_op(nA, opn);                                    // :339
      bag_broadcast(M, bag, nmsg, D, m_make(T_PREPARE, view, 0, r, opn, commit, 0, 0, 1u | ((u32)o.v << 3)), r);   // :340-346
      D.hdr = hdr_set_acked(hdr, o.v, 1);                        // :347
      break;
    }
    default: {  // ---- message-bound actions
      const int mt = m_type(mw), mview = m_view(mw), msrc = m_source(mw), mop = m_op(mw), mcommit = m_commit(mw);
      switch (mt) {
        case T_SVC:
        case T_DVC: {
          if (mview > view) {  // ---- ReceiveHigherSVC (VRAS.tla:575-587) / ReceiveHigherDVC (:656-668)
            D.action = mt == T_SVC ? A_ReceiveHigherSVC : A_ReceiveHigherDVC;
            if (GUARD_ONLY) return true;
            nA = a_set_view(nA, mview);
            nA = a_set_status(nA, ST2_VIEWCHANGE);
            nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);        // ResetVcVars(r, {}) :584 / ResetVcVars(r, {m}) :665
            nB = mt == T_SVC ? (u64)0
                             : d_add(0, mview, msrc, d_make_slot(m_lnv(mw), mop, mcommit, bytes_to_blog(m_lg(mw) & 0xFFFFFF)), &D.err);
            bag_discard(D, o.j, mw);                             // DiscardAndBroadcast :205-211
            bag_broadcast(M, bag, nmsg, D, m_make(T_SVC, mview, 0, r, 0, 0, 0, 0, 0), r);
          } else if (mview == view && status == ST2_VIEWCHANGE) {
            if (mt == T_SVC) {  // ---- ReceiveMatchingSVC (:595-606)
              D.action = A_ReceiveMatchingSVC;
              if (a_sent_dvc(A)) return false;                   // :602 "reduce state space"
              if (GUARD_ONLY) return true;
              bag_discard(D, o.j, mw);                           // :604: the key stays with count 0, SendDVC counts those (:625-629)
            } else {            // ---- ReceiveMatchingDVC (:676-687)
              D.action = A_ReceiveMatchingDVC;
              if (GUARD_ONLY) return true;
              bag_discard(D, o.j, mw);                           // :684
              nB = d_add(B, mview, msrc, d_make_slot(m_lnv(mw), mop, mcommit, bytes_to_blog(m_lg(mw) & 0xFFFFFF)), &D.err);   // :685
            }
          } else {
            return false;
          }
          break;
        }
        case T_SV: {  // ---- ReceiveSV (VRAS.tla:765-788)
          D.action = A_ReceiveSV;
          if (!((mview == view && status == ST2_VIEWCHANGE) || mview > view)) return false;   // :770-772
          if (GUARD_ONLY) return true;
          const u32 ml = bytes_to_blog(m_lg(mw) & 0xFFFFFF);
          nA = a_set_status(nA, ST2_NORMAL);                     // :774
          nA = a_set_view(nA, mview);                            // :775
          nA = b_set_log(nA, ml);                                // :776
          nA = maybe_execute_ops(nA, ml, commit, mcommit, &D.err);   // :777
          nA = a_set_op(nA, mop);                                // :778
          nA = a_set_lnv(nA, mview);                             // :779
          nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);          // ResetVcVars(r, {}) :780
          nB = 0;
          bag_discard(D, o.j, mw);
          if (commit < mop)                                      // :781 (the replica's OLD commit number)
            vrst::bag_send_cnt(M, bag, nmsg, D, m_make(T_PREPAREOK, mview, primary_of(M, mview), r, mop, 0, 0, 0, 0), 1);   // :782-786
          break;
        }
        case T_PREPARE: {
          if (prim || status != ST2_NORMAL) return false;        // IsNormalBackup(r) :365 / :464
          if (mview == view && mop == op + 1) {  // ---- ReceivePrepareMsg (VRAS.tla:360-380)
            D.action = A_ReceivePrepareMsg;
            if (GUARD_ONLY) return true;
            const u32 lg = b_log(A);
            const int pos = blog_len(lg) + 1;                    // Append :369
            if (pos > 3) { D.err = ERR_REP_RANGE; return true; }
            const u32 v = (m_lg(mw) >> 3) & 3;
            const u32 nl = lg | ((1u | (v << 1)) << (3 * (pos - 1)));
            nA = b_set_log(nA, nl);                              // :371
            nA = a_set_op(nA, mop);                              // :372
            nA = maybe_execute_ops(nA, nl, commit, mcommit, &D.err);   // :373
            bag_discard(D, o.j, mw);
            vrst::bag_send_cnt(M, bag, nmsg, D, m_make(T_PREPAREOK, view, msrc, r, mop, 0, 0, 0, 0), 1);   // :374-378
          } else if (mview > view && mop > op + 1) {  // ---- SendGetState (VRAS.tla:461-476); the Prepare stays in the bag
            D.action = A_SendGetState;
            const u64 gs = m_make(T_GETSTATE, mview, ANYDEST, r, commit, 0, 0, 0, 0);   // :469-473
            if (bag_has_key(bag, nmsg, gs)) return false;        // SendOnce :195-197
            if (GUARD_ONLY) return true;
            nA = a_set_status(nA, ST2_STATETRANSFER);            // :468
            vrst::bag_send_cnt(M, bag, nmsg, D, gs, 1);
          } else {
            return false;
          }
          break;
        }
        case T_PREPAREOK: {  // ---- ReceivePrepareOkMsg (VRAS.tla:393-405)
          D.action = A_ReceivePrepareOkMsg;
          if (!(prim && status == ST2_NORMAL)) return false;     // :398
          if (mview != view) return false;                       // :399
          if (!(mop > b_peer(A, msrc))) return false;            // :400
          if (GUARD_ONLY) return true;
          nA = b_set_peer(nA, msrc, mop);                        // :402
          bag_discard(D, o.j, mw);                               // :403
          break;
        }
        case T_GETSTATE: {  // ---- ReceiveGetState (VRAS.tla:490-507)
          D.action = A_ReceiveGetState;
          if (view != mview) return false;                       // :494
          if (status != ST2_NORMAL) return false;                // :495
          if (!(op > mop)) return false;                         // :496
          if (GUARD_ONLY) return true;
          const u32 lg = b_log(A);
          const u32 bytes = blog_to_bytes(lg);
          const int len = blog_len(lg);
          u32 part = 0;                                          // LogSuffix(r, rep_log[r], m.op_number) :265-268: entries mop+1 .. Len(log)
          for (int on = mop + 1; on <= len; on++) part |= ((bytes >> (8 * (on - 1))) & 0xFF) << (8 * (on - 1));
          bag_discard(D, o.j, mw);
          vrst::bag_send_cnt(M, bag, nmsg, D, m_make(T_NEWSTATE, view, msrc, r, op, commit, 0, mop + 1, part), 1);   // :497-505
          break;
        }
        case T_NEWSTATE: {  // ---- ReceiveNewState (VRAS.tla:516-537)
          D.action = A_ReceiveNewState;
          if (status != ST2_STATETRANSFER) return false;         // :519
          if (!(mview > view)) return false;                     // :522
          if (GUARD_ONLY) return true;
          const u32 own = b_log(A), ml = bytes_to_blog(m_lg(mw) & 0xFFFFFF);
          const int first = m_first_op(mw);
          u32 nl = 0;                                            // :524-527
          for (int on = 1; on <= mop; on++) {
            const u32 e = on < first ? (u32)blog_entry(own, on) : (u32)blog_entry(ml, on);
            if (!(e & 1)) { D.err = ERR_EVAL_DOMAIN; return true; }
            nl |= e << (3 * (on - 1));
          }
          nA = a_set_status(nA, ST2_NORMAL);                     // :529
          nA = a_set_view(nA, mview);                            // :530
          nA = a_set_lnv(nA, mview);                             // :531
          nA = b_set_log(nA, nl);                                // :532
          nA = maybe_execute_ops(nA, nl, commit, mcommit, &D.err);   // :533
          nA = a_set_op(nA, mop);                                // :534
          bag_discard(D, o.j, mw);                               // :535
          break;
        }
        default:
          return false;
      }
      break;
    }
  }
  if (!GUARD_ONLY) {
    int na = 0;
#pragma unroll
    for (int k = 0; k < VSR_NSLOT; k++) na += (((D.used >> k) & 1) && D.pj(k) < 0) ? 1 : 0;
    if (nmsg + na > M.max_bag) D.err = D.err ? D.err : ERR_REP_BAG;
    D.hdr = hdr_set_nmsg(D.hdr, nmsg + na);
  }
  return true;
}

// Guards per enumeration slot, as vrst::guard_slot
template <typename PTR>
VSR_HD u32 guard_slot(const Model& M, PTR rec, int slot, int* kind) {
  Delta D;
  *kind = 0;
  if (slot < M.m0) {
    const bool en = vras::gen<true>(M, rec, slot, D);
    *kind = D.action;
    return en ? 1u : 0u;
  }
  const int j = slot - M.m0;
  if (j >= hdr_nmsg(rec[0])) return 0;
  u32 mask = 0;
  const int base = M.m0 + j * (M.R + 1);
  if (m_dest(rec[M.fixed + j]) != ANYDEST) {
    if (vras::gen<true>(M, rec, base, D)) { mask = 1u; *kind = D.action; }
    return mask;
  }
  for (int k = 1; k <= M.R; k++)
    if (vras::gen<true>(M, rec, base + k, D)) { mask |= 1u << k; *kind = D.action; }
  return mask;
}

// view hashes: one salted term per replica word (two words), one per bag entry; no value permutation (no SYMMETRY)
template <typename PTR>
VSR_HD void hash_full(const Model& M, PTR rec, u64* H) {
  const int nmsg = hdr_nmsg(rec[0]);
  u64 sum = 0;
  for (int r = 1; r <= M.R; r++) sum += fmix64(rec[c_ia(r)] ^ (salt_word<0>(r) ^ M.fp_seed)) + fmix64(rec[c_ia(r) + 1] ^ (salt_word<1>(r) ^ M.fp_seed));
  for (int j = 0; j < nmsg; j++) sum += fmix64(rec[M.fixed + j] ^ (SALT_MSG ^ M.fp_seed));
  H[0] = sum;
}
template <typename PTR>
VSR_HD void hash_child(const Model& M, PTR rec, const Delta& D, u64* Hc) {
  u64 h = rec[M.h0];
  const u64 oldA = rec[c_ia(D.r)], oldB = rec[c_ia(D.r) + 1];
  if (oldA != D.rep[0]) h += fmix64(D.rep[0] ^ (salt_word<0>(D.r) ^ M.fp_seed)) - fmix64(oldA ^ (salt_word<0>(D.r) ^ M.fp_seed));
  if (oldB != D.rep[1]) h += fmix64(D.rep[1] ^ (salt_word<1>(D.r) ^ M.fp_seed)) - fmix64(oldB ^ (salt_word<1>(D.r) ^ M.fp_seed));
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if ((D.used >> k) & 1) {
      h += fmix64(D.pnew[k] ^ (SALT_MSG ^ M.fp_seed));
      if (D.pj(k) >= 0) h -= fmix64(rec[M.fixed + D.pj(k)] ^ (SALT_MSG ^ M.fp_seed));
    }
  Hc[0] = h;
}

// Invariants on the child (VRAS.tla:840-894): mask of VIOLATED ones.  bit0 AcknowledgedWriteNotLost, bit1
// AcknowledgedWritesExistOnMajority, bit2 NoLogDivergence, bit3 CommitNumberNeverHigherThanOpNumber, bit4 NoAppStateDivergence.
// NoLogDivergence / NoAppStateDivergence read rep_log[r][op] for op <= commit: beyond the log that is a TLC evaluation error; it is
// reported as a violation of the bit here (MaybeExecuteOps has raised the evaluation error in the action already).
template <typename PTR>
VSR_HD int check_invariants_child(const Model& M, PTR rec, const Delta& D) {
  int bad = 0;
  u64 Aw[4];
#pragma unroll
  for (int r = 1; r <= 3; r++) Aw[r] = r <= M.R ? (r == D.r ? D.rep[0] : rec[c_ia(r)]) : 0;
  if (M.inv_mask & 3)
    for (int v = 0; v < M.n; v++) {
      if (hdr_acked(D.hdr, v) != 2) continue;
      int holders = 0;
#pragma unroll
      for (int r = 1; r <= 3; r++) {
        if (r > M.R) break;
        const u32 lg = b_log(Aw[r]);
        bool has = false;                                        // ReplicaHasOp :861-863
        for (int i = 1; i <= 3; i++) {
          const int e = blog_entry(lg, i);
          if ((e & 1) && (e >> 1) == v) has = true;
        }
        holders += has ? 1 : 0;
      }
      if ((M.inv_mask & 1) && holders == 0) bad |= 1;            // :877-882
      if ((M.inv_mask & 2) && !(holders >= M.R / 2 + 1)) bad |= 2;   // :865-871
    }
  if (M.inv_mask & (4 | 16))
    for (int opn = 1; opn <= M.n; opn++)
#pragma unroll
      for (int r1 = 1; r1 <= 3; r1++)
#pragma unroll
        for (int r2 = 1; r2 <= 3; r2++) {
          if (r1 > M.R || r2 > M.R || r2 == r1) continue;
          if (!(opn <= a_commit(Aw[r1]) && opn <= a_commit(Aw[r2]))) continue;   // :843-844 / :855-856
          const int e1 = blog_entry(b_log(Aw[r1]), opn), e2 = blog_entry(b_log(Aw[r2]), opn);
          if ((M.inv_mask & 4) && (!(e1 & 1) || !(e2 & 1) || e1 != e2)) bad |= 4;   // NoLogDivergence :840-845
          if ((M.inv_mask & 16) && c_app(Aw[r1], opn) != c_app(Aw[r2], opn) &&   // NoAppStateDivergence :857
              (!(e1 & 1) || (e1 >> 1) == c_app(Aw[r1], opn)))                      //                       :858
            bad |= 16;
        }
  if (M.inv_mask & 8)                                            // CommitNumberNeverHigherThanOpNumber :892-894
#pragma unroll
    for (int r = 1; r <= 3; r++)
      if (r <= M.R && !(a_commit(Aw[r]) <= a_op(Aw[r]))) bad |= 8;
  return bad;
}

}  // namespace vras
}  // namespace vsr

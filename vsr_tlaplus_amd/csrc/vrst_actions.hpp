// vrst_actions.hpp — the guarded-update action table of the SECOND model, lowered onto its packed record:
//   /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.tla   (cited as VRST.tla:NNN)
// under VR_STATE_TRANSFER.cfg (VIEW view, no SYMMETRY, NoProgressChangeLimit = 0).  SURVEY.md §8(f) rank 2.
// Same role as vsr_actions.hpp (Tool.getNextStates for one model); the kernels, seen-set, frontier and trace machinery are
// shared, selected at compile time by the model id of the kernel instantiation (vsr_kernels.hpp, ModelOps<1>).
//
// What differs from VSR.tla for the lowering: no clients / client table; a log entry is its value (VRST.tla:104-105); no
// rep_svc_recv / rep_dvc_recv — received SVC / DVC messages are the bag keys with delivery count 0 (VRST.tla:594-598, 666-670);
// status StateTransfer (:54); GetState goes to AnyDest and any replica but the sender takes it (:213-218, :441-445).
//
// Record (device layout), 64-bit words:
//   [0]        header: nmsg(8) | aux_svc(3)<<8 | acked[v](2)<<(11+2v) | no_progress_ctr(3)<<20
//   [1..R]     one word per replica: status(2) view(3)<<2 op(2)<<5 commit(2)<<7 last_normal_view(3)<<9 sent_dvc<<12 sent_sv<<13
//              no_progress<<14 peer_op[p](2)<<(15+2(p-1)) | log entry i = (1 | value<<1) << (25+3(i-1))
//   [1+R]      H[0] (view hash; one permutation: the identity)
//   [2+R ..)   bag: the message word of vsr_model.hpp (type, view, dest, source, op, commit, lnv, first_op, count | entries<<32),
//              entry byte = 1 | value<<3, dest 7 = AnyDest
// Ordinals: [0,R) TimerSendSVC(r) | [R,2R) SendDVC(r) | [2R,3R) SendSV(r) | [3R,4R) ExecuteOp(r) | [4R,m0) ReceiveClientRequest(r,v)
//           m0 + j(R+1) + k: bag entry j received by its dest (k = 0) or, for an AnyDest entry, by replica k (1..R)
#pragma once
#include "vsr_actions.hpp"

namespace vsr {
namespace vrst {

enum { ST2_NORMAL = 0, ST2_VIEWCHANGE = 1, ST2_STATETRANSFER = 2 };                                   // VRST.tla:52-54
enum { ANYDEST = 7 };                                                                                 // VRST.tla:66

VSR_HD int b_noprog(u64 A) { return (int)((A >> 14) & 1); }
VSR_HD int b_peer(u64 A, int p) { return (int)((A >> (15 + 2 * (p - 1))) & 3); }
VSR_HD u64 b_set_peer(u64 A, int p, int v) { return a_set(A, 15 + 2 * (p - 1), 2, v); }
VSR_HD u32 b_log(u64 A) { return (u32)((A >> 25) & 0x1FF); }                                         // 3 entries x 3 bits
VSR_HD u64 b_set_log(u64 A, u32 lg) { return (A & ~((u64)0x1FF << 25)) | ((u64)lg << 25); }
VSR_HD int blog_len(u32 lg) { return (int)((lg & 1) + ((lg >> 3) & 1) + ((lg >> 6) & 1)); }
VSR_HD int blog_entry(u32 lg, int opn) { return (int)((lg >> (3 * (opn - 1))) & 7); }                 // 0 = absent, else 1 | value<<1
// replica-log bits <-> message-log bytes (entry byte = 1 | value<<3, byte opn-1)
VSR_HD u32 blog_to_bytes(u32 lg) {
  u32 out = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const u32 e = (lg >> (3 * i)) & 7;
    if (e & 1) out |= (1u | ((e >> 1) << 3)) << (8 * i);
  }
  return out;
}
VSR_HD u32 bytes_to_blog(u32 bytes) {
  u32 out = 0;
#pragma unroll
  for (int i = 0; i < 3; i++) {
    const u32 e = (bytes >> (8 * i)) & 0xFF;
    if (e & 7) out |= (1u | (((e >> 3) & 3) << 1)) << (3 * i);
  }
  return out;
}

struct Ord2 { int group, r, v, j, k; };
VSR_HD Ord2 ord_decode2(const Model& M, int ord) {
  Ord2 o;
  o.group = 5; o.r = 0; o.v = 0; o.j = 0; o.k = 0;
  if (ord < 4 * M.R) {
    o.group = ord / M.R;
    o.r = ord % M.R + 1;
  } else if (ord < M.m0) {
    const int idx = ord - 4 * M.R;
    o.group = 4;
    o.r = idx / M.n + 1;
    o.v = idx % M.n;
  } else {
    const int q = ord - M.m0;
    o.j = q / (M.R + 1);
    o.k = q % (M.R + 1);
  }
  return o;
}

// SendFunc(m, msgs, deliver_count), VRST.tla:165-168: an existing key gets count + 1, a new key starts at deliver_count
template <typename PTR>
VSR_HD void bag_send_cnt(const Model& M, PTR bag, int nmsg, Delta& D, u64 key, int cnt0) {
  (void)M;
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if (((D.used >> k) & 1) && (D.pnew[k] & KEYMASK) == key) {
      int c = m_count(D.pnew[k]) + 1;
      if (c > 3) { D.err = ERR_REP_COUNT; return; }
      D.pnew[k] = m_set_count(D.pnew[k], c);
      return;
    }
  D.used |= 1 << 1;
  D.set_pj(1, -1);
  D.pnew[1] = m_set_count(key, cnt0);
  for (int j = 0; j < nmsg; j++) {
    const u64 w = bag[j];
    if ((w & KEYMASK) == key) {
      int c = m_count(w) + 1;
      if (c > 3) { D.err = ERR_REP_COUNT; c = 3; }
      D.set_pj(1, j);
      D.pnew[1] = m_set_count(w, c);
      return;
    }
  }
}

// The action table.  D.action is set before the guards return (also in GUARD_ONLY mode): the frontier enumeration asks this
// one statement of the guards for both "enabled?" and "which action" (no second statement to keep in step).
template <bool GUARD_ONLY, typename PTR>
VSR_HD bool gen(const Model& M, PTR rec, int ord, Delta& D) {
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  PTR bag = rec + M.fixed;
  const Ord2 o = ord_decode2(M, ord);
  int r = o.r;
  u64 mw = 0;
  if (o.group == 5) {
    if (o.j >= nmsg) return false;
    mw = bag[o.j];
    if (m_count(mw) == 0) return false;                          // ReceivableMsg: messages[m] > 0          VRST.tla:218
    const int dest = m_dest(mw);
    if (o.k == 0) {                                              //   m.dest = r                            :214
      if (dest == ANYDEST || dest < 1 || dest > M.R) return false;
      r = dest;
    } else {                                                     //   m.dest = AnyDest /\ m.source # r      :215-217
      if (dest != ANYDEST || o.k == m_source(mw)) return false;
      r = o.k;
    }
  }
  const u64 A = rec[r];
  const int view = a_view(A), status = a_status(A), op = a_op(A), commit = a_commit(A);
  const bool prim = primary_of(M, view) == r;
  if (b_noprog(A)) return false;                                 // CanProgress(r)                          :257
  D.action = 0;
  if (!GUARD_ONLY) {
    D.hdr = hdr;
    D.r = r;
    D.used = 0;
    D.clear_pj();
    D.err = 0;
    D.rep[0] = A;
    D.rep[1] = D.rep[2] = D.rep[3] = 0;
  }
  u64& nA = D.rep[0];

  switch (o.group) {
    case 0: {  // ---- TimerSendSVC (VRST.tla:522-535)
      D.action = A_TimerSendSVC;
      if (!(hdr_aux_svc(hdr) < M.L)) return false;               // :524
      if (prim && status == ST2_NORMAL) return false;            // ~IsNormalPrimary(r) :527
      if (GUARD_ONLY) return true;
      if (view + 1 > 7) { D.err = ERR_REP_RANGE; return true; }
      nA = a_set_view(nA, view + 1);                             // :529
      nA = a_set_status(nA, ST2_VIEWCHANGE);                     // :530
      nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);              // :531
      D.hdr = (hdr & ~((u64)7 << 8)) | ((u64)(hdr_aux_svc(hdr) + 1) << 8);   // :532
      bag_broadcast(M, bag, nmsg, D, m_make(T_SVC, view + 1, 0, r, 0, 0, 0, 0, 0), r);   // :533
      break;
    }
    case 1: {  // ---- SendDVC (VRST.tla:588-614)
      D.action = A_SendDVC;
      if (status != ST2_VIEWCHANGE) return false;                // :592
      if (a_sent_dvc(A)) return false;                           // :593
      int q = 0;                                                 // :594-598: SVCs of this view addressed to r that were received
      for (int j = 0; j < nmsg; j++) {
        const u64 w = bag[j];
        q += (m_type(w) == T_SVC && m_dest(w) == r && m_view(w) == view && m_count(w) == 0) ? 1 : 0;
      }
      if (!(q >= M.R / 2)) return false;
      if (GUARD_ONLY) return true;
      nA = a_set_sent_dvc(nA, 1);                                // :600
      const int p = primary_of(M, view);
      const u64 key = m_make(T_DVC, view, p, r, op, commit, a_lnv(A), 0, blog_to_bytes(b_log(A)));   // :601-608
      bag_send_cnt(M, bag, nmsg, D, key, p == r ? 0 : 1);        // SendAsReceived :609-610 / Send :611-612
      break;
    }
    case 2: {  // ---- SendSV (VRST.tla:695-721)
      D.action = A_SendSV;
      if (status != ST2_VIEWCHANGE) return false;                // :699
      if (a_sent_sv(A)) return false;                            // :700
      // ValidDvc (:666-670); HighestLog (:672-680): CHOOSE among the valid DVCs maximal in (last_normal_vn, op_number), first in
      // TLC's value order = smallest (commit_number, source) [TLC-RECALLED, as in vsr_actions.hpp]; HighestCommitNumber :687-693
      int q = 0, best_lnv = -1, best_op = -1, best_commit = 0, best_src = 0, max_commit = -1;
      u32 best_log = 0;
      for (int j = 0; j < nmsg; j++) {
        const u64 w = bag[j];
        if (!(m_view(w) == view && m_type(w) == T_DVC && m_dest(w) == r && m_count(w) == 0)) continue;
        q++;
        const int l = m_lnv(w), o2 = m_op(w), c2 = m_commit(w), s2 = m_source(w);
        if (c2 > max_commit) max_commit = c2;
        const bool better = best_src == 0 || l > best_lnv || (l == best_lnv && o2 > best_op) ||
                            (l == best_lnv && o2 == best_op && (c2 < best_commit || (c2 == best_commit && s2 < best_src)));
        if (better) { best_lnv = l; best_op = o2; best_commit = c2; best_src = s2; best_log = m_lg(w) & 0xFFFFFF; }
      }
      if (!(q >= M.R / 2 + 1)) return false;                     // :701
      if (GUARD_ONLY) return true;
      const u32 nl = bytes_to_blog(best_log);
      const int new_on = blog_len(nl);                           // :682-685
      nA = a_set_status(nA, ST2_NORMAL);                         // :707
      nA = b_set_log(nA, nl);                                    // :708
      nA = a_set_op(nA, new_on);                                 // :709
      for (int p = 1; p <= M.R; p++) nA = b_set_peer(nA, p, 0);  // :710
      nA = a_set_commit(nA, max_commit);                         // :711
      nA = a_set_sent_sv(nA, 1);                                 // :712
      nA = a_set_lnv(nA, view);                                  // :713
      bag_broadcast(M, bag, nmsg, D, m_make(T_SV, view, 0, r, new_on, max_commit, 0, 0, best_log), r);   // :714-720
      break;
    }
    case 3: {  // ---- ExecuteOp (VRST.tla:389-405)
      D.action = A_ExecuteOp;
      if (!(prim && status == ST2_NORMAL)) return false;         // :393
      if (!(commit < op)) return false;                          // :394
      int q = 0;                                                 // IsCommitted :384-387
      for (int p = 1; p <= M.R; p++) q += b_peer(A, p) >= commit + 1 ? 1 : 0;
      if (!(q >= M.R / 2)) return false;                         // :395
      if (GUARD_ONLY) return true;
      const int e = blog_entry(b_log(A), commit + 1);            // :398
      if (!(e & 1)) { D.err = ERR_EVAL_DOMAIN; return true; }
      nA = a_set_commit(nA, commit + 1);                         // :400
      if (hdr_acked(hdr, e >> 1) == 0) { D.err = ERR_EVAL_DOMAIN; return true; }
      D.hdr = hdr_set_acked(hdr, e >> 1, 2);                     // :401
      break;
    }
    case 4: {  // ---- ReceiveClientRequest (VRST.tla:298-318)
      D.action = A_ReceiveClientRequest;
      if (!(prim && status == ST2_NORMAL)) return false;         // :302
      if (hdr_acked(hdr, o.v) != 0) return false;                // :303
      if (GUARD_ONLY) return true;
      const u32 lg = b_log(A);
      const int opn = blog_len(lg) + 1;                          // :305
      if (opn > 3) { D.err = ERR_REP_RANGE; return true; }
      nA = b_set_log(nA, lg | ((1u | ((u32)o.v << 1)) << (3 * (opn - 1))));   // :308
      nA = a_set_op(nA, opn);                                    // :309
      bag_broadcast(M, bag, nmsg, D, m_make(T_PREPARE, view, 0, r, opn, commit, 0, 0, 1u | ((u32)o.v << 3)), r);   // :310-316
      D.hdr = hdr_set_acked(hdr, o.v, 1);                        // :317
      break;
    }
    default: {  // ---- message-bound actions
      const int mt = m_type(mw), mview = m_view(mw), msrc = m_source(mw), mop = m_op(mw), mcommit = m_commit(mw);
      switch (mt) {
        case T_SVC:
        case T_DVC: {
          if (mview > view) {  // ---- ReceiveHigherSVC (VRST.tla:545-556) / ReceiveHigherDVC (:623-634)
            D.action = mt == T_SVC ? A_ReceiveHigherSVC : A_ReceiveHigherDVC;
            if (GUARD_ONLY) return true;
            nA = a_set_view(nA, mview);
            nA = a_set_status(nA, ST2_VIEWCHANGE);
            nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);
            bag_discard(D, o.j, mw);                             // DiscardAndBroadcast :200-206
            bag_broadcast(M, bag, nmsg, D, m_make(T_SVC, mview, 0, r, 0, 0, 0, 0, 0), r);
          } else if (mview == view && status == ST2_VIEWCHANGE) {   // ---- ReceiveMatchingSVC (:565-574) / ReceiveMatchingDVC (:643-652)
            D.action = mt == T_SVC ? A_ReceiveMatchingSVC : A_ReceiveMatchingDVC;
            if (GUARD_ONLY) return true;
            bag_discard(D, o.j, mw);                             // the key stays with count 0: that IS the received record
          } else {
            return false;
          }
          break;
        }
        case T_SV: {  // ---- ReceiveSV (VRST.tla:733-756)
          D.action = A_ReceiveSV;
          if (!((mview == view && status == ST2_VIEWCHANGE) || mview > view)) return false;   // :738-740
          if (GUARD_ONLY) return true;
          nA = a_set_status(nA, ST2_NORMAL);                     // :742
          nA = a_set_view(nA, mview);                            // :743
          nA = b_set_log(nA, bytes_to_blog(m_lg(mw) & 0xFFFFFF));   // :744
          nA = a_set_op(nA, mop);                                // :745
          nA = a_set_commit(nA, mcommit);                        // :746
          nA = a_set_lnv(nA, mview);                             // :747
          nA = a_set_sent_sv(a_set_sent_dvc(nA, 0), 0);          // :748
          bag_discard(D, o.j, mw);
          if (commit < mop)                                      // :749 (the replica's OLD commit number)
            bag_send_cnt(M, bag, nmsg, D, m_make(T_PREPAREOK, mview, primary_of(M, mview), r, mop, 0, 0, 0, 0), 1);   // :750-754
          break;
        }
        case T_PREPARE: {
          if (prim || status != ST2_NORMAL) return false;        // IsNormalBackup(r) :334 / :435
          if (mview == view && mop == op + 1) {  // ---- ReceivePrepareMsg (VRST.tla:330-349)
            D.action = A_ReceivePrepareMsg;
            if (GUARD_ONLY) return true;
            const u32 lg = b_log(A);
            const int pos = blog_len(lg) + 1;                    // Append :339
            if (pos > 3) { D.err = ERR_REP_RANGE; return true; }
            const u32 v = (m_lg(mw) >> 3) & 3;
            nA = b_set_log(nA, lg | ((1u | (v << 1)) << (3 * (pos - 1))));
            nA = a_set_op(nA, mop);                              // :340
            nA = a_set_commit(nA, mcommit);                      // :341
            bag_discard(D, o.j, mw);
            bag_send_cnt(M, bag, nmsg, D, m_make(T_PREPAREOK, view, msrc, r, mop, 0, 0, 0, 0), 1);   // :342-346
          } else if (mview > view && mop > op + 1) {  // ---- SendGetState (VRST.tla:431-447); the Prepare stays in the bag
            D.action = A_SendGetState;
            const u64 gs = m_make(T_GETSTATE, mview, ANYDEST, r, commit, 0, 0, 0, 0);   // :441-445
            if (bag_has_key(bag, nmsg, gs)) return false;        // SendOnce :190-192
            if (GUARD_ONLY) return true;
            nA = a_set_status(nA, ST2_STATETRANSFER);            // :440
            bag_send_cnt(M, bag, nmsg, D, gs, 1);
          } else {
            return false;
          }
          break;
        }
        case T_PREPAREOK: {  // ---- ReceivePrepareOkMsg (VRST.tla:361-372)
          D.action = A_ReceivePrepareOkMsg;
          if (!(prim && status == ST2_NORMAL)) return false;     // :365
          if (mview != view) return false;                       // :367
          if (!(mop > b_peer(A, msrc))) return false;            // :368
          if (GUARD_ONLY) return true;
          nA = b_set_peer(nA, msrc, mop);                        // :370
          bag_discard(D, o.j, mw);                               // :371
          break;
        }
        case T_GETSTATE: {  // ---- ReceiveGetState (VRST.tla:460-478)
          D.action = A_ReceiveGetState;
          if (status != ST2_NORMAL) return false;                // :464
          if (view != mview) return false;                       // :466
          if (!(op > mop)) return false;                         // :467
          if (GUARD_ONLY) return true;
          const u32 bytes = blog_to_bytes(b_log(A));
          u32 part = 0;                                          // entries mop+1 .. op    :472-473
          for (int on = mop + 1; on <= op; on++) {
            const u32 e = (bytes >> (8 * (on - 1))) & 0xFF;
            if (!e) { D.err = ERR_EVAL_DOMAIN; return true; }
            part |= e << (8 * (on - 1));
          }
          bag_discard(D, o.j, mw);
          bag_send_cnt(M, bag, nmsg, D, m_make(T_NEWSTATE, view, msrc, r, op, commit, 0, mop + 1, part), 1);   // :470-477
          break;
        }
        case T_NEWSTATE: {  // ---- ReceiveNewState (VRST.tla:488-508)
          D.action = A_ReceiveNewState;
          if (status != ST2_STATETRANSFER) return false;         // :491
          if (!(mview > view)) return false;                     // :494
          if (GUARD_ONLY) return true;
          const u32 own = b_log(A), ml = bytes_to_blog(m_lg(mw) & 0xFFFFFF);
          const int first = m_first_op(mw);
          u32 nl = 0;                                            // :499-503
          for (int on = 1; on <= mop; on++) {
            const u32 e = on < first ? (u32)blog_entry(own, on) : (u32)blog_entry(ml, on);
            if (!(e & 1)) { D.err = ERR_EVAL_DOMAIN; return true; }
            nl |= e << (3 * (on - 1));
          }
          nA = a_set_status(nA, ST2_NORMAL);                     // :496
          nA = a_set_view(nA, mview);                            // :497
          nA = a_set_lnv(nA, mview);                             // :498
          nA = b_set_log(nA, nl);
          nA = a_set_op(nA, mop);                                // :504
          nA = a_set_commit(nA, mcommit);                        // :505
          bag_discard(D, o.j, mw);                               // :506
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

// Guards per enumeration slot (see guard_slot_pre of vsr_actions.hpp): bit k of the result <=> ordinal base + k is enabled;
// *kind0 = action id of the lowest enabled bit's... every enabled bit of one slot has the same action (bit 0: by message type;
// bits 1..R only exist for AnyDest entries, i.e. ReceiveGetState), so one id serves the slot.
template <typename PTR>
VSR_HD u32 guard_slot(const Model& M, PTR rec, int slot, int* kind) {
  Delta D;
  *kind = 0;
  if (slot < M.m0) {
    const bool en = vrst::gen<true>(M, rec, slot, D);
    *kind = D.action;
    return en ? 1u : 0u;
  }
  const int j = slot - M.m0;
  if (j >= hdr_nmsg(rec[0])) return 0;
  u32 mask = 0;
  const int base = M.m0 + j * (M.R + 1);
  if (m_dest(rec[M.fixed + j]) != ANYDEST) {
    if (vrst::gen<true>(M, rec, base, D)) { mask = 1u; *kind = D.action; }
    return mask;
  }
  for (int k = 1; k <= M.R; k++)
    if (vrst::gen<true>(M, rec, base + k, D)) { mask |= 1u << k; *kind = D.action; }
  return mask;
}

// view hashes: one salted term for the replica word, one per bag entry; no value permutation (no SYMMETRY)
template <typename PTR>
VSR_HD void hash_full(const Model& M, PTR rec, u64* H) {
  const int nmsg = hdr_nmsg(rec[0]);
  u64 sum = 0;
  for (int r = 1; r <= M.R; r++) sum += fmix64(rec[r] ^ (salt_word<0>(r) ^ M.fp_seed));
  for (int j = 0; j < nmsg; j++) sum += fmix64(rec[M.fixed + j] ^ (SALT_MSG ^ M.fp_seed));
  H[0] = sum;
}
template <typename PTR>
VSR_HD void hash_child(const Model& M, PTR rec, const Delta& D, u64* Hc) {
  u64 h = rec[M.h0];
  const u64 oldA = rec[D.r];
  if (oldA != D.rep[0]) h += fmix64(D.rep[0] ^ (salt_word<0>(D.r) ^ M.fp_seed)) - fmix64(oldA ^ (salt_word<0>(D.r) ^ M.fp_seed));
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if ((D.used >> k) & 1) {
      h += fmix64(D.pnew[k] ^ (SALT_MSG ^ M.fp_seed));
      if (D.pj(k) >= 0) h -= fmix64(rec[M.fixed + D.pj(k)] ^ (SALT_MSG ^ M.fp_seed));
    }
  Hc[0] = h;
}

// Invariants on the child (VRST.tla:806-847): mask of VIOLATED ones.  bit0 AcknowledgedWriteNotLost, bit1
// AcknowledgedWritesExistOnMajority, bit2 NoLogDivergence, bit3 CommitNumberNeverHigherThanOpNumber.  NoLogDivergence reads
// rep_log[r][op] for op <= commit: beyond the log that is a TLC evaluation error; it is reported as a violation of bit2 here
// (bit3 fails in the same state).
template <typename PTR>
VSR_HD int check_invariants_child(const Model& M, PTR rec, const Delta& D) {
  int bad = 0;
  u64 Aw[6];
#pragma unroll
  for (int r = 1; r <= 5; r++) Aw[r] = r <= M.R ? (r == D.r ? D.rep[0] : rec[r]) : 0;
  if (M.inv_mask & 3)
    for (int v = 0; v < M.n; v++) {
      if (hdr_acked(D.hdr, v) != 2) continue;
      int holders = 0;
#pragma unroll
      for (int r = 1; r <= 5; r++) {
        if (r > M.R) break;
        const u32 lg = b_log(Aw[r]);
        bool has = false;                                        // ReplicaHasOp :814-816
        for (int i = 1; i <= 3; i++) {
          const int e = blog_entry(lg, i);
          if ((e & 1) && (e >> 1) == v) has = true;
        }
        holders += has ? 1 : 0;
      }
      if ((M.inv_mask & 1) && holders == 0) bad |= 1;            // :830-835
      if ((M.inv_mask & 2) && !(holders >= M.R / 2 + 1)) bad |= 2;   // :818-824
    }
  if (M.inv_mask & 4)                                            // NoLogDivergence :806-811
    for (int opn = 1; opn <= M.n; opn++)
#pragma unroll
      for (int r1 = 1; r1 <= 5; r1++)
#pragma unroll
        for (int r2 = 1; r2 <= 5; r2++) {
          if (r1 > M.R || r2 > M.R || r2 <= r1) continue;
          if (!(opn <= a_commit(Aw[r1]) && opn <= a_commit(Aw[r2]))) continue;
          const int e1 = blog_entry(b_log(Aw[r1]), opn), e2 = blog_entry(b_log(Aw[r2]), opn);
          if (!(e1 & 1) || !(e2 & 1) || e1 != e2) bad |= 4;
        }
  if (M.inv_mask & 8)                                            // CommitNumberNeverHigherThanOpNumber :845-847
#pragma unroll
    for (int r = 1; r <= 5; r++)
      if (r <= M.R && !(a_commit(Aw[r]) <= a_op(Aw[r]))) bad |= 8;
  return bad;
}

}  // namespace vrst
}  // namespace vsr

// vsr_actions.hpp — the guarded-update action table of VSR.tla, lowered onto the packed record.
//
// Replaces tlc2.tool.impl.Tool.getNextStates for this one model.
// One call = one (action, binding) instance, identified by an ordinal `ord` relative to the parent record:
//   [0,R)            TimerSendSVC(r)                    VSR.tla:578-590
//   [R,2R)           SendDVC(r)                         VSR.tla:648-669
//   [2R,3R)          SendSV(r)                          VSR.tla:735-760
//   [3R,4R)          ExecuteOp(r)                       VSR.tla:462-476
//   [4R,m0)          ReceiveClientRequest(r,c,v)        VSR.tla:366-394
//   m0 + j*(R+1) + k the action that receives bag entry j (r = its dest, VSR.tla:272-275):
//        k = 0       by message type: ReceiveHigherSVC 602-613 / ReceiveMatchingSVC 625-634 /
//                    ReceiveHigherDVC 677-688 / ReceiveMatchingDVC 696-703 / ReceiveSV 773-793 /
//                    ReceivePrepareMsg 405-428 / ReceivePrepareOkMsg 437-447 / ReceiveGetState 526-543 /
//                    ReceiveNewState 551-567
//        k = rDest   SendGetState(r, rDest, m)          VSR.tla:496-516   (Prepare entries only)
// The four recovery actions (VSR.tla:813-894) are dead at RestartEmptyLimit = 0 and are rejected at model load.
//
// gen<true>  evaluates the guards only (used by frontier enumeration);
// gen<false> additionally produces the Delta: new header, new replica block, and the bag patches.
#pragma once
#include "vsr_model.hpp"

namespace vsr {

#define VSR_NSLOT 6        // patch slot 0: the received entry (count - 1); slot 1: a single send, or, for broadcasts,
                           // slot d = the copy addressed to replica d (1..5).  Slots are compile-time indices everywhere
                           // (fully unrolled loops), so a Delta lives in registers, never in scratch memory.

#ifndef VSR_PACK_DELTA      // 1: the small fields of a Delta share registers (bit-fields; the patched bag indices one byte each in one 64-bit word)
#define VSR_PACK_DELTA 1
#endif
struct Delta {
  u64 hdr;                 // new header (nmsg already updated)
  u64 rep[4];              // new replica block of replica r (wpr <= 4 words; rep[3] is 0 when wpr == 3)
#if VSR_PACK_DELTA
  u32 r : 3;               // the one replica an action updates
  u32 action : 5;          // A_* id (for traces)
  u32 used : 8;            // bit s: patch slot s is in use
  int err;                 // ERR_* (a word of its own: the analysis models' helpers take its address)
  u64 pjw;                 // byte s: 1 + the bag index slot s patches, 0 = appended entry
  VSR_HD int pj(int s) const { return (int)((pjw >> (8 * s)) & 0xFF) - 1; }
  VSR_HD void set_pj(int s, int j) { pjw = (pjw & ~((u64)0xFF << (8 * s))) | ((u64)(u32)(j + 1) << (8 * s)); }
  VSR_HD void clear_pj() { pjw = 0; }
#else
  int r;                   // the one replica an action updates
  int action;              // A_* id (for traces)
  int used;                // bit s: patch slot s is in use
  int err;
  int pj_[VSR_NSLOT];      // bag index patched, or -1 = appended entry
  VSR_HD int pj(int s) const { return pj_[s]; }
  VSR_HD void set_pj(int s, int j) { pj_[s] = j; }
  VSR_HD void clear_pj() {}
#endif
  u64 pnew[VSR_NSLOT];     // the new word of slot s (the previous one is the parent's bag word pj(s), read where it is needed)
};

// ---- x-slot access: replica block in memory (parent record) ... ------------------------------------------------------
template <typename PTR>
VSR_HD u32 blk_x(PTR b, int i) { return (u32)(b[1 + (i >> 1)] >> (32 * (i & 1))); }
template <typename PTR>
VSR_HD int blk_dvc_count(const Model& M, PTR b) {
  int c = 0;
  for (int s = 1; s <= M.R; s++) c += (int)(blk_x(b, s) & 1);
  return c;
}
// ---- ... and in the Delta's register copy (every index written out so that no dynamic register indexing remains) -----
VSR_HD void rep_setx(u64* b, int i, u32 x) {
  // all three words are rewritten under masks: an if-chain over b[1] / b[2] / b[3] gets merged by the optimiser into one
  // store through a selected pointer, i.e. a dynamically indexed register array, which then lives in scratch memory
  const int w = 1 + (i >> 1), sh = 32 * (i & 1);
  const u64 m = (u64)0xFFFFFFFFu << sh, v = (u64)x << sh;
  const u64 m1 = w == 1 ? m : 0, m2 = w == 2 ? m : 0, m3 = w == 3 ? m : 0;
  b[1] = (b[1] & ~m1) | (v & m1);
  b[2] = (b[2] & ~m2) | (v & m2);
  b[3] = (b[3] & ~m3) | (v & m3);
}
VSR_HD void rep_clear_dvc(u64* b) {                            // rep_dvc_recv[r] = {}  (keeps x0 = own log)
  b[1] &= 0xFFFFFFFFull;
  b[2] = 0;
  b[3] = 0;
}

// ---- bag algebra (VSR.tla:228-270) on parent bag + patch slots -----------------------------------------------------
// DiscardFunc (VSR.tla:244-245): count - 1, the key stays in the domain.  Always slot 0.
VSR_HD void bag_discard(Delta& D, int j, u64 w) {
  D.used |= 1;
  D.set_pj(0, j);
  D.pnew[0] = m_set_count(w, m_count(w) - 1);
}
// SendFunc (VSR.tla:228-231): existing key -> count + 1 (even from 0), new key -> count 1.  SLOT is a compile-time constant
// at every call site.
template <typename PTR>
VSR_HD void bag_send_at(const Model& M, PTR bag, int nmsg, Delta& D, u64 key, const int SLOT) {
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if (((D.used >> k) & 1) && (D.pnew[k] & KEYMASK) == key) {   // key touched earlier in this action
      int c = m_count(D.pnew[k]) + 1;
      if (c > 3) { D.err = ERR_REP_COUNT; return; }
      D.pnew[k] = m_set_count(D.pnew[k], c);
      return;
    }
  D.used |= 1 << SLOT;
  D.set_pj(SLOT, -1);
  D.pnew[SLOT] = m_set_count(key, 1);
  // four bag words per trip: the loads are independent, so a trip costs one LDS latency instead of four (keys are unique in a bag:
  // at most one entry matches; 0 is no key)
  for (int j0 = 0; j0 < nmsg; j0 += 4) {
    const u64 w0 = bag[j0], w1 = j0 + 1 < nmsg ? bag[j0 + 1] : 0, w2 = j0 + 2 < nmsg ? bag[j0 + 2] : 0, w3 = j0 + 3 < nmsg ? bag[j0 + 3] : 0;
    const bool h0 = (w0 & KEYMASK) == key, h1 = (w1 & KEYMASK) == key, h2 = (w2 & KEYMASK) == key, h3 = (w3 & KEYMASK) == key;
    if (h0 | h1 | h2 | h3) {
      const u64 w = h0 ? w0 : h1 ? w1 : h2 ? w2 : w3;
      int c = m_count(w) + 1;
      if (c > 3) { D.err = ERR_REP_COUNT; c = 3; }
      D.set_pj(SLOT, j0 + (h0 ? 0 : h1 ? 1 : h2 ? 2 : 3));
      D.pnew[SLOT] = m_set_count(w, c);
      return;
    }
  }
}
template <typename PTR>
VSR_HD void bag_send(const Model& M, PTR bag, int nmsg, Delta& D, u64 key) { bag_send_at(M, bag, nmsg, D, key, 1); }
// BroadcastFunc (VSR.tla:233-240): one copy per replica other than the source, dest overwritten.  One scan of the bag
// serves all copies.
template <typename PTR>
VSR_HD void bag_broadcast(const Model& M, PTR bag, int nmsg, Delta& D, u64 key, int source) {
#pragma unroll
  for (int d = 1; d <= 5; d++)
    if (d <= M.R && d != source) {
      D.used |= 1 << d;
      D.set_pj(d, -1);
      D.pnew[d] = m_set_count(m_set_dest(key, d), 1);
    }
  const u64 nodest = ~((u64)7 << 6);
  for (int j0 = 0; j0 < nmsg; j0 += 4) {                         // four independent loads per trip (see bag_send_at)
    u64 wq[4];
    wq[0] = bag[j0];
    wq[1] = j0 + 1 < nmsg ? bag[j0 + 1] : 0;
    wq[2] = j0 + 2 < nmsg ? bag[j0 + 2] : 0;
    wq[3] = j0 + 3 < nmsg ? bag[j0 + 3] : 0;
    bool any = false;
#pragma unroll
    for (int u = 0; u < 4; u++) any |= ((wq[u] ^ key) & KEYMASK & nodest) == 0;
    if (!any) continue;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const u64 w = wq[u];
      if (((w ^ key) & KEYMASK & nodest) != 0) continue;       // same record up to dest
      const int d = m_dest(w);
      int c = m_count(w) + 1;
      if (c > 3) { D.err = ERR_REP_COUNT; c = 3; }
#pragma unroll
      for (int q = 1; q <= 5; q++)
        if (q == d && ((D.used >> q) & 1)) {
          D.set_pj(q, j0 + u);
          D.pnew[q] = m_set_count(w, c);
        }
    }
  }
  // a copy whose key is the entry this action just discarded (slot 0) stacks on that patch instead
  if (D.used & 1) {
#pragma unroll
    for (int q = 1; q <= 5; q++)
      if (((D.used >> q) & 1) && ((D.pnew[q] ^ D.pnew[0]) & KEYMASK) == 0) {
        int c = m_count(D.pnew[0]) + 1;
        if (c > 3) { D.err = ERR_REP_COUNT; c = 3; }
        D.pnew[0] = m_set_count(D.pnew[0], c);
        D.used &= ~(1 << q);
      }
  }
}
template <typename PTR>
VSR_HD bool bag_has_key(PTR bag, int nmsg, u64 key) {          // key \in DOMAIN messages (any count)
  for (int j0 = 0; j0 < nmsg; j0 += 4) {
    const u64 w0 = bag[j0], w1 = j0 + 1 < nmsg ? bag[j0 + 1] : 0, w2 = j0 + 2 < nmsg ? bag[j0 + 2] : 0, w3 = j0 + 3 < nmsg ? bag[j0 + 3] : 0;
    if (((w0 & KEYMASK) == key) | ((w1 & KEYMASK) == key) | ((w2 & KEYMASK) == key) | ((w3 & KEYMASK) == key)) return true;
  }
  return false;
}

// ResetRecvMsgs + ResetSentVars (VSR.tla:299-305) on a replica block
VSR_HD void blk_reset_recv(const Model& M, u64* b) {
  (void)M;
  b[0] = a_set_svcmask(b[0], 0);
  rep_clear_dvc(b);
}
VSR_HD void blk_reset_sent(u64* b) { b[0] = a_set_sent_sv(a_set_sent_dvc(b[0], 0), 0); }

// Decode an ordinal into (group, r, c, v, j, k).  group: 0 timer, 1 sendDVC, 2 sendSV, 3 execute, 4 client, 5 message.
struct Ord { int group, r, c, v, j, k; };
VSR_HD Ord ord_decode(const Model& M, int ord) {
  Ord o;
  o.group = 5; o.r = 0; o.c = 0; o.v = 0; o.j = 0; o.k = 0;
  if (ord < 4 * M.R) {
    o.group = ord / M.R;
    o.r = ord % M.R + 1;
  } else if (ord < M.m0) {
    int idx = ord - 4 * M.R;
    o.group = 4;
    o.r = idx / (M.C * M.n) + 1;
    o.c = (idx / M.n) % M.C + 1;
    o.v = idx % M.n;
  } else {
    int q = ord - M.m0;
    o.j = q / (M.R + 1);
    o.k = q % (M.R + 1);
  }
  return o;
}

// The action table.  `rec` is the parent record (device layout).  Returns true iff the instance is enabled.
template <bool GUARD_ONLY, typename PTR>
VSR_HD bool gen(const Model& M, PTR rec, int ord, Delta& D) {
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  PTR bag = rec + M.fixed;
  Ord o = ord_decode(M, ord);
  int r = o.r;
  u64 mw = 0;
  if (o.group == 5) {
    if (o.j >= nmsg) return false;
    mw = bag[o.j];
    if (m_count(mw) == 0) return false;                        // ReceivableMsg: messages[m] > 0   VSR.tla:275
    r = m_dest(mw);                                            //                m.dest = r        VSR.tla:274
    if (o.k != 0 && (m_type(mw) != T_PREPARE || o.k == r)) return false;
  }
  PTR pb = rec + 1 + (r - 1) * M.wpr;
  const u64 A = pb[0];
  const int view = a_view(A), status = a_status(A), op = a_op(A), commit = a_commit(A);
  const bool is_primary = primary_of(M, view) == r;           // IsPrimary VSR.tla:290-291

  if (!GUARD_ONLY) {
    D.hdr = hdr;
    D.r = r;
    D.used = 0;
    D.clear_pj();
    D.err = 0;
    D.action = 0;
    D.rep[0] = pb[0];
    D.rep[1] = pb[1];
    D.rep[2] = pb[2];
    D.rep[3] = M.wpr > 3 ? pb[3] : 0;
  }
  u64* nb = D.rep;
  // the one Send / Broadcast of an action (VSR.tla:247-270) is carried out after the switch, at ONE place: the bag scan is the longest
  // loop of an action body, and lanes of different actions meet in it again instead of each running its own copy
  int send_mode = 0;                                           // 0 none, 1 Send(send_key), 2 Broadcast(send_key) to every replica but r
  u64 send_key = 0;

  switch (o.group) {
    case 0: {  // ---- TimerSendSVC (VSR.tla:578-590)
      if (!(hdr_aux_svc(hdr) < M.L)) return false;             // :579
      if (is_primary) return false;                            // :581
      if (GUARD_ONLY) return true;
      D.action = A_TimerSendSVC;
      if (view + 1 > 7) { D.err = ERR_REP_RANGE; return true; }
      nb[0] = a_set_view(nb[0], view + 1);                     // :582
      nb[0] = a_set_status(nb[0], ST_VIEWCHANGE);              // :583
      blk_reset_recv(M, nb);                                   // :584
      blk_reset_sent(nb);                                      // :585
      D.hdr = (hdr & ~((u64)7 << 8)) | ((u64)(hdr_aux_svc(hdr) + 1) << 8);   // :586
      { send_mode = 2; send_key = m_make(T_SVC, view + 1, 0, r, 0, 0, 0, 0, 0); }   // :587
      break;
    }
    case 1: {  // ---- SendDVC (VSR.tla:648-669)
      if (status != ST_VIEWCHANGE) return false;               // :650
      if (a_sent_dvc(A)) return false;                         // :651
      if (!(__builtin_popcount(a_svcmask(A)) >= M.R / 2)) return false;      // :652
      if (GUARD_ONLY) return true;
      D.action = A_SendDVC;
      nb[0] = a_set_sent_dvc(nb[0], 1);                        // :653
      u32 lg = blk_x(pb, 0);
      int prim = primary_of(M, view);
      if (prim == r) {                                         // :662-664  rep_dvc_recv[r] \union {msg}
        u32 slot = dvc_make(a_lnv(A), op, commit, lg);
        u32 cur = blk_x(pb, r);
        if ((cur & 1) && cur != slot) { D.err = ERR_REP_I2; return true; }
        rep_setx(nb, r, slot);
      } else {                                                 // :665-667
        { send_mode = 1; send_key = m_make(T_DVC, view, prim, r, op, commit, a_lnv(A), 0, lg); }
      }
      break;
    }
    case 2: {  // ---- SendSV (VSR.tla:735-760)
      if (status != ST_VIEWCHANGE) return false;               // :737
      if (a_sent_sv(A)) return false;                          // :738
      if (!(blk_dvc_count(M, pb) >= M.R / 2 + 1)) return false;   // :739
      if (GUARD_ONLY) return true;
      D.action = A_SendSV;
      // HighestLog (VSR.tla:716-722): CHOOSE among the DVCs maximal in (last_normal_vn, op_number); TLC's CHOOSE
      // takes the first such record in its value order [view, type, op, commit, dest, source, log, lnv]
      //  = smallest (commit_number, source).  HighestCommitNumber (VSR.tla:729-733) = max commit.
      int best_s = 0, best_lnv = -1, best_op = -1, best_commit = 0, max_commit = -1;
      for (int s = 1; s <= M.R; s++) {
        u32 x = blk_x(pb, s);
        if (!(x & 1)) continue;
        int l = dvc_lnv(x), o2 = dvc_op(x), c2 = dvc_commit(x);
        if (c2 > max_commit) max_commit = c2;
        bool better = (l > best_lnv) || (l == best_lnv && o2 > best_op) ||
                      (l == best_lnv && o2 == best_op && c2 < best_commit);
        if (better) { best_s = s; best_lnv = l; best_op = o2; best_commit = c2; }
      }
      if (!best_s) { D.err = ERR_EVAL_CHOOSE; return true; }
      u32 new_log = dvc_log(blk_x(pb, best_s));                // :740
      int new_on = log_len(new_log);                           // :741  Len(HighestLog(r))
      nb[0] = a_set_status(nb[0], ST_NORMAL);                  // :744
      rep_setx(nb, 0, new_log);                                // :746
      nb[0] = a_set_op(nb[0], new_on);                         // :747
      for (int p = 1; p <= M.R; p++) nb[0] = a_set_peer(nb[0], p, 0);   // :748
      nb[0] = a_set_commit(nb[0], max_commit);                 // :749
      nb[0] = a_set_sent_sv(nb[0], 1);                         // :750
      nb[0] = a_set_lnv(nb[0], view);                          // :751
      { send_mode = 2; send_key = m_make(T_SV, view, 0, r, new_on, max_commit, 0, 0, new_log); }   // :752-758
      break;
    }
    case 3: {  // ---- ExecuteOp (VSR.tla:462-476)
      if (!is_primary) return false;                           // :464
      if (status != ST_NORMAL) return false;                   // :465
      if (!(commit < op)) return false;                        // :466
      int q = 0;                                               // IsCommitted :457-460
      for (int p = 1; p <= M.R; p++) q += a_peer(A, p) >= commit + 1 ? 1 : 0;
      if (!(q >= M.R / 2)) return false;                       // :467
      if (GUARD_ONLY) return true;
      D.action = A_ExecuteOp;
      int opn = commit + 1;                                    // :468
      int e = log_byte(blk_x(pb, 0), opn);                     // :469
      if (!e) { D.err = ERR_EVAL_DOMAIN; return true; }
      nb[0] = a_set_commit(nb[0], opn);                        // :471
      int c = entry_client(e);
      if (c > M.C) { D.err = ERR_EVAL_DOMAIN; return true; }
      nb[0] = a_set_ctrow(nb[0], c, a_ctrow(A, c) | 16);       // :472  executed = TRUE
      if (hdr_acked(hdr, entry_val(e)) == 0) { D.err = ERR_EVAL_DOMAIN; return true; }
      D.hdr = hdr_set_acked(hdr, entry_val(e), 2);             // :473
      break;
    }
    case 4: {  // ---- ReceiveClientRequest (VSR.tla:366-394)
      if (!is_primary) return false;                           // :368
      if (status != ST_NORMAL) return false;                   // :369
      if (hdr_acked(hdr, o.v) != 0) return false;              // :370
      int row = a_ctrow(A, o.c);
      if (!ct_exec(row)) return false;                         // :371
      if (GUARD_ONLY) return true;
      D.action = A_ReceiveClientRequest;
      int req = ct_req(row) + 1;                               // :372
      u32 lg = blk_x(pb, 0);
      int opn = log_len(lg) + 1;                               // :373
      if (req > 3 || opn > 3) { D.err = ERR_REP_RANGE; return true; }
      int e = entry_make(view, o.v, o.c, req);                 // :374-377
      rep_setx(nb, 0, lg | ((u32)e << (8 * (opn - 1))));       // :379
      nb[0] = a_set_op(nb[0], opn);                            // :380
      nb[0] = a_set_ctrow(nb[0], o.c, ct_make(req, opn, 0));   // :381-384
      { send_mode = 2; send_key = m_make(T_PREPARE, view, 0, r, opn, commit, 0, 0, (u32)e); }   // :385-391
      D.hdr = hdr_set_acked(hdr, o.v, 1);                      // :392
      break;
    }
    default: {  // ---- message-bound actions
      const int mt = m_type(mw), mview = m_view(mw), msrc = m_source(mw), mop = m_op(mw), mcommit = m_commit(mw);
      if (o.k != 0) {  // ---- SendGetState (VSR.tla:496-516), m is a Prepare addressed to r, rDest = k
        if (is_primary) return false;                          // :498
        if (status != ST_NORMAL) return false;                 // :501
        if (!(mview > view)) return false;                     // :502
        if (!(mop > op + 1)) return false;                     // :503
        u32 lg = blk_x(pb, 0);
        int t = commit < log_len(lg) ? commit : log_len(lg);   // :504 MinVal
        u64 gs = m_make(T_GETSTATE, mview, o.k, r, t, 0, 0, 0, 0);   // :510-514
        if (bag_has_key(bag, nmsg, gs)) return false;          // SendOnce :250-251
        if (GUARD_ONLY) return true;
        D.action = A_SendGetState;
        // the mask/slot forms of rep_svc_recv / rep_dvc_recv assume records of the replica's own view
        if (a_svcmask(A) || blk_dvc_count(M, pb)) { D.err = ERR_REP_I1; return true; }
        rep_setx(nb, 0, log_prefix(lg, t));                    // :506
        nb[0] = a_set_op(nb[0], t);                            // :507
        nb[0] = a_set_view(nb[0], mview);                      // :508
        nb[0] = a_set_lnv(nb[0], mview);                       // :509
        { send_mode = 1; send_key = gs; }                         // :252
        break;
      }
      switch (mt) {
        case T_SVC: {
          if (mview > view) {  // ---- ReceiveHigherSVC (VSR.tla:602-613)
            if (GUARD_ONLY) return true;
            D.action = A_ReceiveHigherSVC;
            nb[0] = a_set_view(nb[0], mview);                  // :606
            nb[0] = a_set_status(nb[0], ST_VIEWCHANGE);        // :607
            nb[0] = a_set_svcmask(nb[0], 1 << (msrc - 1));     // :608
            rep_clear_dvc(nb);                              // :609
            blk_reset_sent(nb);                                // :610
            bag_discard(D, o.j, mw);                           // :611
            { send_mode = 2; send_key = m_make(T_SVC, mview, 0, r, 0, 0, 0, 0, 0); }
          } else if (mview == view && status == ST_VIEWCHANGE) {   // ---- ReceiveMatchingSVC (VSR.tla:625-634)
            if (GUARD_ONLY) return true;
            D.action = A_ReceiveMatchingSVC;
            nb[0] = a_set_svcmask(nb[0], a_svcmask(A) | (1 << (msrc - 1)));   // :631
            bag_discard(D, o.j, mw);                           // :632
          } else {
            return false;
          }
          break;
        }
        case T_DVC: {
          u32 slot = dvc_make(m_lnv(mw), mop, mcommit, m_lg(mw) & 0xFFFFFF);
          if (mview > view) {  // ---- ReceiveHigherDVC (VSR.tla:677-688)
            if (GUARD_ONLY) return true;
            D.action = A_ReceiveHigherDVC;
            nb[0] = a_set_view(nb[0], mview);                  // :681
            nb[0] = a_set_status(nb[0], ST_VIEWCHANGE);        // :682
            nb[0] = a_set_svcmask(nb[0], 0);                   // :683
            rep_clear_dvc(nb);                              // :684
            rep_setx(nb, msrc, slot);
            blk_reset_sent(nb);                                // :685
            bag_discard(D, o.j, mw);                           // :686
            { send_mode = 2; send_key = m_make(T_SVC, mview, 0, r, 0, 0, 0, 0, 0); }
          } else if (mview == view) {  // ---- ReceiveMatchingDVC (VSR.tla:696-703)
            if (GUARD_ONLY) return true;
            D.action = A_ReceiveMatchingDVC;
            u32 cur = blk_x(pb, msrc);
            if ((cur & 1) && cur != slot) { D.err = ERR_REP_I2; return true; }
            rep_setx(nb, msrc, slot);                          // :700
            bag_discard(D, o.j, mw);                           // :701
          } else {
            return false;
          }
          break;
        }
        case T_SV: {  // ---- ReceiveSV (VSR.tla:773-793)
          if (!(mview >= view)) return false;                  // :776
          if (GUARD_ONLY) return true;
          D.action = A_ReceiveSV;
          nb[0] = a_set_status(nb[0], ST_NORMAL);              // :777
          nb[0] = a_set_view(nb[0], mview);                    // :778
          rep_setx(nb, 0, m_lg(mw) & 0xFFFFFF);                // :779
          nb[0] = a_set_op(nb[0], mop);                        // :780
          nb[0] = a_set_commit(nb[0], mcommit);                // :781
          nb[0] = a_set_lnv(nb[0], mview);                     // :782
          blk_reset_recv(M, nb);                               // :783
          blk_reset_sent(nb);                                  // :784
          bag_discard(D, o.j, mw);
          if (commit < mop)                                    // :785 (old commit number)
            { send_mode = 1; send_key = m_make(T_PREPAREOK, mview, primary_of(M, mview), r, mop, 0, 0, 0, 0); }   // :786-790
          break;
        }
        case T_PREPARE: {  // ---- ReceivePrepareMsg (VSR.tla:405-428)
          if (status != ST_NORMAL) return false;               // :408
          if (mview != view) return false;                     // :409
          if (mop != op + 1) return false;                     // :410
          if (GUARD_ONLY) return true;
          D.action = A_ReceivePrepareMsg;
          int e = (int)(m_lg(mw) & 0xFF);
          u32 lg = blk_x(pb, 0);
          int pos = log_len(lg) + 1;                           // Append :411
          if (pos > 3) { D.err = ERR_REP_RANGE; return true; }
          rep_setx(nb, 0, lg | ((u32)e << (8 * (pos - 1))));
          nb[0] = a_set_op(nb[0], mop);                        // :412
          nb[0] = a_set_commit(nb[0], mcommit);                // :413
          for (int c = 1; c <= M.C; c++) {                     // :414-421
            if (c == entry_client(e)) {
              nb[0] = a_set_ctrow(nb[0], c, ct_make(entry_req(e), mop, mop <= mcommit ? 1 : 0));
            } else {
              // VSR.tla:421 reads `m.commit`, a field PrepareMsg does not have: TLC evaluation error
              if (!M.assume_commit) { D.err = ERR_EVAL_421; return true; }
              int row = a_ctrow(A, c);
              nb[0] = a_set_ctrow(nb[0], c, ct_make(ct_req(row), ct_op(row), ct_op(row) <= mcommit ? 1 : 0));
            }
          }
          bag_discard(D, o.j, mw);
          { send_mode = 1; send_key = m_make(T_PREPAREOK, view, msrc, r, mop, 0, 0, 0, 0); }   // :422-426
          break;
        }
        case T_PREPAREOK: {  // ---- ReceivePrepareOkMsg (VSR.tla:437-447)
          if (!is_primary) return false;                       // :440
          if (status != ST_NORMAL) return false;               // :441
          if (mview != view) return false;                     // :442
          if (!(mop > a_peer(A, msrc))) return false;          // :443
          if (GUARD_ONLY) return true;
          D.action = A_ReceivePrepareOkMsg;
          nb[0] = a_set_peer(nb[0], msrc, mop);                // :444
          bag_discard(D, o.j, mw);                             // :445
          break;
        }
        case T_GETSTATE: {  // ---- ReceiveGetState (VSR.tla:526-543)
          if (view != mview) return false;                     // :529
          if (status != ST_NORMAL) return false;               // :530
          if (!(op > mop)) return false;                       // :531
          if (GUARD_ONLY) return true;
          D.action = A_ReceiveGetState;
          u32 lg = blk_x(pb, 0);
          u32 part = log_prefix(lg, op) & ~log_prefix(0xFFFFFF, mop);   // entries mop+1 .. op   :535-536
          for (int on = mop + 1; on <= op; on++)
            if (!log_byte(lg, on)) { D.err = ERR_EVAL_DOMAIN; return true; }
          bag_discard(D, o.j, mw);
          { send_mode = 1; send_key = m_make(T_NEWSTATE, view, msrc, r, op, commit, 0, mop + 1, part); }   // :533-541
          break;
        }
        case T_NEWSTATE: {  // ---- ReceiveNewState (VSR.tla:551-567)
          if (view != mview) return false;                     // :554
          if (status != ST_NORMAL) return false;               // :555
          if (!(op == m_first_op(mw) - 1)) return false;       // :556
          if (GUARD_ONLY) return true;
          D.action = A_ReceiveNewState;
          u32 lg = blk_x(pb, 0), ml = m_lg(mw) & 0xFFFFFF;
          u32 nl = log_prefix(lg, op) | (log_prefix(ml, mop) & ~log_prefix(0xFFFFFF, op));   // :557-561
          for (int on = 1; on <= mop; on++)
            if (!log_byte(nl, on)) { D.err = ERR_EVAL_DOMAIN; return true; }
          rep_setx(nb, 0, nl);
          nb[0] = a_set_op(nb[0], mop);                        // :562
          bag_discard(D, o.j, mw);                             // :564 (client table untouched, :563)
          break;
        }
        default:
          return false;
      }
      break;
    }
  }
  if (!GUARD_ONLY) {
    if (send_mode == 1) bag_send(M, bag, nmsg, D, send_key);
    else if (send_mode == 2) bag_broadcast(M, bag, nmsg, D, send_key, r);
    int na = 0;
#pragma unroll
    for (int k = 0; k < VSR_NSLOT; k++) na += (((D.used >> k) & 1) && D.pj(k) < 0) ? 1 : 0;
    if (nmsg + na > M.max_bag) D.err = D.err ? D.err : ERR_REP_BAG;
    D.hdr = hdr_set_nmsg(D.hdr, nmsg + na);
  }
  return true;
}

// ---- guards only, for frontier enumeration ------------------------------------------------------------------------
// The same guards as gen<>, restated per *slot* so that a wave can evaluate one slot kind for 64 different records in
// lock-step: slot < m0 = the replica-bound ordinal itself; slot = m0 + j = "whatever receives bag entry j".
// Returns a bit mask: bit k set <=> ordinal (slot < m0 ? slot : m0 + j*(R+1) + k) is enabled; *kind0 = action id of bit 0.
// k_expand re-checks every emitted instance with gen<false> (a disagreement raises ERR_INTERNAL), and the parity tests
// compare the generated counts with the oracle, so the two statements of the guards cannot drift apart silently.
// `Areg[r]` = A word of replica r (1..5), already in registers: the frontier enumeration evaluates ~50 slots per record and
// must not re-read them from LDS every time.  Selected by value (never through a pointer into the array).
VSR_HD u64 areg_of(const u64* Areg, int r) {
  return r == 1 ? Areg[1] : r == 2 ? Areg[2] : r == 3 ? Areg[3] : r == 4 ? Areg[4] : Areg[5];
}
// decode of a replica-bound slot: group | r << 3 | c << 6 | v << 8 (integer divisions by run-time constants: k_expand
// tabulates this once per block and passes the entry as `info`; info < 0 = decode here)
VSR_HD int slot_info(const Model& M, int slot) {
  if (slot < 4 * M.R) return (slot / M.R) | ((slot % M.R + 1) << 3);
  const int idx = slot - 4 * M.R;
  return 4 | ((idx / (M.C * M.n) + 1) << 3) | (((idx / M.n) % M.C + 1) << 6) | ((idx % M.n) << 8);
}
template <typename PTR>
VSR_HD u32 guard_slot_pre(const Model& M, PTR rec, u64 hdr, const u64* Areg, int slot, int* kind0, int info = -1) {
  if (slot < M.m0) {
    if (info < 0) info = slot_info(M, slot);
    const int group = info & 7, r = (info >> 3) & 7, c = (info >> 6) & 3, v = (info >> 8) & 3;
    PTR pb = rec + 1 + (r - 1) * M.wpr;
    const u64 A = areg_of(Areg, r);
    const bool prim = primary_of(M, a_view(A)) == r;
    const int st = a_status(A);
    switch (group) {
      case 0: *kind0 = A_TimerSendSVC; return (hdr_aux_svc(hdr) < M.L && !prim) ? 1u : 0u;                  // VSR.tla:579-581
      case 1: *kind0 = A_SendDVC;                                                                             // :650-652
        return (st == ST_VIEWCHANGE && !a_sent_dvc(A) && __builtin_popcount(a_svcmask(A)) >= M.R / 2) ? 1u : 0u;
      case 2: *kind0 = A_SendSV;                                                                              // :737-739
        return (st == ST_VIEWCHANGE && !a_sent_sv(A) && blk_dvc_count(M, pb) >= M.R / 2 + 1) ? 1u : 0u;
      case 3: {                                                                                               // :464-467
        *kind0 = A_ExecuteOp;
        if (!(prim && st == ST_NORMAL && a_commit(A) < a_op(A))) return 0;
        int q = 0;
        for (int p = 1; p <= M.R; p++) q += a_peer(A, p) >= a_commit(A) + 1 ? 1 : 0;
        return q >= M.R / 2 ? 1u : 0u;
      }
      default: *kind0 = A_ReceiveClientRequest;                                                               // :368-371
        return (prim && st == ST_NORMAL && hdr_acked(hdr, v) == 0 && ct_exec(a_ctrow(A, c))) ? 1u : 0u;
    }
  }
  const int j = slot - M.m0;
  if (j >= hdr_nmsg(hdr)) return 0;
  const u64 mw = rec[M.fixed + j];
  if (m_count(mw) == 0) return 0;                               // ReceivableMsg VSR.tla:272-275
  const int r = m_dest(mw);
  PTR pb = rec + 1 + (r - 1) * M.wpr;
  const u64 A = areg_of(Areg, r);
  // Branch-free: the lanes of a wave look at 64 different messages of 64 different types; every type's guard is evaluated
  // with predicated arithmetic and selected by type, instead of seven divergent branches.
  const int t = m_type(mw);
  const int view = a_view(A), op = a_op(A);
  const int mview = m_view(mw), mop = m_op(mw);
  const bool normal = a_status(A) == ST_NORMAL, vc = a_status(A) == ST_VIEWCHANGE;
  const bool prim = primary_of(M, view) == r;
  const bool higher = mview > view, same = mview == view;
  const bool en_svc = higher | (same & vc);                                                                   // :605 / :628-630
  const bool en_dvc = higher | same;                                                                          // :680 / :699 ; SV :776
  const bool en_prep = normal & same & (mop == op + 1);                                                       // :408-410
  const bool en_pok = prim & normal & same & (mop > a_peer(A, m_source(mw)));                                 // :440-443
  const bool en_gs = same & normal & (op > mop);                                                              // :529-531
  const bool en_ns = same & normal & (op == m_first_op(mw) - 1);                                              // :554-556
  const bool en = t == T_SVC ? en_svc : (t == T_DVC || t == T_SV) ? en_dvc : t == T_PREPARE ? en_prep : t == T_PREPAREOK ? en_pok
                  : t == T_GETSTATE ? en_gs : t == T_NEWSTATE ? en_ns : false;
  // action id by (type, higher): one 64-bit table lookup, 4 bits per entry, index = type * 2 + higher
  const u64 KIND = ((u64)A_ReceiveMatchingSVC << (4 * (2 * T_SVC))) | ((u64)A_ReceiveHigherSVC << (4 * (2 * T_SVC + 1))) |
                   ((u64)A_ReceivePrepareMsg << (4 * (2 * T_PREPARE))) | ((u64)A_ReceivePrepareMsg << (4 * (2 * T_PREPARE + 1))) |
                   ((u64)A_ReceivePrepareOkMsg << (4 * (2 * T_PREPAREOK))) | ((u64)A_ReceivePrepareOkMsg << (4 * (2 * T_PREPAREOK + 1))) |
                   ((u64)A_ReceiveMatchingDVC << (4 * (2 * T_DVC))) | ((u64)A_ReceiveHigherDVC << (4 * (2 * T_DVC + 1))) |
                   ((u64)A_ReceiveSV << (4 * (2 * T_SV))) | ((u64)A_ReceiveSV << (4 * (2 * T_SV + 1))) |
                   ((u64)A_ReceiveGetState << (4 * (2 * T_GETSTATE))) | ((u64)A_ReceiveGetState << (4 * (2 * T_GETSTATE + 1))) |
                   ((u64)A_ReceiveNewState << (4 * (2 * T_NEWSTATE))) | ((u64)A_ReceiveNewState << (4 * (2 * T_NEWSTATE + 1)));
  *kind0 = (int)((KIND >> (4 * (2 * t + (higher ? 1 : 0)))) & 15);
  u32 mask = en ? 1u : 0u;
  if (t == T_PREPARE && !prim && normal && higher && mop > op + 1) {                                          // SendGetState :498-503 (rare)
    const u32 lg = blk_x(pb, 0);
    const int tr = a_commit(A) < log_len(lg) ? a_commit(A) : log_len(lg);
    for (int d = 1; d <= M.R; d++)
      if (d != r && !bag_has_key(rec + M.fixed, hdr_nmsg(hdr), m_make(T_GETSTATE, mview, d, r, tr, 0, 0, 0, 0))) mask |= 1u << d;
  }
  return mask;
}

// ---- two-stage enumeration (k_expand, VSR.tla model): a cheap superset test per bag entry, the exact guard only on the survivors --
// prefilter_lut(A, r): bit (type + 8 * view_number) is set iff a message of that type and view addressed to replica r COULD be
// received (or, a Prepare, could start a SendGetState) in the state A of r — every conjunct of guard_slot_pre that only looks at
// (type, view_number) and r's status / view / primary-ness, none of those that compare op numbers.  The low six bits of a bag
// word ARE type | view << 3, so the test is one shift.  A superset of "guard_slot_pre(...) != 0" by construction: each row
// below is the guard's own condition with the op-number conjunct dropped.
VSR_HD u64 prefilter_lut(const Model& M, u64 A, int r) {
  const int view = a_view(A), st = a_status(A);
  const bool prim = primary_of(M, view) == r;
  const u64 COL = 0x0101010101010101ULL;
  const u64 ge = ~(u64)0 << (8 * view), eq = (u64)0xFF << (8 * view), gt = ge & ~eq;
  u64 lut = ((COL << T_SVC) & (gt | (st == ST_VIEWCHANGE ? eq : (u64)0)))                     // VSR.tla:605 / :628-630
            | (((COL << T_DVC) | (COL << T_SV)) & ge);                                          // :680 / :699 / :776
  if (st == ST_NORMAL) {
    lut |= eq & ((COL << T_PREPARE) | (COL << T_GETSTATE) | (COL << T_NEWSTATE)                // :408-409, :529-530, :554-555
                 | (prim ? (COL << T_PREPAREOK) : (u64)0));                                     // :440-442
    if (!prim) lut |= gt & (COL << T_PREPARE);                                                  // SendGetState :498-502
  }
  return lut;
}
// the replica-bound instances of replica r as a bit mask: bit 0 TimerSendSVC(r), 1 SendDVC(r), 2 SendSV(r), 3 ExecuteOp(r),
// 4 + (c-1)*n + v ReceiveClientRequest(r, c, v) — the same guards as the slot < m0 half of guard_slot_pre
template <typename PTR>
VSR_HD u32 rep_slots_mask(const Model& M, PTR rec, u64 hdr, u64 A, int r) {
  PTR pb = rec + 1 + (r - 1) * M.wpr;
  const bool prim = primary_of(M, a_view(A)) == r;
  const int st = a_status(A);
  u32 m = (hdr_aux_svc(hdr) < M.L && !prim) ? 1u : 0u;                                          // VSR.tla:579-581
  if (st == ST_VIEWCHANGE) {
    if (!a_sent_dvc(A) && __builtin_popcount(a_svcmask(A)) >= M.R / 2) m |= 2u;                 // :650-652
    if (!a_sent_sv(A) && blk_dvc_count(M, pb) >= M.R / 2 + 1) m |= 4u;                          // :737-739
  } else if (st == ST_NORMAL && prim) {
    if (a_commit(A) < a_op(A)) {                                                                // :464-467
      int q = 0;
      for (int p = 1; p <= M.R; p++) q += a_peer(A, p) >= a_commit(A) + 1 ? 1 : 0;
      if (q >= M.R / 2) m |= 8u;
    }
    for (int c = 1; c <= M.C; c++)                                                              // :368-371
      if (ct_exec(a_ctrow(A, c)))
        for (int v = 0; v < M.n; v++)
          if (hdr_acked(hdr, v) == 0) m |= 16u << ((c - 1) * M.n + v);
  }
  return m;
}

template <typename PTR>
VSR_HD u32 guard_slot(const Model& M, PTR rec, int slot, int* kind0) {
  u64 Areg[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int r = 1; r <= 5; r++)
    if (r <= M.R) Areg[r] = rec[1 + (r - 1) * M.wpr];
  return guard_slot_pre(M, rec, rec[0], Areg, slot, kind0);
}

// Incremental view hashes of the child: Hc[i] = Hp[i] + sum over the changed words of (new term - old term).  A word without
// values (no in-use log entry byte: every A word, SVC / PrepareOk / GetState messages, empty DVC slots) has the same term under
// every permutation: it is hashed once (`dinv`), not once per permutation.
template <int K>
VSR_HD void hash_word_delta(const Model& M, u64 w_old, u64 w_new, int r, u64& dinv, u64* d) {
  if (w_old == w_new) return;
  constexpr u64 m01 = K == 0 ? (u64)0 : K == 1 ? LOGB_REP1 : LOGB_REPK;
  const u64 s = salt_seeded<K>(M.fp_seed, r);
  if (K == 0 || !(word_has_values(w_old, m01) | word_has_values(w_new, m01))) {
    dinv += fmix64(w_new ^ s) - fmix64(w_old ^ s);
    return;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (i >= M.np) break;
    const u32 pt = M.pitab[i];
    d[i] += fmix64(permute_word(w_new, m01, pt) ^ s) - fmix64(permute_word(w_old, m01, pt) ^ s);
  }
}
VSR_HD void hash_msg_term(const Model& M, u64 w, bool add, u64& dinv, u64* d) {
  if (!word_has_values(w, LOGB_MSG)) {
    const u64 h = fmix64(w ^ (SALT_MSG ^ M.fp_seed));
    dinv += add ? h : (u64)0 - h;
    return;
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (i >= M.np) break;
    const u64 h = hash_msg(M, w, M.pitab[i]);
    d[i] += add ? h : (u64)0 - h;
  }
}
template <typename PTR>
VSR_HD void hash_child(const Model& M, PTR rec, const Delta& D, u64* Hc) {
  PTR pb = rec + 1 + (D.r - 1) * M.wpr;
  u64 dinv = 0;
  u64 d[6] = {0, 0, 0, 0, 0, 0};
  hash_word_delta<0>(M, pb[0], D.rep[0], D.r, dinv, d);
  hash_word_delta<1>(M, pb[1], D.rep[1], D.r, dinv, d);
  if (M.wpr > 2) hash_word_delta<2>(M, pb[2], D.rep[2], D.r, dinv, d);
  if (M.wpr > 3) hash_word_delta<3>(M, pb[3], D.rep[3], D.r, dinv, d);
#pragma unroll
  for (int k = 0; k < VSR_NSLOT; k++)
    if ((D.used >> k) & 1) {
      hash_msg_term(M, D.pnew[k], true, dinv, d);
      if (D.pj(k) >= 0) hash_msg_term(M, rec[M.fixed + D.pj(k)], false, dinv, d);
    }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    if (i >= M.np) break;
    Hc[i] = rec[M.h0 + i] + dinv + d[i];
  }
}

// Invariants on the child (VSR.tla:933-950).  Returns the mask of VIOLATED invariants.
template <typename PTR>
VSR_HD int check_invariants_child(const Model& M, PTR rec, const Delta& D) {
  if (!(M.inv_mask & 3)) return 0;
  int bad = 0;
  const u32 own_log = (u32)D.rep[1];   // read once: a select between &D.rep[1] and &rec[..] would make the Delta addressable
  for (int v = 0; v < M.n; v++) {
    if (hdr_acked(D.hdr, v) != 2) continue;                    // aux_client_acked[v] = TRUE   :940, :948
    int holders = 0;
    for (int r = 1; r <= M.R; r++) {
      u32 lg = (r == D.r) ? own_log : (u32)rec[1 + (r - 1) * M.wpr + 1];
      bool has = false;                                        // ReplicaHasOp :933-935
      for (int i = 1; i <= 3; i++) {
        int e = log_byte(lg, i);
        if (e && entry_val(e) == v) has = true;
      }
      holders += has ? 1 : 0;
    }
    if ((M.inv_mask & 1) && holders == 0) bad |= 1;            // AcknowledgedWriteNotLost :945-950
    if ((M.inv_mask & 2) && !(holders >= M.R / 2 + 1)) bad |= 2;   // AcknowledgedWritesExistOnMajority :937-943
  }
  return bad;
}

// Number of ordinals to scan for a parent record.
VSR_HD int ord_count(const Model& M, int nmsg) { return M.m0 + nmsg * (M.R + 1); }

}  // namespace vsr

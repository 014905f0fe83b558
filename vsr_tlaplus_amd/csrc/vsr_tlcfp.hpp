// vsr_tlcfp.hpp — TLC's own fingerprint of a VSR.tla state, computed from the packed record.
//
// EVERYTHING here that concerns TLC is [TLC-RECALLED]: TLC (tla2tools.jar) is not in /root/reference, no JVM exists in the build image, the
// reference pins no TLC version and no fingerprint; nothing below could be checked against a single bit TLC produces.  What IS pinned:
// the Rabin arithmetic against an independent bit-by-bit polynomial division (tests/test_tlc_fp64.py: known answers, GF(2) linearity,
// concatenation), and this serialiser — which walks the PACKED record — against the one of the CPU oracle, which builds a generic value tree from
// the UNPACKED state (tlc_fp64.cpp of the test infrastructure): the two share no code and must produce the same byte stream for every state of configs 1-2.
//
// (1) tlc2.util.FP64: a 64-bit Rabin fingerprint in reflected bit order (bit 63 = x^0).  IrredPoly = Polys[0] = 0x911498AE0E66BAD6 (`-fp 0`);
//     New() = IrredPoly; Extend(fp, byte b) = (fp >>> 8) ^ ByteModTable_7[(b ^ fp) & 0xFF], ByteModTable_7[i] = XOR over the set bits k of i of
//     x^(71-k) mod P; ints extend four bytes, least significant first; strings one byte per char.
// (2) Value.fingerPrint (tlc2.value.impl.*): a type tag byte, then the content —
//       IntValue      INTVALUE(1) int                       BoolValue   BOOLVALUE(0) 't' / 'f'
//       ModelValue    MODELVALUE(21) int index              (index = creation order = order of first appearance in the cfg: the Values, then
//                                                            Normal, ViewChange, Recovering, RequestMsg, ..., RecoveryResponseMsg, Nil — VSR.cfg:6-24)
//                     DOUBTFUL: ModelValue.fingerPrint may extend with the value's UniqueString TOKEN (val.fingerPrint -> FP64.Extend(fp, tok)),
//                     assigned in interning order over all strings of the parsed spec, not over the model values alone; then these bytes — and
//                     the min-permutation choice, which compares model values by this index — differ from TLC's (round-3 advice; unpinnable here)
//       StringValue   STRINGVALUE(3) int length, chars
//       RecordValue   FCNRCDVALUE(9) int #fields, then per field in normal order: the NAME as a string value, the value
//       TupleValue    FCNRCDVALUE(9) int length, then per element: INTVALUE(1) int index (from 1), the element
//       FcnRcdValue   FCNRCDVALUE(9) int size, then per pair in normal domain order: the domain value (interval domain lo..hi: INTVALUE int), the range value
//       SetEnumValue / IntervalValue   SETENUMVALUE(5) int size, the elements in normal order (interval: INTVALUE int each)
//     Normal order (partly evidenced by the reference's printed trace): record fields by the interning order of their
//     names = first occurrence in VSR.tla; the elements of sets and the domains of functions by Value.compareTo (see (4)).
// (3) TLCState.fingerPrint with `VIEW view` (VSR.tla:149-150): the fingerprint of the VALUE of `view`, a tuple of tuples of the state
//     variables, started from FP64.New().
// (4) SYMMETRY (TLCStateMut.fingerPrint): the values of ALL state variables are permuted, the permuted states are compared variable by variable in
//     declaration order with Value.compareTo, and the view of the SMALLEST permuted state is fingerprinted (cmp_state / min_permutation below).
//     Value.compareTo as used here: ints numerically; booleans FALSE < TRUE; model values by the interning order of their names (= cfg order);
//     RecordValue: arity, then field by field name (interning order) and value, interleaved; TupleValue: length, then elementwise; SetEnumValue:
//     size, then elementwise in normal order; FcnRcdValue: size, then (interval domains) the lower bound and the values, (explicit domains) pair by
//     pair domain value and range value.
#pragma once
#include "vsr_model.hpp"
#include "vsr_actions.hpp"   // blk_x

namespace vsr {
namespace tlcfp {

constexpr u64 IRRED_POLY = 0x911498AE0E66BAD6ULL;   // tlc2.util.FP64.Polys[0]  [TLC-RECALLED]
enum { TAG_BOOL = 0, TAG_INT = 1, TAG_STRING = 3, TAG_SETENUM = 5, TAG_FCNRCD = 9, TAG_MODEL = 21 };   // tlc2.value.ValueConstants

// x^i mod P for i = 64 .. 71 in reflected representation, folded at compile time: T[k] = x^(71-k)
struct ByteTable { u64 t[256]; };
constexpr ByteTable make_table() {
  ByteTable T{};
  u64 pw[72] = {};
  u64 t = 0x8000000000000000ULL;                      // x^0
  for (int i = 0; i < 72; i++) {
    pw[i] = t;
    t = (t >> 1) ^ ((t & 1) ? IRRED_POLY : 0);        // times x, reduced
  }
  for (int i = 0; i < 256; i++) {
    u64 v = 0;
    for (int k = 0; k < 8; k++)
      if (i & (1 << k)) v ^= pw[71 - k];
    T.t[i] = v;
  }
  return T;
}
#if defined(__HIPCC__)
__device__ const ByteTable D_TABLE = make_table();    // the kernels copy it to LDS: every lane indexes it with its own byte
#endif
static const ByteTable H_TABLE = make_table();

// ---- sinks: the serialiser feeds bytes to one of these -------------------------------------------------------------------------
struct FpSink {                                       // the running FP64; tab = ByteModTable_7 (host: H_TABLE.t, device: its copy in LDS)
  u64 fp;
  const u64* tab;
  VSR_HD void byte(u32 b) { fp = (fp >> 8) ^ tab[(b ^ (u32)fp) & 0xFF]; }
};
struct ByteSink {                                     // the byte stream itself (host: diagnostics, the tests' stream comparison)
  unsigned char* out;
  u64 cap, n;
  VSR_HD void byte(u32 b) {
    if (n < cap) out[n] = (unsigned char)b;
    n++;
  }
};

template <typename S> VSR_HD void put_int(S& s, int x) { for (int i = 0; i < 4; i++) s.byte(((u32)x >> (8 * i)) & 0xFF); }
template <typename S> VSR_HD void put_Int(S& s, int x) { s.byte(TAG_INT); put_int(s, x); }
template <typename S> VSR_HD void put_Bool(S& s, bool b) { s.byte(TAG_BOOL); s.byte(b ? 't' : 'f'); }
template <typename S> VSR_HD void put_Model(S& s, int index) { s.byte(TAG_MODEL); put_int(s, index); }
template <typename S> VSR_HD void put_name(S& s, const char* name) {                     // a field name as a string value
  int len = 0;
  while (name[len]) len++;
  s.byte(TAG_STRING);
  put_int(s, len);
  for (int i = 0; i < len; i++) s.byte((unsigned char)name[i]);
}
template <typename S> VSR_HD void put_fcn(S& s, int n) { s.byte(TAG_FCNRCD); put_int(s, n); }   // header of a record / tuple / function
template <typename S> VSR_HD void put_set(S& s, int n) { s.byte(TAG_SETENUM); put_int(s, n); }
template <typename S> VSR_HD void put_idx(S& s, int i) { s.byte(TAG_INT); put_int(s, i); }      // tuple index / interval domain element

// model value indices: creation order in the cfg (VSR.cfg:6-24): the n Values first
VSR_HD int mv_value(const Model& M, int v) { (void)M; return v; }
VSR_HD int mv_status(const Model& M, int st) { return M.n + st; }                        // Normal, ViewChange, Recovering
VSR_HD int mv_msgtype(const Model& M, int t) {                                           // RequestMsg = n + 3 ... (packed T_* ids -> cfg order)
  // cfg order: RequestMsg, ReplyMsg, PrepareMsg, PrepareOkMsg, CommitMsg, StartViewChangeMsg, DoViewChangeMsg, StartViewMsg, GetStateMsg, NewStateMsg, ...
  const int pos = t == T_PREPARE ? 2 : t == T_PREPAREOK ? 3 : t == T_SVC ? 5 : t == T_DVC ? 6 : t == T_SV ? 7 : t == T_GETSTATE ? 8 : 9;
  return M.n + 3 + pos;
}

// a log entry byte (view(3) | value(2) << 3 | (client-1) << 5 | request(2) << 6) as the record [view_number, operation, client_id, request_number]
template <typename S> VSR_HD void put_entry(const Model& M, S& s, int e, u32 pt) {
  put_fcn(s, 4);
  put_name(s, "view_number"); put_Int(s, e & 7);
  put_name(s, "operation"); put_Model(s, mv_value(M, (int)((pt >> (2 * entry_val(e))) & 3)));
  put_name(s, "client_id"); put_Int(s, entry_client(e));
  put_name(s, "request_number"); put_Int(s, entry_req(e));
}
// the entries first .. last of a packed log as a tuple (first == 1) or a function over first..last (NewState.log, VSR.tla:535-536)
template <typename S> VSR_HD void put_log(const Model& M, S& s, u32 lg, int first, int last, u32 pt) {
  const int n = last >= first ? last - first + 1 : 0;
  put_fcn(s, n);
  for (int i = first; i <= last; i++) {
    put_idx(s, i);
    put_entry(M, s, log_byte(lg, i), pt);
  }
}

enum { MAX_MSGS = 128 };   // BFS records hold < 95 bag words (Model::max_bag); the batch entry point refuses longer ones

// ---- order of message records.  RecordValue.compareTo: arity, then field by field the NAME and then the VALUE (interleaved) -----------------------
// Every message type starts [view_number, type, ...]: records of one arity order by view_number, then by type (model values: by index), and only
// records of one type ever reach their third field.  All comparisons take one permutation per side: sorting under a permutation uses the same one
// twice, choosing TLC's representative (min_permutation below) compares the same state under two.
VSR_HD int msg_arity(int t) { return t == T_SVC ? 4 : (t == T_PREPAREOK || t == T_GETSTATE) ? 5 : (t == T_PREPARE || t == T_SV) ? 7 : 8; }
// entry bytes compare as records [view_number, operation, client_id, request_number]; operation = a model value: by index under the side's permutation
VSR_HD int cmp_entry(int a, u32 pa, int b, u32 pb) {
  if ((a & 7) != (b & 7)) return (a & 7) < (b & 7) ? -1 : 1;
  const int va = (int)((pa >> (2 * entry_val(a))) & 3), vb = (int)((pb >> (2 * entry_val(b))) & 3);
  if (va != vb) return va < vb ? -1 : 1;
  if (entry_client(a) != entry_client(b)) return entry_client(a) < entry_client(b) ? -1 : 1;
  if (entry_req(a) != entry_req(b)) return entry_req(a) < entry_req(b) ? -1 : 1;
  return 0;
}
VSR_HD int cmp_log(u32 la, int fa, int na, u32 pa, u32 lb, int fb, int nb, u32 pb) {    // length, then (interval functions) the domain's start, then elementwise
  if (na != nb) return na < nb ? -1 : 1;
  if (na == 0) return 0;
  if (fa != fb) return fa < fb ? -1 : 1;
  for (int i = 0; i < na; i++) {
    const int c = cmp_entry(log_byte(la, fa + i), pa, log_byte(lb, fb + i), pb);
    if (c) return c;
  }
  return 0;
}
// bag word a under permutation pa against bag word b under pb, as message records (delivery counts are not part of the record)
VSR_HD int cmp_msg(const Model& M, u64 a, u32 pa, u64 b, u32 pb) {
  const int ta = m_type(a), tb = m_type(b);
  if (msg_arity(ta) != msg_arity(tb)) return msg_arity(ta) < msg_arity(tb) ? -1 : 1;
  if (m_view(a) != m_view(b)) return m_view(a) < m_view(b) ? -1 : 1;
  if (ta != tb) return mv_msgtype(M, ta) < mv_msgtype(M, tb) ? -1 : 1;
  const u32 la = m_lg(a) & 0xFFFFFF, lb = m_lg(b) & 0xFFFFFF;
#define VSR_CMPF(fa, fb) if ((fa) != (fb)) return (fa) < (fb) ? -1 : 1;
  if (ta == T_PREPARE) {
    const int c = cmp_entry((int)(la & 0xFF), pa, (int)(lb & 0xFF), pb);
    if (c) return c;
  }
  if (ta != T_SVC) { VSR_CMPF(m_op(a), m_op(b)) }
  if (ta == T_PREPARE || ta == T_SV || ta == T_DVC || ta == T_NEWSTATE) { VSR_CMPF(m_commit(a), m_commit(b)) }
  VSR_CMPF(m_dest(a), m_dest(b))
  VSR_CMPF(m_source(a), m_source(b))
  if (ta == T_SV || ta == T_DVC) {
    const int c = cmp_log(la, 1, log_len(la), pa, lb, 1, log_len(lb), pb);
    if (c) return c;
    if (ta == T_DVC) { VSR_CMPF(m_lnv(a), m_lnv(b)) }
  } else if (ta == T_NEWSTATE) {
    const int fa = m_first_op(a), fb = m_first_op(b);
    const int na = m_op(a) - fa + 1 > 0 ? m_op(a) - fa + 1 : 0, nb = m_op(b) - fb + 1 > 0 ? m_op(b) - fb + 1 : 0;
    const int c = cmp_log(la, fa, na, pa, lb, fb, nb, pb);
    if (c) return c;
    VSR_CMPF(fa, fb)
  }
#undef VSR_CMPF
  return 0;
}

// ord[k] = index of the k-th smallest of n message records under permutation pt (keys distinct: rank = number of smaller ones)
VSR_HD void msg_order(const Model& M, const u64* w, int n, u32 pt, unsigned char* ord) {
  for (int j = 0; j < n && j < MAX_MSGS; j++) ord[j] = (unsigned char)j;   // a record with two equal keys (never a reachable state) leaves a rank unused: still an index inside the bag
  for (int j = 0; j < n && j < MAX_MSGS; j++) {
    int smaller = 0;
    for (int i = 0; i < n; i++) smaller += (i != j && cmp_msg(M, w[i], pt, w[j], pt) < 0) ? 1 : 0;
    ord[smaller & (MAX_MSGS - 1)] = (unsigned char)j;
  }
}
// the DoViewChange records replica r holds (its view, addressed to it, one per source at most) as bag-style words, sorted under pt
VSR_HD int dvc_set(const Model& M, const u64* blk, int r, u32 pt, u64* dv) {
  const u64 A = blk[0];
  int nd = 0;
  for (int src = 1; src <= M.R; src++) {
    const u32 x = blk_x(blk, src);
    if (!(x & 1)) continue;
    const u64 w = m_make(T_DVC, a_view(A), r, src, dvc_op(x), dvc_commit(x), dvc_lnv(x), 0, dvc_log(x));
    int k = nd++;
    while (k > 0 && cmp_msg(M, w, pt, dv[k - 1], pt) < 0) { dv[k] = dv[k - 1]; k--; }
    dv[k] = w;
  }
  return nd;
}

// one message record (the bag word without its count) in TLC's field order
template <typename S> VSR_HD void put_msg(const Model& M, S& s, u64 w, u32 pt) {
  const int t = m_type(w);
  const u32 lg = m_lg(w) & 0xFFFFFF;
  put_fcn(s, msg_arity(t));
  put_name(s, "view_number"); put_Int(s, m_view(w));
  put_name(s, "type"); put_Model(s, mv_msgtype(M, t));
  if (t == T_PREPARE) { put_name(s, "message"); put_entry(M, s, (int)(lg & 0xFF), pt); }
  if (t != T_SVC) { put_name(s, "op_number"); put_Int(s, m_op(w)); }
  if (t == T_PREPARE || t == T_SV || t == T_DVC || t == T_NEWSTATE) { put_name(s, "commit_number"); put_Int(s, m_commit(w)); }
  put_name(s, "dest"); put_Int(s, m_dest(w));
  put_name(s, "source"); put_Int(s, m_source(w));
  if (t == T_SV || t == T_DVC) { put_name(s, "log"); put_log(M, s, lg, 1, log_len(lg), pt); }
  if (t == T_DVC) { put_name(s, "last_normal_vn"); put_Int(s, m_lnv(w)); }
  if (t == T_NEWSTATE) {
    put_name(s, "log"); put_log(M, s, lg, m_first_op(w), m_op(w), pt);
    put_name(s, "first_op"); put_Int(s, m_first_op(w));
  }
}

// The `view` value of a record (header + R replica blocks at rec, bag words at bag: either layout) under value permutation pt, fed to sink s.
// view == << rep_state_vars, rep_rec_vars, rep_vc_vars, client_vars, replicas, clients, messages >>      VSR.tla:140-150
template <typename S> VSR_HD void put_view(const Model& M, S& s, const u64* rec, const u64* bag, u32 pt) {
  const int R = M.R, C = M.C, wpr = M.wpr, nmsg = hdr_nmsg(rec[0]);
  auto A = [&](int r) { return rec[1 + (r - 1) * wpr]; };
  auto X = [&](int r, int i) { return blk_x(rec + 1 + (r - 1) * wpr, i); };
  put_fcn(s, 7);
  // ---- 1: rep_state_vars = << rep_status, rep_log, rep_view_number, rep_op_number, rep_peer_op_number, rep_commit_number, rep_client_table, rep_last_normal_view >>
  put_idx(s, 1); put_fcn(s, 8);
  put_idx(s, 1); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Model(s, mv_status(M, a_status(A(r)))); }
  put_idx(s, 2); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); const u32 lg = X(r, 0) & 0xFFFFFF; put_log(M, s, lg, 1, log_len(lg), pt); }
  put_idx(s, 3); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Int(s, a_view(A(r))); }
  put_idx(s, 4); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Int(s, a_op(A(r))); }
  put_idx(s, 5); put_fcn(s, R);
  for (int r = 1; r <= R; r++) { put_idx(s, r); put_fcn(s, R); for (int p = 1; p <= R; p++) { put_idx(s, p); put_Int(s, a_peer(A(r), p)); } }
  put_idx(s, 6); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Int(s, a_commit(A(r))); }
  put_idx(s, 7); put_fcn(s, R);
  for (int r = 1; r <= R; r++) {
    put_idx(s, r); put_fcn(s, C);
    for (int c = 1; c <= C; c++) {
      const int row = a_ctrow(A(r), c);
      put_idx(s, c); put_fcn(s, 3);
      put_name(s, "request_number"); put_Int(s, ct_req(row));
      put_name(s, "op_number"); put_Int(s, ct_op(row));
      put_name(s, "executed"); put_Bool(s, ct_exec(row) != 0);
    }
  }
  put_idx(s, 8); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Int(s, a_lnv(A(r))); }
  // ---- 2: rep_rec_vars = << rep_rec_number, rep_rec_recv >>: constant with RestartEmptyLimit = 0 (all 0, all {})
  put_idx(s, 2); put_fcn(s, 2);
  put_idx(s, 1); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Int(s, 0); }
  put_idx(s, 2); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_set(s, 0); }
  // ---- 3: rep_vc_vars = << rep_svc_recv, rep_dvc_recv, rep_sent_dvc, rep_sent_sv >>
  put_idx(s, 3); put_fcn(s, 4);
  put_idx(s, 1); put_fcn(s, R);
  for (int r = 1; r <= R; r++) {      // SVC records [view_number, type, dest, source] of r's own view: they differ in source only -> by source
    const int mask = a_svcmask(A(r));
    put_idx(s, r); put_set(s, __builtin_popcount((unsigned)mask));
    for (int src = 1; src <= R; src++)
      if ((mask >> (src - 1)) & 1) put_msg(M, s, m_make(T_SVC, a_view(A(r)), r, src, 0, 0, 0, 0, 0), pt);
  }
  put_idx(s, 2); put_fcn(s, R);
  for (int r = 1; r <= R; r++) {
    u64 dv[5];
    const int nd = dvc_set(M, rec + 1 + (r - 1) * wpr, r, pt, dv);
    put_idx(s, r); put_set(s, nd);
    for (int k = 0; k < nd; k++) put_msg(M, s, dv[k], pt);
  }
  put_idx(s, 3); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Bool(s, a_sent_dvc(A(r)) != 0); }
  put_idx(s, 4); put_fcn(s, R); for (int r = 1; r <= R; r++) { put_idx(s, r); put_Bool(s, a_sent_sv(A(r)) != 0); }
  // ---- 4: client_vars = << >>;  5: replicas = 1..R;  6: clients = 1..C
  put_idx(s, 4); put_fcn(s, 0);
  put_idx(s, 5); put_set(s, R); for (int r = 1; r <= R; r++) put_Int(s, r);
  put_idx(s, 6); put_set(s, C); for (int c = 1; c <= C; c++) put_Int(s, c);
  // ---- 7: messages: a function record -> Nat, pairs in the normal order of the records
  put_idx(s, 7); put_fcn(s, nmsg);
  unsigned char ord[MAX_MSGS];
  msg_order(M, bag, nmsg, pt, ord);
  for (int k = 0; k < nmsg && k < MAX_MSGS; k++) {
    const u64 w = bag[ord[k]];
    put_msg(M, s, w, pt);
    put_Int(s, m_count(w));
  }
}

// ---- SYMMETRY: the permutation TLC fingerprints (TLCStateMut.fingerPrint) ------------------------------------------------------------------------
// TLC permutes every state variable and keeps the permuted state that is smallest when the variables are compared one after the other, in the order
// of their declaration (VSR.tla:118-137), by Value.compareTo; the view is evaluated in THAT state.  Of the twenty variables four contain Values:
// rep_log, rep_dvc_recv, messages, aux_client_acked — in this order of declaration; the others compare equal under every permutation.
// state under pa against the same state under pb
VSR_HD int cmp_state(const Model& M, const u64* rec, const u64* bag, u32 pa, u32 pb) {
  const int R = M.R, wpr = M.wpr, nmsg = hdr_nmsg(rec[0]);
  for (int r = 1; r <= R; r++) {                                   // rep_log: a tuple of tuples of entries (the lengths are the same state's)
    const u32 lg = blk_x(rec + 1 + (r - 1) * wpr, 0) & 0xFFFFFF;
    const int c = cmp_log(lg, 1, log_len(lg), pa, lg, 1, log_len(lg), pb);
    if (c) return c;
  }
  for (int r = 1; r <= R; r++) {                                   // rep_dvc_recv: a tuple of sets; sets compare by size, then element by element in normal order
    u64 da[5], db[5];
    const int nd = dvc_set(M, rec + 1 + (r - 1) * wpr, r, pa, da);
    dvc_set(M, rec + 1 + (r - 1) * wpr, r, pb, db);
    for (int k = 0; k < nd; k++) {
      const int c = cmp_msg(M, da[k], pa, db[k], pb);
      if (c) return c;
    }
  }
  {                                                                // messages: FcnRcdValue.compareTo: size, then pair by pair the domain value and the range value
    unsigned char oa[MAX_MSGS], ob[MAX_MSGS];
    msg_order(M, bag, nmsg, pa, oa);
    msg_order(M, bag, nmsg, pb, ob);
    for (int k = 0; k < nmsg && k < MAX_MSGS; k++) {
      const u64 wa = bag[oa[k]], wb = bag[ob[k]];
      const int c = cmp_msg(M, wa, pa, wb, pb);
      if (c) return c;
      if (m_count(wa) != m_count(wb)) return m_count(wa) < m_count(wb) ? -1 : 1;
    }
  }
  {                                                                // aux_client_acked: a function Values -> BOOLEAN over the acknowledged values (FALSE < TRUE)
    int ia[4] = {0, 0, 0, 0}, ib[4] = {0, 0, 0, 0};                // by permuted value index: 0 = not in the domain, 1 = FALSE, 2 = TRUE
    for (int v = 0; v < M.n; v++) {
      const int st = hdr_acked(rec[0], v);
      const int xa = (int)((pa >> (2 * v)) & 3), xb = (int)((pb >> (2 * v)) & 3);
      for (int q = 0; q < 4; q++) { ia[q] = q == xa ? st : ia[q]; ib[q] = q == xb ? st : ib[q]; }
    }
    int ka = 0, kb = 0;                                            // walk the two sorted domains in step (their sizes are equal)
    for (;;) {
      while (ka < 4 && !ia[ka]) ka++;
      while (kb < 4 && !ib[kb]) kb++;
      if (ka >= 4 || kb >= 4) break;
      if (ka != kb) return ka < kb ? -1 : 1;
      if (ia[ka] != ib[kb]) return ia[ka] < ib[kb] ? -1 : 1;
      ka++; kb++;
    }
  }
  return 0;
}
VSR_HD int min_permutation(const Model& M, const u64* rec, const u64* bag) {
  int best = 0;
  for (int i = 1; i < M.np; i++)
    if (cmp_state(M, rec, bag, M.pitab[i], M.pitab[best]) < 0) best = i;
  return best;
}

// TLC's fingerprint of a record: FP64 over the view value of the representative TLC picks under SYMMETRY (the state itself without it)
VSR_HD u64 fingerprint(const Model& M, const u64* rec, const u64* bag, const u64* tab) {
  FpSink s;
  s.fp = IRRED_POLY;
  s.tab = tab;
  put_view(M, s, rec, bag, M.pitab[min_permutation(M, rec, bag)]);
  return s.fp;
}

#if defined(__HIPCC__)
// one lane per record.  refs == nullptr: record p is words[off[p] ..) (device layout, off = n + 1 word offsets);  otherwise refs[p] = a frontier
// reference (word offset << 8 | length, 0 = unused index -> out[p] = 0)
__global__ void __launch_bounds__(256) k_tlc_fingerprints(Model M, const u64* words, const u64* off, const u64* refs, u64 n, u64* out) {
  __shared__ u64 tab[256];
  tab[threadIdx.x] = D_TABLE.t[threadIdx.x];
  __syncthreads();
  for (u64 p = (u64)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (u64)gridDim.x * blockDim.x) {
    u64 pos;
    if (refs) {
      const u64 r = refs[p];
      if (!r) { out[p] = 0; continue; }
      pos = r >> 8;
    } else {
      pos = off[p];
    }
    const u64* rec = words + pos;
    out[p] = fingerprint(M, rec, rec + M.fixed, tab);
  }
}
#endif

}  // namespace tlcfp
}  // namespace vsr

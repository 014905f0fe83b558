// oracle/tlc_fp64.cpp — CPU ORACLE for the TLC-style fingerprint mode (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
//
// What TLC computes as the fingerprint of a VSR.tla state under `VIEW view`, restated from the UNPACKED state of
// vsr_oracle.hpp: the state becomes a generic value tree (ints, booleans, model values, records, tuples, functions, sets), the tree is put
// into TLC's normal form by a generic comparison, serialised by the generic rule of each value kind, and fingerprinted with a BIT-SERIAL
// Rabin division (no byte table).  The product (vsr_tlaplus_amd/csrc/vsr_tlcfp.hpp) walks the packed record with hand-placed field orders and
// a byte table instead: the two share no code; tests/test_tlc_fp64.py requires identical byte streams and fingerprints.
// Every TLC-specific fact below is [TLC-RECALLED] (no TLC, no JVM, no pinned fingerprint anywhere in the reference): see vsr_tlcfp.hpp.
#include <algorithm>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "vsr_oracle.hpp"

using namespace vsr_oracle;

namespace {

// ---- the value tree ------------------------------------------------------------------------------------------------------------
struct V {
  enum Kind { INT, BOOL, MODEL, REC, TUP, FCN, SET } kind = INT;
  int i = 0;                                   // INT value / BOOL / MODEL index / FCN: interval domain start (when dom_interval)
  bool dom_interval = false;                   // FCN whose domain is an integer interval i .. i + elems.size() - 1
  std::vector<std::string> names;              // REC: field names (any order before normalise())
  std::vector<V> elems;                        // REC: field values; TUP / SET: elements; FCN: range values
  std::vector<V> dom;                          // FCN with an explicit domain
};
V Int(int x) { V v; v.kind = V::INT; v.i = x; return v; }
V Bool(bool b) { V v; v.kind = V::BOOL; v.i = b; return v; }
V ModelV(int index) { V v; v.kind = V::MODEL; v.i = index; return v; }
V Tup() { V v; v.kind = V::TUP; return v; }
V Set() { V v; v.kind = V::SET; return v; }
V Rec() { V v; v.kind = V::REC; return v; }
void field(V& r, const char* name, V val) { r.names.push_back(name); r.elems.push_back(std::move(val)); }

// interning order of the field names = order of their first occurrence in VSR.tla (LogEntryType :157-161, message types :163-225,
// client rows :318-321, NewState :533-541)   [SURVEY App. B4]
int name_rank(const std::string& n) {
  static const char* order[] = {"view_number", "operation", "client_id", "request_number", "type", "message", "op_number", "commit_number", "dest",
                                "source", "log", "last_normal_vn", "x", "executed", "first_op"};
  for (int k = 0; k < (int)(sizeof(order) / sizeof(order[0])); k++)
    if (n == order[k]) return k;
  throw RepError("unknown field name " + n);
}

int cmp(const V& a, const V& b);
int cmp_seq(const std::vector<V>& a, const std::vector<V>& b) {
  if (a.size() != b.size()) return a.size() < b.size() ? -1 : 1;
  for (size_t k = 0; k < a.size(); k++) {
    const int c = cmp(a[k], b[k]);
    if (c) return c;
  }
  return 0;
}
// Value.compareTo as far as this model needs it: values of one kind only ever meet values of the same kind
int cmp(const V& a, const V& b) {
  if (a.kind != b.kind) throw RepError("comparison of values of different kinds");
  switch (a.kind) {
    case V::INT: case V::BOOL: case V::MODEL: return a.i < b.i ? -1 : a.i > b.i ? 1 : 0;
    case V::REC: {                              // RecordValue.compareTo: arity, then field by field (normal order) the name, then the value
      if (a.names.size() != b.names.size()) return a.names.size() < b.names.size() ? -1 : 1;
      for (size_t k = 0; k < a.names.size(); k++) {
        const int ra = name_rank(a.names[k]), rb = name_rank(b.names[k]);
        if (ra != rb) return ra < rb ? -1 : 1;
        const int c = cmp(a.elems[k], b.elems[k]);
        if (c) return c;
      }
      return 0;
    }
    case V::TUP: case V::SET: return cmp_seq(a.elems, b.elems);
    case V::FCN: {
      if (a.elems.size() != b.elems.size()) return a.elems.size() < b.elems.size() ? -1 : 1;
      if (a.elems.empty()) return 0;
      if (a.dom_interval != b.dom_interval) throw RepError("comparison of functions with different kinds of domain");
      if (a.dom_interval) {                       // FcnRcdValue.compareTo, two interval domains: the lower bound, then the values
        if (a.i != b.i) return a.i < b.i ? -1 : 1;
        return cmp_seq(a.elems, b.elems);
      }
      for (size_t k = 0; k < a.elems.size(); k++) {   // explicit domains: pair by pair the domain value, then the range value
        int c = cmp(a.dom[k], b.dom[k]);
        if (c) return c;
        c = cmp(a.elems[k], b.elems[k]);
        if (c) return c;
      }
      return 0;
    }
  }
  return 0;
}

void normalise(V& v) {
  for (V& e : v.elems) normalise(e);
  for (V& e : v.dom) normalise(e);
  if (v.kind == V::REC && !v.names.empty()) {                        // fields by the interning order of their names
    std::vector<size_t> idx(v.names.size());
    for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
    std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return name_rank(v.names[x]) < name_rank(v.names[y]); });
    std::vector<std::string> n2;
    std::vector<V> e2;
    for (size_t k : idx) { n2.push_back(v.names[k]); e2.push_back(v.elems[k]); }
    v.names.swap(n2);
    v.elems.swap(e2);
  } else if (v.kind == V::SET) {
    std::sort(v.elems.begin(), v.elems.end(), [](const V& x, const V& y) { return cmp(x, y) < 0; });
  } else if (v.kind == V::FCN && !v.dom_interval) {   // pairs by the domain value
    std::vector<size_t> idx(v.dom.size());
    for (size_t k = 0; k < idx.size(); k++) idx[k] = k;
    std::sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return cmp(v.dom[x], v.dom[y]) < 0; });
    std::vector<V> d2, e2;
    for (size_t k : idx) { d2.push_back(v.dom[k]); e2.push_back(v.elems[k]); }
    v.dom.swap(d2);
    v.elems.swap(e2);
  }
}

// ---- serialisation (Value.fingerPrint of each kind) ------------------------------------------------------------------------------
enum { TAG_BOOL = 0, TAG_INT = 1, TAG_STRING = 3, TAG_SETENUM = 5, TAG_FCNRCD = 9, TAG_MODEL = 21 };
typedef std::vector<unsigned char> Bytes;
void b_int(Bytes& o, int x) { for (int k = 0; k < 4; k++) o.push_back((unsigned char)(((unsigned)x >> (8 * k)) & 0xFF)); }
void ser(const V& v, Bytes& o) {
  switch (v.kind) {
    case V::INT: o.push_back(TAG_INT); b_int(o, v.i); break;
    case V::BOOL: o.push_back(TAG_BOOL); o.push_back(v.i ? 't' : 'f'); break;
    case V::MODEL: o.push_back(TAG_MODEL); b_int(o, v.i); break;
    case V::REC:
      o.push_back(TAG_FCNRCD); b_int(o, (int)v.names.size());
      for (size_t k = 0; k < v.names.size(); k++) {
        o.push_back(TAG_STRING); b_int(o, (int)v.names[k].size());
        for (char ch : v.names[k]) o.push_back((unsigned char)ch);
        ser(v.elems[k], o);
      }
      break;
    case V::TUP:
      o.push_back(TAG_FCNRCD); b_int(o, (int)v.elems.size());
      for (size_t k = 0; k < v.elems.size(); k++) { o.push_back(TAG_INT); b_int(o, (int)k + 1); ser(v.elems[k], o); }
      break;
    case V::FCN:
      o.push_back(TAG_FCNRCD); b_int(o, (int)v.elems.size());
      for (size_t k = 0; k < v.elems.size(); k++) {
        if (v.dom_interval) { o.push_back(TAG_INT); b_int(o, v.i + (int)k); }
        else ser(v.dom[k], o);
        ser(v.elems[k], o);
      }
      break;
    case V::SET:
      o.push_back(TAG_SETENUM); b_int(o, (int)v.elems.size());
      for (const V& e : v.elems) ser(e, o);
      break;
  }
}

// ---- FP64, bit by bit: the fingerprint register holds a polynomial over GF(2), bit 63 = x^0 (tlc2.util.FP64's reflected order).
// Extending by a byte b = xor b into the eight highest-degree coefficients (the low byte), then eight times: multiply by x, reduce by P.
const u64 IRRED = 0x911498AE0E66BAD6ULL;
u64 fp64_extend_byte(u64 fp, unsigned char b) {
  fp ^= (u64)b;
  for (int k = 0; k < 8; k++) fp = (fp >> 1) ^ ((fp & 1) ? IRRED : 0);
  return fp;
}
u64 fp64(const Bytes& bytes) {
  u64 fp = IRRED;                                // FP64.New()
  for (unsigned char b : bytes) fp = fp64_extend_byte(fp, b);
  return fp;
}

// ---- the view of a state as a value (VSR.tla:140-150) ----------------------------------------------------------------------------
int mv_index(const Params& P, const char* name) {   // creation order of the model values in the cfg: the Values, then VSR.cfg:9-24
  static const char* rest[] = {"Normal", "ViewChange", "Recovering", "RequestMsg", "ReplyMsg", "PrepareMsg", "PrepareOkMsg", "CommitMsg",
                               "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg", "GetStateMsg", "NewStateMsg", "RecoveryMsg",
                               "RecoveryResponseMsg", "Nil"};
  for (int k = 0; k < 16; k++)
    if (!std::strcmp(name, rest[k])) return P.n + k;
  throw RepError("unknown model value");
}
const char* type_name(int t) {
  switch (t) {
    case T_SVC: return "StartViewChangeMsg";
    case T_PREPARE: return "PrepareMsg";
    case T_PREPAREOK: return "PrepareOkMsg";
    case T_DVC: return "DoViewChangeMsg";
    case T_SV: return "StartViewMsg";
    case T_GETSTATE: return "GetStateMsg";
    case T_NEWSTATE: return "NewStateMsg";
  }
  throw RepError("message type outside the live set");
}
V entry_value(const Entry& e) {
  V r = Rec();
  field(r, "view_number", Int(e.view));
  field(r, "operation", ModelV(e.op));            // value index = model value index (the Values come first in the cfg)
  field(r, "client_id", Int(e.client));
  field(r, "request_number", Int(e.req));
  return r;
}
V seq_value(const Log& l) {                        // a TLA+ sequence: a tuple
  V t = Tup();
  for (int k = l.lo; k <= l.hi && l.len() > 0; k++) t.elems.push_back(entry_value(l.e[k]));
  return t;
}
V msg_value(const Params& P, const Msg& m) {
  V r = Rec();
  field(r, "type", ModelV(mv_index(P, type_name(m.type))));
  field(r, "view_number", Int(m.view));
  field(r, "dest", Int(m.dest));
  field(r, "source", Int(m.source));
  switch (m.type) {
    case T_SVC: break;
    case T_PREPARE:
      field(r, "message", entry_value(m.entry));
      field(r, "op_number", Int(m.op));
      field(r, "commit_number", Int(m.commit));
      break;
    case T_PREPAREOK: case T_GETSTATE: field(r, "op_number", Int(m.op)); break;
    case T_DVC:
      field(r, "log", seq_value(m.log));
      field(r, "last_normal_vn", Int(m.lnv));
      field(r, "op_number", Int(m.op));
      field(r, "commit_number", Int(m.commit));
      break;
    case T_SV:
      field(r, "log", seq_value(m.log));
      field(r, "op_number", Int(m.op));
      field(r, "commit_number", Int(m.commit));
      break;
    case T_NEWSTATE: {
      V f;                                          // [on \in first_op .. op_number |-> rep_log[r][on]]   VSR.tla:535-536
      f.kind = V::FCN;
      f.dom_interval = true;
      f.i = m.first_op;
      for (int k = m.first_op; k <= m.op; k++) f.elems.push_back(entry_value(m.log.e[k]));
      field(r, "log", f);
      field(r, "first_op", Int(m.first_op));
      field(r, "op_number", Int(m.op));
      field(r, "commit_number", Int(m.commit));
      break;
    }
    default: throw RepError("message type outside the live set");
  }
  return r;
}
template <typename F> V per_replica(const Params& P, F f) {   // [r \in replicas |-> f(r)]: a function over 1..R = a tuple
  V t = Tup();
  for (int r = 1; r <= P.R; r++) t.elems.push_back(f(r));
  return t;
}
V interval_set(int lo, int hi) { V s = Set(); for (int k = lo; k <= hi; k++) s.elems.push_back(Int(k)); return s; }

// the state variables by name (VSR.tla:118-137), each in normal form
struct Vars { std::map<std::string, V> v; };
Vars state_vars(const Params& P, const State& s) {
  Vars o;
  o.v["replicas"] = interval_set(1, P.R);
  o.v["clients"] = interval_set(1, P.C);
  o.v["rep_status"] = per_replica(P, [&](int r) { return ModelV(mv_index(P, s.rep[r].status == Normal ? "Normal" : s.rep[r].status == ViewChange ? "ViewChange" : "Recovering")); });
  o.v["rep_log"] = per_replica(P, [&](int r) { return seq_value(s.rep[r].log); });
  o.v["rep_view_number"] = per_replica(P, [&](int r) { return Int(s.rep[r].view); });
  o.v["rep_op_number"] = per_replica(P, [&](int r) { return Int(s.rep[r].op); });
  o.v["rep_peer_op_number"] = per_replica(P, [&](int r) { V t = Tup(); for (int p = 1; p <= P.R; p++) t.elems.push_back(Int(s.rep[r].peer_op[p])); return t; });
  o.v["rep_commit_number"] = per_replica(P, [&](int r) { return Int(s.rep[r].commit); });
  o.v["rep_client_table"] = per_replica(P, [&](int r) {
    V t = Tup();
    for (int c = 1; c <= P.C; c++) {
      V row = Rec();
      field(row, "executed", Bool(s.rep[r].ct[c].exec));
      field(row, "op_number", Int(s.rep[r].ct[c].op));
      field(row, "request_number", Int(s.rep[r].ct[c].req));
      t.elems.push_back(row);
    }
    return t;
  });
  o.v["rep_last_normal_view"] = per_replica(P, [&](int r) { return Int(s.rep[r].lnv); });
  o.v["rep_rec_number"] = per_replica(P, [&](int) { return Int(0); });          // never written (RestartEmptyLimit = 0)
  o.v["rep_rec_recv"] = per_replica(P, [&](int) { return Set(); });
  o.v["rep_svc_recv"] = per_replica(P, [&](int r) { V st = Set(); for (const Msg& m : s.rep[r].svc_recv) st.elems.push_back(msg_value(P, m)); return st; });
  o.v["rep_dvc_recv"] = per_replica(P, [&](int r) { V st = Set(); for (const Msg& m : s.rep[r].dvc_recv) st.elems.push_back(msg_value(P, m)); return st; });
  o.v["rep_sent_dvc"] = per_replica(P, [&](int r) { return Bool(s.rep[r].sent_dvc); });
  o.v["rep_sent_sv"] = per_replica(P, [&](int r) { return Bool(s.rep[r].sent_sv); });
  V msgs;
  msgs.kind = V::FCN;
  for (const auto& mc : s.messages) { msgs.dom.push_back(msg_value(P, mc.first)); msgs.elems.push_back(Int(mc.second)); }
  o.v["messages"] = msgs;
  o.v["aux_svc"] = Int(s.aux_svc);
  o.v["aux_restart"] = Int(0);
  V acked;                                                                        // a function over the acknowledged-or-pending values (VSR.tla:377, :474)
  acked.kind = V::FCN;
  for (int v = 0; v < P.n; v++)
    if (s.acked[v]) { acked.dom.push_back(ModelV(v)); acked.elems.push_back(Bool(s.acked[v] == 2)); }
  o.v["aux_client_acked"] = acked;
  for (auto& kv : o.v) normalise(kv.second);
  return o;
}

// view == << rep_state_vars, rep_rec_vars, rep_vc_vars, client_vars, replicas, clients, messages >>      VSR.tla:140-150
V view_value(const Params& P, const State& s) {
  Vars o = state_vars(P, s);
  auto tup = [&](std::initializer_list<const char*> names) { V t = Tup(); for (const char* n : names) t.elems.push_back(o.v.at(n)); return t; };
  V view = Tup();
  view.elems.push_back(tup({"rep_status", "rep_log", "rep_view_number", "rep_op_number", "rep_peer_op_number", "rep_commit_number", "rep_client_table", "rep_last_normal_view"}));
  view.elems.push_back(tup({"rep_rec_number", "rep_rec_recv"}));
  view.elems.push_back(tup({"rep_svc_recv", "rep_dvc_recv", "rep_sent_dvc", "rep_sent_sv"}));
  view.elems.push_back(Tup());                                                      // client_vars == << >>
  view.elems.push_back(o.v.at("replicas"));
  view.elems.push_back(o.v.at("clients"));
  view.elems.push_back(o.v.at("messages"));
  return view;
}

// TLCStateMut.fingerPrint under SYMMETRY: the permuted states compare variable by variable in the order of declaration (VSR.tla:118-137)
int cmp_states(const Vars& a, const Vars& b) {
  static const char* decl[] = {"replicas", "rep_status", "rep_log", "rep_view_number", "rep_op_number", "rep_commit_number", "rep_peer_op_number",
                               "rep_client_table", "rep_last_normal_view", "rep_svc_recv", "rep_dvc_recv", "rep_sent_dvc", "rep_sent_sv", "rep_rec_number",
                               "rep_rec_recv", "clients", "messages", "aux_svc", "aux_restart", "aux_client_acked"};
  for (const char* n : decl) {
    const int c = cmp(a.v.at(n), b.v.at(n));
    if (c) return c;
  }
  return 0;
}

std::string g_tlc_err;

}  // namespace

extern "C" {

const char* orc_tlc_last_error() { return g_tlc_err.c_str(); }

// the byte stream TLC's fingerprint of the state's view is taken over, under the value permutation number `perm` (identity = 0, in
// std::next_permutation order); returns its length (the first `cap` bytes are stored), < 0 on error
long long orc_tlc_view_bytes(const int* params, const u64* rec, int perm, unsigned char* out, long long cap) {
  try {
    Params P = params_from_array(params);
    State s = decode(P, rec, nullptr);
    int pi[4] = {0, 1, 2, 3};
    for (int k = 0; k < perm; k++)
      if (!std::next_permutation(pi, pi + P.n)) throw RepError("no such permutation");
    State t = permute(P, s, pi);
    Bytes b;
    ser(view_value(P, t), b);
    for (long long k = 0; k < (long long)b.size() && k < cap; k++) out[k] = b[k];
    return (long long)b.size();
  } catch (const std::exception& e) {
    g_tlc_err = e.what();
    return -1;
  }
}

// FP64 of the view of the state TLC picks: the state itself without symmetry; with it the permuted state that is smallest by compareTo over all
// variables (see vsr_tlcfp.hpp (4)).  *perm_out (may be null) = the number of that permutation.
int orc_tlc_fingerprint(const int* params, const u64* rec, u64* fp_out, int* perm_out) {
  try {
    Params P = params_from_array(params);
    State s = decode(P, rec, nullptr);
    int pi[4] = {0, 1, 2, 3};
    State best = s;
    Vars best_vars = state_vars(P, s);
    int k = 0, best_k = 0;
    while (P.symmetry && std::next_permutation(pi, pi + P.n)) {
      k++;
      State t = permute(P, s, pi);
      Vars tv = state_vars(P, t);
      if (cmp_states(tv, best_vars) < 0) { best = t; best_vars = tv; best_k = k; }
    }
    Bytes b;
    ser(view_value(P, best), b);
    *fp_out = fp64(b);
    if (perm_out) *perm_out = best_k;
    return 0;
  } catch (const std::exception& e) {
    g_tlc_err = e.what();
    return -1;
  }
}

// the wire record of the state permuted by permutation number `perm` (another member of the same symmetry class); returns its length in words
long long orc_tlc_permute_record(const int* params, const u64* rec, int perm, u64* out, long long cap) {
  try {
    Params P = params_from_array(params);
    State s = decode(P, rec, nullptr);
    int pi[4] = {0, 1, 2, 3};
    for (int k = 0; k < perm; k++)
      if (!std::next_permutation(pi, pi + P.n)) throw RepError("no such permutation");
    std::vector<u64> w;
    encode(P, permute(P, s, pi), w);
    if ((long long)w.size() > cap) throw RepError("buffer too small");
    for (size_t k = 0; k < w.size(); k++) out[k] = w[k];
    return (long long)w.size();
  } catch (const std::exception& e) {
    g_tlc_err = e.what();
    return -1;
  }
}

// FP64 of an arbitrary byte string from FP64.New() (known-answer tests of the arithmetic against the Python reference)
u64 orc_fp64_bytes(const unsigned char* bytes, long long n) {
  u64 fp = IRRED;
  for (long long k = 0; k < n; k++) fp = fp64_extend_byte(fp, bytes[k]);
  return fp;
}

}  // extern "C"

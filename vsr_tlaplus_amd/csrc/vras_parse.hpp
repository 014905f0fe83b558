// vras_parse.hpp — TLC's printed states of the two analysis models -> wire records: the reader half of vrst_format.hpp /
// vras_format.hpp (a TLC trace expression, a `tlc2.TLC -dump` file or console "State k:" blocks of VR_STATE_TRANSFER.tla /
// VR_APP_STATE.tla).  The text parser (TlcParser, TVal) is vsr_parse.hpp's; this file holds the encoder for the record layouts
// of vrst_actions.hpp (model 1: one word per replica) and vras_actions.hpp (model 2: a second word with rep_recv_dvc).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "vras_actions.hpp"
#include "vsr_parse.hpp"

namespace vsr {

class StateEncoder2 {
 public:
  StateEncoder2(const Model& M, const std::vector<std::string>& vals) : M_(M), vals_(vals) {}
  std::string error;

  bool encode(const TVal& st, std::vector<u64>* out) {
    if (st.kind != TVal::REC) return fail("a state must be a record of variables");
    const Model& M = M_;
    const bool app = M.model_id == 2;                           // VR_APP_STATE: rep_app_state, rep_recv_dvc, rep_rec_*, aux_restart
    std::vector<u64> rec((size_t)M.h0, 0);
    for (int r = 1; r <= M.R; r++) rec[(size_t)aidx(r)] = a_set_lnv(a_set_view(a_set_status(0, vrst::ST2_NORMAL), 1), 1);   // Init
    u64 hdr = 0;
    std::vector<u64> bag;
    int app_len[6] = {-1, -1, -1, -1, -1, -1};                        // Len(rep_app_state[r]) as read, -1 = the text does not give it
    static const char* const ORDER[] = {"rep_view_number", "rep_status", "rep_op_number", "rep_commit_number", "rep_last_normal_view",
                                        "rep_sent_dvc", "rep_sent_sv", "rep_peer_op_number", "rep_log", "rep_app_state", "rep_recv_dvc",
                                        "rep_rec_number", "rep_rec_recv", "no_progress", "no_progress_ctr", "aux_svc", "aux_client_acked",
                                        "aux_restart", "messages", "replicas"};
    for (const auto& f : st.fields) {
      bool known = f.first == "_TEAction";
      for (const char* o : ORDER) known = known || f.first == o;
      if (!known) return fail("unknown variable " + f.first);
      if (!app && (f.first == "rep_app_state" || f.first == "rep_recv_dvc" || f.first == "rep_rec_number" || f.first == "rep_rec_recv" ||
                   f.first == "aux_restart"))
        return fail(f.first + " is not a variable of VR_STATE_TRANSFER");
    }
    for (const char* name : ORDER) {
      const TVal* v = st.field(name);
      if (!v) continue;
      const std::string n = name;
      long x;
      if (n == "aux_svc") {
        if (!integer(*v, 0, 7, n, &x)) return false;
        hdr |= (u64)x << 8;
      } else if (n == "no_progress_ctr") {
        if (!integer(*v, 0, 7, n, &x)) return false;
        hdr |= (u64)x << 20;
      } else if (n == "aux_restart") {
        if (!integer(*v, 0, 0, n, &x)) return false;             // never written (VR_APP_STATE.tla:91)
      } else if (n == "replicas") {
        if (v->kind != TVal::RANGE || v->i != 1 || v->j != M.R) return fail("replicas does not match ReplicaCount");
      } else if (n == "aux_client_acked") {
        if (v->kind == TVal::SEQ && v->items.empty()) continue;
        if (v->kind != TVal::FCN) return fail("aux_client_acked must be a function");
        for (const auto& kv : v->pairs) {
          int vi, b;
          if (!value_index(kv.first, &vi) || !boolean(kv.second, &b)) return false;
          hdr = hdr_set_acked(hdr, vi, b ? 2 : 1);
        }
      } else if (n == "messages") {
        if (v->kind == TVal::SEQ && v->items.empty()) continue;
        if (v->kind != TVal::FCN) return fail("messages must be a function (bag)");
        for (const auto& kv : v->pairs) {
          u64 w;
          long cnt;
          if (!message(kv.first, &w) || !integer(kv.second, 0, 3, "delivery count", &cnt)) return false;
          bag.push_back(w | ((u64)cnt << 21));
        }
      } else {                                                  // per-replica variables
        if (v->kind != TVal::SEQ || (int)v->items.size() != M.R) return fail(n + " must be a sequence of ReplicaCount elements");
        for (int r = 1; r <= M.R; r++) {
          const TVal& e = v->items[(size_t)(r - 1)];
          u64& A = rec[(size_t)aidx(r)];
          long k;
          int b;
          if (n == "rep_view_number") { if (!integer(e, 0, 7, n, &k)) return false; A = a_set_view(A, (int)k); }
          else if (n == "rep_op_number") { if (!integer(e, 0, 3, n, &k)) return false; A = a_set_op(A, (int)k); }
          else if (n == "rep_commit_number") { if (!integer(e, 0, 3, n, &k)) return false; A = a_set_commit(A, (int)k); }
          else if (n == "rep_last_normal_view") { if (!integer(e, 0, 7, n, &k)) return false; A = a_set_lnv(A, (int)k); }
          else if (n == "rep_sent_dvc") { if (!boolean(e, &b)) return false; A = a_set_sent_dvc(A, b); }
          else if (n == "rep_sent_sv") { if (!boolean(e, &b)) return false; A = a_set_sent_sv(A, b); }
          else if (n == "no_progress") { if (!boolean(e, &b)) return false; A = a_set(A, 14, 1, b); }
          else if (n == "rep_rec_number") { if (!integer(e, 0, 0, n, &k)) return false; }
          else if (n == "rep_rec_recv") { if (e.kind != TVal::SET || !e.items.empty()) return fail("rep_rec_recv must be empty (never written)"); }
          else if (n == "rep_status") {
            if (e.kind != TVal::ID) return fail("rep_status must be a model value");
            if (e.s == "Normal") A = a_set_status(A, vrst::ST2_NORMAL);
            else if (e.s == "ViewChange") A = a_set_status(A, vrst::ST2_VIEWCHANGE);
            else if (e.s == "StateTransfer") A = a_set_status(A, vrst::ST2_STATETRANSFER);
            else return fail("rep_status " + e.s + " is not a status of this model");
          } else if (n == "rep_peer_op_number") {
            if (e.kind != TVal::SEQ || (int)e.items.size() != M.R) return fail(n + ": one entry per replica expected");
            for (int p = 1; p <= M.R; p++) {
              if (!integer(e.items[(size_t)(p - 1)], 0, 3, n, &k)) return false;
              A = vrst::b_set_peer(A, p, (int)k);
            }
          } else if (n == "rep_log") {
            u32 bits;
            if (!log_bits(e, 1, &bits)) return false;
            A = vrst::b_set_log(A, bits);
          } else if (n == "rep_app_state") {
            if (e.kind != TVal::SEQ || e.items.size() > 3) return fail("rep_app_state: a sequence of at most 3 entries expected");
            for (size_t i = 0; i < e.items.size(); i++) {
              int vi;
              if (!entry(e.items[i], &vi)) return false;
              A = a_set(A, 34 + 2 * (int)i, 2, vi);
            }
            app_len[r] = (int)e.items.size();
          } else if (n == "rep_recv_dvc") {
            if (e.kind != TVal::SET) return fail("rep_recv_dvc: a set of DoViewChange records expected");
            u64& B = rec[(size_t)aidx(r) + 1];
            for (const TVal& mv : e.items) {
              u64 w;
              if (!message(mv, &w)) return false;
              if (m_type(w) != T_DVC || m_dest(w) != r) return fail("rep_recv_dvc holds a record that is not a DoViewChange to this replica");
              int err = 0;
              B = vras::d_add(B, m_view(w), m_source(w), vras::d_make_slot(m_lnv(w), m_op(w), m_commit(w), vrst::bytes_to_blog(m_lg(w) & 0xFFFFFF)), &err);
              if (err) return fail("rep_recv_dvc holds members of two views or two records from one source");
            }
          }
        }
      }
    }
    for (int r = 1; r <= M.R; r++) {                            // what the packed record relies on (vrst_ / vras_actions.hpp)
      const u64 A = rec[(size_t)aidx(r)];
      if (app_len[r] >= 0 && app_len[r] != a_commit(A)) return fail("Len(rep_app_state[r]) differs from rep_commit_number[r]");
      if (a_op(A) != vrst::blog_len(vrst::b_log(A))) return fail("rep_op_number[r] differs from Len(rep_log[r])");
    }
    if ((int)bag.size() > 255) return fail("more than 255 distinct messages");
    std::sort(bag.begin(), bag.end());
    for (size_t k = 1; k < bag.size(); k++)
      if ((bag[k] & KEYMASK) == (bag[k - 1] & KEYMASK)) return fail("the same message occurs twice in the bag");
    rec[0] = hdr_set_nmsg(hdr, (int)bag.size());
    rec.insert(rec.end(), bag.begin(), bag.end());
    *out = std::move(rec);
    return true;
  }

 private:
  int aidx(int r) const { return 1 + (r - 1) * M_.wpr; }
  bool fail(const std::string& msg) {
    if (error.empty()) error = msg;
    return false;
  }
  bool integer(const TVal& v, long lo, long hi, const std::string& what, long* out) {
    if (v.kind != TVal::INT) return fail(what + ": an integer expected");
    if (v.i < lo || v.i > hi) return fail(what + " = " + std::to_string(v.i) + " is outside the range the packed record holds (" +
                                          std::to_string(lo) + ".." + std::to_string(hi) + ")");
    *out = v.i;
    return true;
  }
  bool boolean(const TVal& v, int* out) {
    if (v.kind != TVal::ID || (v.s != "TRUE" && v.s != "FALSE")) return fail("TRUE / FALSE expected");
    *out = v.s == "TRUE";
    return true;
  }
  bool value_index(const TVal& v, int* out) {
    if (v.kind == TVal::ID)
      for (int k = 0; k < M_.n && k < (int)vals_.size(); k++)
        if (vals_[(size_t)k] == v.s) { *out = k; return true; }
    return fail("'" + v.s + "' is not an element of Values");
  }
  bool entry(const TVal& v, int* out) {                         // LogEntryType == [operation: Values]
    if (v.kind != TVal::REC || v.fields.size() != 1 || !v.field("operation")) return fail("log entry: [operation |-> value] expected");
    return value_index(*v.field("operation"), out);
  }
  // a replica log / rep_recv_dvc log as 3-bit entries (1 | value<<1), a message log as bytes (1 | value<<3): `shift` bits per entry
  bool log_generic(const TVal& v, int first, int shift, int vshift, u32* out) {
    u32 lg = 0;
    if (v.kind == TVal::SEQ) {
      if ((int)v.items.size() + first - 1 > 3) return fail("log longer than 3 entries");
      for (size_t k = 0; k < v.items.size(); k++) {
        int vi;
        if (!entry(v.items[k], &vi)) return false;
        lg |= (1u | ((u32)vi << vshift)) << (shift * ((int)k + first - 1));
      }
    } else if (v.kind == TVal::FCN) {
      for (const auto& kv : v.pairs) {
        long opn;
        int vi;
        if (!integer(kv.first, 1, 3, "log index", &opn) || !entry(kv.second, &vi)) return false;
        lg |= (1u | ((u32)vi << vshift)) << (shift * (opn - 1));
      }
    } else {
      return fail("a log must be a sequence or a function");
    }
    *out = lg;
    return true;
  }
  bool log_bits(const TVal& v, int first, u32* out) { return log_generic(v, first, 3, 1, out); }
  bool log_bytes(const TVal& v, int first, u32* out) { return log_generic(v, first, 8, 3, out); }
  bool message(const TVal& v, u64* out) {                       // the message record types of the two modules
    if (v.kind != TVal::REC || !v.field("type") || v.field("type")->kind != TVal::ID) return fail("message: a record with a type expected");
    const std::string& ty = v.field("type")->s;
    int t = ty == "StartViewChangeMsg" ? T_SVC : ty == "PrepareMsg" ? T_PREPARE : ty == "PrepareOkMsg" ? T_PREPAREOK
            : ty == "DoViewChangeMsg" ? T_DVC : ty == "StartViewMsg" ? T_SV : ty == "GetStateMsg" ? T_GETSTATE
            : ty == "NewStateMsg" ? T_NEWSTATE : 0;
    if (!t) return fail("message type " + ty + " is not a type of this model");
    auto num = [&](const char* name, long lo, long hi, long* out_) {
      const TVal* f = v.field(name);
      if (!f) return fail(std::string("message ") + ty + ": field " + name + " missing");
      return integer(*f, lo, hi, name, out_);
    };
    long view = 0, dest = 0, source = 0, op = 0, commit = 0, lnv = 0, fo = 0;
    u32 lg = 0;
    if (!num("view_number", 0, 7, &view) || !num("source", 1, M_.R, &source)) return false;
    const TVal* d = v.field("dest");
    if (!d) return fail("message " + ty + ": field dest missing");
    if (d->kind == TVal::ID && d->s == "AnyDest") dest = vrst::ANYDEST;
    else if (!integer(*d, 1, M_.R, "dest", &dest)) return false;
    size_t want = 4;
    if (t != T_SVC) { if (!num("op_number", 0, 3, &op)) return false; want++; }
    if (t == T_PREPARE || t == T_DVC || t == T_SV || t == T_NEWSTATE) { if (!num("commit_number", 0, 3, &commit)) return false; want++; }
    if (t == T_PREPARE) {
      int vi;
      if (!v.field("message") || !entry(*v.field("message"), &vi)) return fail("PrepareMsg: field message missing or malformed");
      lg = 1u | ((u32)vi << 3);
      want++;
    }
    if (t == T_DVC) { if (!num("last_normal_vn", 0, 7, &lnv)) return false; want++; }
    if (t == T_NEWSTATE) { if (!num("first_op", 1, 3, &fo)) return false; want++; }
    if (t == T_DVC || t == T_SV || t == T_NEWSTATE) {
      if (!v.field("log") || !log_bytes(*v.field("log"), t == T_NEWSTATE ? (int)fo : 1, &lg)) return fail("message " + ty + ": field log missing or malformed");
      want++;
    }
    if (v.fields.size() != want) return fail("message " + ty + ": unexpected fields");
    *out = m_make(t, (int)view, (int)dest, (int)source, (int)op, (int)commit, (int)lnv, (int)fo, lg);
    return true;
  }

  const Model& M_;
  const std::vector<std::string>& vals_;
};

inline bool encode_state_analysis_models(const Model& M, const std::vector<std::string>& vals, const TVal& st, std::vector<u64>* out,
                                         std::string* err) {
  StateEncoder2 enc(M, vals);
  if (enc.encode(st, out)) return true;
  *err = enc.error;
  return false;
}

}  // namespace vsr

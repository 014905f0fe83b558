// vsr_parse.hpp — host-side reader: TLC's value syntax -> wire-layout records.  The inverse of vsr_format.hpp.
//
// Accepts what TLC writes for behaviours of VSR.tla:
//   (a) a trace expression: << [ _TEAction |-> [position |-> k, name |-> "...", location |-> "..."], var |-> value, ... ], ... >>
//       (the format of /root/reference/state_transfer_violation_trace.txt, trace:1-30),
//   (b) one state record  [ var |-> value, ... ]  (what vsrmc_model_format_state prints),
//   (c) TLC's console form:  State k: <Action line ..>  followed by  /\ var = value  conjuncts.
// Variables that the text does not mention keep their Init value (the reference's trace predates rep_rec_number,
// rep_rec_recv and aux_restart — SURVEY §8c); `clients` / `replicas` are checked against the model, not stored.
// The bag words of a parsed record are sorted ascending (any order denotes the same bag; this one is canonical).
#pragma once
#include <algorithm>
#include <cctype>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "vsr_model.hpp"

namespace vsr {

struct TVal {
  enum Kind { INT, STR, ID, SEQ, SET, REC, FCN, RANGE } kind = INT;
  long i = 0, j = 0;                                    // INT: i;  RANGE: i..j
  std::string s;                                        // STR / ID
  std::vector<TVal> items;                              // SEQ / SET
  std::vector<std::pair<std::string, TVal>> fields;     // REC
  std::vector<std::pair<TVal, TVal>> pairs;             // FCN  (k :> v @@ ...)
  const TVal* field(const std::string& name) const {
    for (const auto& f : fields)
      if (f.first == name) return &f.second;
    return nullptr;
  }
};

// encoder of the two analysis models' states (vras_parse.hpp)
inline bool encode_state_analysis_models(const Model& M, const std::vector<std::string>& vals, const TVal& st, std::vector<u64>* out,
                                         std::string* err);

class TlcParser {
 public:
  explicit TlcParser(const std::string& text) : t_(text) {}
  std::string error;

  void skip() {
    while (p_ < t_.size()) {
      if (std::isspace((unsigned char)t_[p_])) p_++;
      else if (t_.compare(p_, 2, "\\*") == 0) while (p_ < t_.size() && t_[p_] != '\n') p_++;
      else break;
    }
  }
  bool eof() { skip(); return p_ >= t_.size(); }
  bool peek(const char* tok) { skip(); return t_.compare(p_, std::strlen(tok), tok) == 0; }
  bool eat(const char* tok) {
    if (!peek(tok)) return false;
    p_ += std::strlen(tok);
    return true;
  }
  bool expect(const char* tok) {
    if (eat(tok)) return true;
    return fail(std::string("expected '") + tok + "'");
  }
  bool fail(const std::string& msg) {
    if (error.empty()) {
      size_t line = 1 + (size_t)std::count(t_.begin(), t_.begin() + (long)std::min(p_, t_.size()), '\n');
      error = msg + " at line " + std::to_string(line) + " near '" + t_.substr(p_, 24) + "'";
    }
    return false;
  }
  bool ident(std::string* out) {
    skip();
    size_t q = p_;
    while (q < t_.size() && (std::isalnum((unsigned char)t_[q]) || t_[q] == '_')) q++;
    if (q == p_) return fail("expected an identifier");
    *out = t_.substr(p_, q - p_);
    p_ = q;
    return true;
  }

  bool value(TVal* v) {
    if (!atom(v)) return false;
    if (v->kind == TVal::INT && peek("..")) {             // a..b
      p_ += 2;
      TVal hi;
      if (!atom(&hi) || hi.kind != TVal::INT) return fail("expected an integer after '..'");
      v->kind = TVal::RANGE;
      v->j = hi.i;
    }
    return true;
  }

  // conjunct list  /\ var = value /\ ...   up to the end of the text (console form)
  bool conjuncts(TVal* rec) {
    rec->kind = TVal::REC;
    while (eat("/\\")) {
      std::string name;
      TVal v;
      if (!ident(&name) || !expect("=") || !value(&v)) return false;
      rec->fields.emplace_back(name, std::move(v));
    }
    return true;
  }

 private:
  bool atom(TVal* v) {
    skip();
    if (p_ >= t_.size()) return fail("unexpected end of text");
    const char c = t_[p_];
    if (eat("<<")) {
      v->kind = TVal::SEQ;
      if (eat(">>")) return true;
      do {
        TVal x;
        if (!value(&x)) return false;
        v->items.push_back(std::move(x));
      } while (eat(","));
      return expect(">>");
    }
    if (c == '{') {
      p_++;
      v->kind = TVal::SET;
      if (eat("}")) return true;
      do {
        TVal x;
        if (!value(&x)) return false;
        v->items.push_back(std::move(x));
      } while (eat(","));
      return expect("}");
    }
    if (c == '[') {
      p_++;
      v->kind = TVal::REC;
      if (eat("]")) return true;
      do {
        std::string name;
        TVal x;
        if (!ident(&name) || !expect("|->") || !value(&x)) return false;
        v->fields.emplace_back(name, std::move(x));
      } while (eat(","));
      return expect("]");
    }
    if (c == '(') {
      p_++;
      v->kind = TVal::FCN;
      do {
        TVal k, x;
        if (!value(&k) || !expect(":>") || !value(&x)) return false;
        v->pairs.emplace_back(std::move(k), std::move(x));
      } while (eat("@@"));
      return expect(")");
    }
    if (c == '"') {
      size_t q = t_.find('"', p_ + 1);
      if (q == std::string::npos) return fail("unterminated string");
      v->kind = TVal::STR;
      v->s = t_.substr(p_ + 1, q - p_ - 1);
      p_ = q + 1;
      return true;
    }
    if (std::isdigit((unsigned char)c) || (c == '-' && p_ + 1 < t_.size() && std::isdigit((unsigned char)t_[p_ + 1]))) {
      char* end = nullptr;
      v->kind = TVal::INT;
      v->i = std::strtol(t_.c_str() + p_, &end, 10);
      p_ = (size_t)(end - t_.c_str());
      return true;
    }
    v->kind = TVal::ID;
    return ident(&v->s);
  }

  const std::string& t_;
  size_t p_ = 0;
};

// ---- structured value -> wire record -------------------------------------------------------------------------------
class StateEncoder {
 public:
  StateEncoder(const Model& M, const std::vector<std::string>& vals) : M_(M), vals_(vals) {}
  std::string error;

  // rec: a REC of `var |-> value`.  out: wire record (h0 fixed words + bag words, sorted).
  bool encode(const TVal& st, std::vector<u64>* out) {
    if (st.kind != TVal::REC) return fail("a state must be a record of variables");
    const Model& M = M_;
    std::vector<u64> rec((size_t)M.h0, 0);
    for (int r = 1; r <= M.R; r++) {                      // Init values (VSR.tla:323-348) for what the text leaves out
      u64 A = a_set_view(a_set_status(0, ST_NORMAL), 1);
      for (int c = 1; c <= M.C; c++) A = a_set_ctrow(A, c, ct_make(0, 0, 1));
      rec[(size_t)aidx(r)] = A;
    }
    u64 hdr = 0;
    std::vector<u64> bag;
    // rep_view_number first: rep_svc_recv / rep_dvc_recv are checked against it
    static const char* const ORDER[] = {"rep_view_number", "rep_status", "rep_op_number", "rep_commit_number", "rep_last_normal_view",
                                        "rep_sent_dvc", "rep_sent_sv", "rep_peer_op_number", "rep_client_table", "rep_log",
                                        "rep_svc_recv", "rep_dvc_recv", "aux_svc", "aux_client_acked", "messages", "clients", "replicas",
                                        "aux_restart", "rep_rec_number", "rep_rec_recv"};
    for (const auto& f : st.fields) {
      bool known = f.first == "_TEAction";
      for (const char* o : ORDER) known = known || f.first == o;
      if (!known) return fail("unknown variable " + f.first);
    }
    for (const char* name : ORDER) {
      const TVal* v = st.field(name);
      if (!v) continue;
      const std::string n = name;
      if (n == "aux_svc") {
        long x;
        if (!integer(*v, 0, 7, n, &x)) return false;
        hdr |= (u64)x << 8;
      } else if (n == "aux_restart") {
        long x;
        if (!integer(*v, 0, 0, n, &x)) return false;       // RestartEmptyLimit = 0
      } else if (n == "clients" || n == "replicas") {
        const long want = n == "clients" ? M.C : M.R;
        // TLC keeps 1..N an interval (the reference's trace: `clients |-> 1..1`); the same value enumerated ({1, 2, 3}) is accepted too
        bool ok = v->kind == TVal::RANGE && v->i == 1 && v->j == want;
        if (v->kind == TVal::SET && (long)v->items.size() == want) {
          long seen = 0;
          for (const TVal& e : v->items)
            if (e.kind == TVal::INT && e.i >= 1 && e.i <= want) seen |= 1L << e.i;
          ok = seen == ((1L << (want + 1)) - 2);
        }
        if (!ok) return fail(n + " does not match the model's constants");
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
      } else {                                            // per-replica variables
        if (v->kind != TVal::SEQ || (int)v->items.size() != M.R) return fail(n + " must be a sequence of ReplicaCount elements");
        for (int r = 1; r <= M.R; r++) {
          const TVal& x = v->items[(size_t)(r - 1)];
          u64& A = rec[(size_t)aidx(r)];
          long k;
          int b;
          if (n == "rep_view_number") { if (!integer(x, 0, 7, n, &k)) return false; A = a_set_view(A, (int)k); }
          else if (n == "rep_op_number") { if (!integer(x, 0, 3, n, &k)) return false; A = a_set_op(A, (int)k); }
          else if (n == "rep_commit_number") { if (!integer(x, 0, 3, n, &k)) return false; A = a_set_commit(A, (int)k); }
          else if (n == "rep_last_normal_view") { if (!integer(x, 0, 7, n, &k)) return false; A = a_set_lnv(A, (int)k); }
          else if (n == "rep_sent_dvc") { if (!boolean(x, &b)) return false; A = a_set_sent_dvc(A, b); }
          else if (n == "rep_sent_sv") { if (!boolean(x, &b)) return false; A = a_set_sent_sv(A, b); }
          else if (n == "rep_rec_number") { if (!integer(x, 0, 0, n, &k)) return false; }
          else if (n == "rep_rec_recv") { if (x.kind != TVal::SET || !x.items.empty()) return fail("rep_rec_recv must be empty (recovery is dead code)"); }
          else if (n == "rep_status") {
            if (x.kind != TVal::ID) return fail("rep_status must be a model value");
            if (x.s == "Normal") A = a_set_status(A, ST_NORMAL);
            else if (x.s == "ViewChange") A = a_set_status(A, ST_VIEWCHANGE);
            else return fail("rep_status " + x.s + " cannot occur with RestartEmptyLimit = 0");
          } else if (n == "rep_peer_op_number") {
            if (x.kind != TVal::SEQ || (int)x.items.size() != M.R) return fail(n + ": one entry per replica expected");
            for (int p = 1; p <= M.R; p++) {
              if (!integer(x.items[(size_t)(p - 1)], 0, 3, n, &k)) return false;
              A = a_set_peer(A, p, (int)k);
            }
          } else if (n == "rep_client_table") {
            if (x.kind != TVal::SEQ || (int)x.items.size() != M.C) return fail(n + ": one row per client expected");
            for (int c = 1; c <= M.C; c++) {
              const TVal& row = x.items[(size_t)(c - 1)];
              long rq, op;
              int ex;
              if (row.kind != TVal::REC || !row.field("request_number") || !row.field("op_number") || !row.field("executed"))
                return fail("client table row: [request_number, op_number, executed] expected");
              if (!integer(*row.field("request_number"), 0, 3, "request_number", &rq) || !integer(*row.field("op_number"), 0, 3, "op_number", &op) ||
                  !boolean(*row.field("executed"), &ex))
                return false;
              A = a_set_ctrow(A, c, ct_make((int)rq, (int)op, ex));
            }
          } else if (n == "rep_log") {
            u32 lg;
            if (!log_value(x, 1, &lg)) return false;
            set_x(rec, r, 0, lg);
          } else if (n == "rep_svc_recv") {
            if (x.kind != TVal::SET) return fail(n + ": a set expected");
            int mask = 0;
            for (const TVal& e : x.items) {
              u64 w;
              if (!message(e, &w)) return false;
              if (m_type(w) != T_SVC || m_view(w) != a_view(A) || m_dest(w) != r)
                return fail("rep_svc_recv holds a record the mask form cannot express (type / view / dest)");
              mask |= 1 << (m_source(w) - 1);
            }
            A = a_set_svcmask(A, mask);
          } else if (n == "rep_dvc_recv") {
            if (x.kind != TVal::SET) return fail(n + ": a set expected");
            for (const TVal& e : x.items) {
              u64 w;
              if (!message(e, &w)) return false;
              if (m_type(w) != T_DVC || m_view(w) != a_view(A) || m_dest(w) != r)
                return fail("rep_dvc_recv holds a record the slot form cannot express (type / view / dest)");
              if (get_x(rec, r, m_source(w)) & 1) return fail("two DoViewChange records from one source");
              set_x(rec, r, m_source(w), dvc_make(m_lnv(w), m_op(w), m_commit(w), m_lg(w) & 0xFFFFFF));
            }
          }
        }
      }
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
  u32 get_x(const std::vector<u64>& rec, int r, int i) const { return (u32)(rec[(size_t)(aidx(r) + 1 + (i >> 1))] >> (32 * (i & 1))); }
  void set_x(std::vector<u64>& rec, int r, int i, u32 x) const {
    u64& w = rec[(size_t)(aidx(r) + 1 + (i >> 1))];
    const int sh = 32 * (i & 1);
    w = (w & ~((u64)0xFFFFFFFFu << sh)) | ((u64)x << sh);
  }
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
  bool entry(const TVal& v, int* out) {                   // log_entry, VSR.tla:375
    long vn, cl, rq;
    int vi;
    if (v.kind != TVal::REC || !v.field("view_number") || !v.field("operation") || !v.field("client_id") || !v.field("request_number"))
      return fail("log entry: [view_number, operation, client_id, request_number] expected");
    if (!integer(*v.field("view_number"), 1, 7, "entry view_number", &vn) || !value_index(*v.field("operation"), &vi) ||
        !integer(*v.field("client_id"), 1, 2, "client_id", &cl) || !integer(*v.field("request_number"), 0, 3, "request_number", &rq))
      return false;
    *out = entry_make((int)vn, vi, (int)cl, (int)rq);
    return true;
  }
  // a log: a sequence (entry i at op number first + i - 1) or a function op number :> entry
  bool log_value(const TVal& v, int first, u32* out) {
    u32 lg = 0;
    if (v.kind == TVal::SEQ) {
      if ((int)v.items.size() + first - 1 > 3) return fail("log longer than 3 entries");
      for (size_t k = 0; k < v.items.size(); k++) {
        int e;
        if (!entry(v.items[k], &e)) return false;
        lg |= (u32)e << (8 * ((int)k + first - 1));
      }
    } else if (v.kind == TVal::FCN) {
      for (const auto& kv : v.pairs) {
        long opn;
        int e;
        if (!integer(kv.first, 1, 3, "log index", &opn) || !entry(kv.second, &e)) return false;
        lg |= (u32)e << (8 * (opn - 1));
      }
    } else {
      return fail("a log must be a sequence or a function");
    }
    *out = lg;
    return true;
  }
  bool message(const TVal& v, u64* out) {                 // message records, VSR.tla:158-196, 537
    if (v.kind != TVal::REC || !v.field("type") || v.field("type")->kind != TVal::ID) return fail("message: a record with a type expected");
    const std::string& ty = v.field("type")->s;
    int t = ty == "StartViewChangeMsg" ? T_SVC : ty == "PrepareMsg" ? T_PREPARE : ty == "PrepareOkMsg" ? T_PREPAREOK
            : ty == "DoViewChangeMsg" ? T_DVC : ty == "StartViewMsg" ? T_SV : ty == "GetStateMsg" ? T_GETSTATE
            : ty == "NewStateMsg" ? T_NEWSTATE : 0;
    if (!t) return fail("message type " + ty + " is not produced by the live actions");
    auto num = [&](const char* name, long lo, long hi, long* out_) {
      const TVal* f = v.field(name);
      if (!f) return fail(std::string("message ") + ty + ": field " + name + " missing");
      return integer(*f, lo, hi, name, out_);
    };
    long view = 0, dest = 0, source = 0, op = 0, commit = 0, lnv = 0, fo = 0;
    u32 lg = 0;
    if (!num("view_number", 0, 7, &view) || !num("dest", 1, M_.R, &dest) || !num("source", 1, M_.R, &source)) return false;
    size_t want = 4;
    if (t != T_SVC) { if (!num("op_number", 0, 3, &op)) return false; want++; }
    if (t == T_PREPARE || t == T_DVC || t == T_SV || t == T_NEWSTATE) { if (!num("commit_number", 0, 3, &commit)) return false; want++; }
    if (t == T_PREPARE) {
      int e;
      if (!v.field("message") || !entry(*v.field("message"), &e)) return fail("PrepareMsg: field message missing or malformed");
      lg = (u32)e;
      want++;
    }
    if (t == T_DVC) { if (!num("last_normal_vn", 0, 7, &lnv)) return false; want++; }
    if (t == T_NEWSTATE) { if (!num("first_op", 1, 3, &fo)) return false; want++; }
    if (t == T_DVC || t == T_SV || t == T_NEWSTATE) {
      if (!v.field("log") || !log_value(*v.field("log"), t == T_NEWSTATE ? (int)fo : 1, &lg)) return fail("message " + ty + ": field log missing or malformed");
      want++;
    }
    if (v.fields.size() != want) return fail("message " + ty + ": unexpected fields");
    *out = m_make(t, (int)view, (int)dest, (int)source, (int)op, (int)commit, (int)lnv, (int)fo, lg);
    return true;
  }

  const Model& M_;
  const std::vector<std::string>& vals_;
};

struct ParsedState {
  std::vector<u64> rec;      // wire record, bag sorted
  std::string action;        // "" when the text does not say
};

// text -> states.  Returns false and sets *err on malformed input.
inline bool parse_states_tlc(const Model& M, const std::vector<std::string>& vals, const std::string& text,
                             std::vector<ParsedState>* out, std::string* err) {
  out->clear();
  auto encode_one = [&](const TVal& st, const std::string& action) {
    ParsedState ps;
    if (M.model_id != 0) {                                       // VR_STATE_TRANSFER / VR_APP_STATE: their own record layouts
      std::string e2;
      if (!encode_state_analysis_models(M, vals, st, &ps.rec, &e2)) {
        *err = "state " + std::to_string(out->size() + 1) + ": " + e2;
        return false;
      }
    } else {
      StateEncoder enc(M, vals);
      if (!enc.encode(st, &ps.rec)) {
        *err = "state " + std::to_string(out->size() + 1) + ": " + enc.error;
        return false;
      }
    }
    ps.action = action;
    if (const TVal* te = st.field("_TEAction"))
      if (te->kind == TVal::REC && te->field("name") && te->field("name")->kind == TVal::STR) ps.action = te->field("name")->s;
    out->push_back(std::move(ps));
    return true;
  };
  size_t first = text.find_first_not_of(" \t\r\n");
  if (first == std::string::npos) { *err = "empty text"; return false; }
  if (text.compare(first, 2, "<<") == 0 || text[first] == '[') {           // (a) trace expression, (b) one state record
    TlcParser p(text);
    TVal v;
    if (!p.value(&v) || !p.eof()) {
      if (p.error.empty()) p.fail("trailing text");
      *err = p.error;
      return false;
    }
    if (v.kind == TVal::REC) return encode_one(v, "");
    for (const TVal& st : v.items)
      if (!encode_one(st, "")) return false;
    return true;
  }
  // (c) console form: blocks that start with "State k: <Action ...>" (or a bare conjunct list for one state)
  size_t pos = first;
  while (pos < text.size()) {
    std::string action;
    if (text.compare(pos, 6, "State ") == 0) {
      size_t eol = text.find('\n', pos);
      if (eol == std::string::npos) eol = text.size();
      size_t lt = text.find('<', pos);
      if (lt != std::string::npos && lt < eol) {
        size_t e = lt + 1;
        while (e < eol && text[e] != ' ' && text[e] != '>') e++;
        action = text.substr(lt + 1, e - lt - 1);
        if (action == "Initial") action = "Initial predicate";
      }
      pos = eol;
    }
    size_t next = text.find("\nState ", pos);
    const std::string block = text.substr(pos, next == std::string::npos ? std::string::npos : next - pos);
    TlcParser p(block);
    TVal st;
    if (!p.conjuncts(&st) || !p.eof()) {
      if (p.error.empty()) p.fail("trailing text");
      *err = "state " + std::to_string(out->size() + 1) + ": " + p.error;
      return false;
    }
    if (!encode_one(st, action)) return false;
    if (next == std::string::npos) break;
    pos = next + 1;
  }
  return true;
}

}  // namespace vsr

// vsr_format.hpp — host-side printer: wire-layout record -> TLC's value syntax, one `var |-> value` line per variable in
// TLC's (alphabetical) trace-expression order, values in TLC's normal form.  Mirrors what TLC prints for a state in
// /root/reference/state_transfer_violation_trace.txt (format evidence: trace:8-24, 563-577; ordering rules SURVEY
// App. B4).  Pinned by tests against the per-line SHA-256 digests of that file (tests/golden/state_transfer_trace.json).
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "vsr_model.hpp"

namespace vsr {

inline std::string fmt_entry(const std::vector<std::string>& vals, int e) {
  return "[view_number |-> " + std::to_string(e & 7) + ", operation |-> " + vals[entry_val(e)] + ", client_id |-> " +
         std::to_string(entry_client(e)) + ", request_number |-> " + std::to_string(entry_req(e)) + "]";
}
inline std::string fmt_seq_log(const std::vector<std::string>& vals, u32 lg) {
  std::string s = "<<";
  bool first = true;
  for (int i = 1; i <= 3; i++) {
    int e = log_byte(lg, i);
    if (!e) continue;
    if (!first) s += ", ";
    s += fmt_entry(vals, e);
    first = false;
  }
  return s + ">>";
}
inline const char* fmt_type(int t) {
  switch (t) {
    case T_SVC: return "StartViewChangeMsg";
    case T_PREPARE: return "PrepareMsg";
    case T_PREPAREOK: return "PrepareOkMsg";
    case T_DVC: return "DoViewChangeMsg";
    case T_SV: return "StartViewMsg";
    case T_GETSTATE: return "GetStateMsg";
    case T_NEWSTATE: return "NewStateMsg";
  }
  return "?";
}

// A message in normal form: field names ordered by TLC's interning order (first occurrence in VSR.tla:158-196, 537).
inline std::string fmt_msg(const std::vector<std::string>& vals, u64 w) {
  int t = m_type(w);
  std::string s = "[view_number |-> " + std::to_string(m_view(w)) + ", type |-> " + fmt_type(t);
  auto od = [&](bool op, bool commit) {
    if (op) s += ", op_number |-> " + std::to_string(m_op(w));
    if (commit) s += ", commit_number |-> " + std::to_string(m_commit(w));
    s += ", dest |-> " + std::to_string(m_dest(w)) + ", source |-> " + std::to_string(m_source(w));
  };
  switch (t) {
    case T_SVC: od(false, false); break;
    case T_PREPAREOK:
    case T_GETSTATE: od(true, false); break;
    case T_PREPARE:
      s += ", message |-> " + fmt_entry(vals, (int)(m_lg(w) & 0xFF));
      od(true, true);
      break;
    case T_SV:
      od(true, true);
      s += ", log |-> " + fmt_seq_log(vals, m_lg(w) & 0xFFFFFF);
      break;
    case T_DVC:
      od(true, true);
      s += ", log |-> " + fmt_seq_log(vals, m_lg(w) & 0xFFFFFF) + ", last_normal_vn |-> " + std::to_string(m_lnv(w));
      break;
    case T_NEWSTATE: {
      od(true, true);
      int fo = m_first_op(w);
      u32 lg = m_lg(w) & 0xFFFFFF;
      if (fo == 1) {
        s += ", log |-> " + fmt_seq_log(vals, lg);
      } else if (!(lg >> (8 * (fo - 1)))) {  // first_op > op_number: the empty function
        s += ", log |-> <<>>";
      } else {                             // a function on first_op..op_number that is not a sequence
        s += ", log |-> (";
        bool first = true;
        for (int i = fo; i <= 3; i++) {
          int e = log_byte(lg, i);
          if (!e) continue;
          if (!first) s += " @@ ";
          s += std::to_string(i) + " :> " + fmt_entry(vals, e);
          first = false;
        }
        s += ")";
      }
      s += ", first_op |-> " + std::to_string(fo);
      break;
    }
  }
  return s + "]";
}

// TLC's order between two message records (RecordValue.compareTo, [TLC-RECALLED]): fewer fields first, then field by field the name and the value —
// every type starts [view_number, type, ...], so: arity, view_number, type (model values: cfg order), then the remaining fields of that one type.
inline std::vector<int> msg_sort_key(u64 w) {
  int t = m_type(w);
  std::vector<int> k;
  static const int TYPE_ORDER[8] = {0, 6, 3, 4, 7, 8, 9, 10};   // declaration order of the type constants VSR.tla:104-115
  switch (t) {
    case T_SVC: k = {4, m_view(w), TYPE_ORDER[t], m_dest(w), m_source(w)}; break;
    case T_PREPAREOK:
    case T_GETSTATE: k = {5, m_view(w), TYPE_ORDER[t], m_op(w), m_dest(w), m_source(w)}; break;
    case T_PREPARE: {
      int e = (int)(m_lg(w) & 0xFF);
      k = {7, m_view(w), TYPE_ORDER[t], e & 7, entry_val(e), entry_client(e), entry_req(e), m_op(w), m_commit(w), m_dest(w), m_source(w)};
      break;
    }
    case T_SV:
    case T_DVC:
    case T_NEWSTATE: {
      int arity = t == T_SV ? 7 : 8;
      k = {arity, m_view(w), TYPE_ORDER[t], m_op(w), m_commit(w), m_dest(w), m_source(w)};
      u32 lg = m_lg(w) & 0xFFFFFF;
      k.push_back(log_len(lg));            // sequences: shorter first, then element-wise; NewState.log (a function over first_op..op_number): then the lower bound
      if (t == T_NEWSTATE) k.push_back(m_first_op(w));
      for (int i = 1; i <= 3; i++) {
        int e = log_byte(lg, i);
        if (e) { k.push_back(e & 7); k.push_back(entry_val(e)); k.push_back(entry_client(e)); k.push_back(entry_req(e)); }
      }
      k.push_back(t == T_DVC ? m_lnv(w) : m_first_op(w));
      break;
    }
    default: k = {99};
  }
  return k;
}

template <typename F>
inline std::string fmt_per_replica(const Model& M, F f) {
  std::string s = "<<";
  for (int r = 1; r <= M.R; r++) s += (r > 1 ? ", " : "") + f(r);
  return s + ">>";
}

// `rec` is a wire-layout record.  Returns "[\nvar |-> value,\n...\n]".
inline std::string format_state_tlc(const Model& M, const std::vector<std::string>& vals, const u64* rec) {
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  auto A = [&](int r) { return rec[1 + (r - 1) * M.wpr]; };
  auto X = [&](int r, int i) { return (u32)(rec[1 + (r - 1) * M.wpr + 1 + (i >> 1)] >> (32 * (i & 1))); };
  auto B = [](int b) { return std::string(b ? "TRUE" : "FALSE"); };
  std::vector<std::string> lines;
  {  // aux_client_acked (VSR.tla:138): partial function Values -> BOOLEAN
    std::string s;
    int cnt = 0;
    for (int v = 0; v < M.n; v++)
      if (hdr_acked(hdr, v)) {
        s += (cnt ? " @@ " : "") + vals[v] + " :> " + B(hdr_acked(hdr, v) == 2);
        cnt++;
      }
    lines.push_back("aux_client_acked |-> " + (cnt ? "(" + s + ")" : std::string("<<>>")));
  }
  lines.push_back("aux_restart |-> 0");
  lines.push_back("aux_svc |-> " + std::to_string(hdr_aux_svc(hdr)));
  lines.push_back("clients |-> 1.." + std::to_string(M.C));
  {  // messages (VSR.tla:135): bag, zero-count keys stay
    std::vector<u64> ms(rec + M.h0, rec + M.h0 + nmsg);
    std::sort(ms.begin(), ms.end(), [](u64 a, u64 b) { return msg_sort_key(a) < msg_sort_key(b); });
    std::string s;
    for (int j = 0; j < nmsg; j++) s += (j ? " @@ " : "") + fmt_msg(vals, ms[j]) + " :> " + std::to_string(m_count(ms[j]));
    lines.push_back("messages |-> " + (nmsg ? "(" + s + ")" : std::string("<<>>")));
  }
  lines.push_back("rep_client_table |-> " + fmt_per_replica(M, [&](int r) {
    std::string s = "<<";
    for (int c = 1; c <= M.C; c++) {
      int row = a_ctrow(A(r), c);
      s += std::string(c > 1 ? ", " : "") + "[request_number |-> " + std::to_string(ct_req(row)) + ", op_number |-> " +
           std::to_string(ct_op(row)) + ", executed |-> " + B(ct_exec(row)) + "]";
    }
    return s + ">>";
  }));
  lines.push_back("rep_commit_number |-> " + fmt_per_replica(M, [&](int r) { return std::to_string(a_commit(A(r))); }));
  lines.push_back("rep_dvc_recv |-> " + fmt_per_replica(M, [&](int r) {
    std::vector<u64> set;
    for (int s = 1; s <= M.R; s++) {
      u32 x = X(r, s);
      if (x & 1) set.push_back(m_make(T_DVC, a_view(A(r)), r, s, dvc_op(x), dvc_commit(x), dvc_lnv(x), 0, dvc_log(x)));
    }
    std::sort(set.begin(), set.end(), [](u64 a, u64 b) { return msg_sort_key(a) < msg_sort_key(b); });
    std::string out = "{";
    for (size_t i = 0; i < set.size(); i++) out += (i ? ", " : "") + fmt_msg(vals, set[i]);
    return out + "}";
  }));
  lines.push_back("rep_last_normal_view |-> " + fmt_per_replica(M, [&](int r) { return std::to_string(a_lnv(A(r))); }));
  lines.push_back("rep_log |-> " + fmt_per_replica(M, [&](int r) { return fmt_seq_log(vals, X(r, 0)); }));
  lines.push_back("rep_op_number |-> " + fmt_per_replica(M, [&](int r) { return std::to_string(a_op(A(r))); }));
  lines.push_back("rep_peer_op_number |-> " + fmt_per_replica(M, [&](int r) {
    std::string s = "<<";
    for (int p = 1; p <= M.R; p++) s += (p > 1 ? ", " : "") + std::to_string(a_peer(A(r), p));
    return s + ">>";
  }));
  lines.push_back("rep_rec_number |-> " + fmt_per_replica(M, [&](int) { return std::string("0"); }));
  lines.push_back("rep_rec_recv |-> " + fmt_per_replica(M, [&](int) { return std::string("{}"); }));
  lines.push_back("rep_sent_dvc |-> " + fmt_per_replica(M, [&](int r) { return B(a_sent_dvc(A(r))); }));
  lines.push_back("rep_sent_sv |-> " + fmt_per_replica(M, [&](int r) { return B(a_sent_sv(A(r))); }));
  lines.push_back("rep_status |-> " + fmt_per_replica(M, [&](int r) {
    int st = a_status(A(r));
    return std::string(st == ST_NORMAL ? "Normal" : st == ST_VIEWCHANGE ? "ViewChange" : "Recovering");
  }));
  lines.push_back("rep_svc_recv |-> " + fmt_per_replica(M, [&](int r) {
    std::string out = "{";
    bool first = true;
    for (int s = 1; s <= M.R; s++)
      if ((a_svcmask(A(r)) >> (s - 1)) & 1) {
        out += (first ? "" : ", ") + fmt_msg(vals, m_make(T_SVC, a_view(A(r)), r, s, 0, 0, 0, 0, 0));
        first = false;
      }
    return out + "}";
  }));
  lines.push_back("rep_view_number |-> " + fmt_per_replica(M, [&](int r) { return std::to_string(a_view(A(r))); }));
  lines.push_back("replicas |-> 1.." + std::to_string(M.R));
  std::string out = "[\n";
  for (size_t i = 0; i < lines.size(); i++) out += lines[i] + (i + 1 < lines.size() ? ",\n" : "\n");
  return out + "]";
}

}  // namespace vsr

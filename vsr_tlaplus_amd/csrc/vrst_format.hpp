// vrst_format.hpp — host-side printer of the SECOND model's states (VR_STATE_TRANSFER.tla:69-87): wire-layout record -> TLC's
// value syntax, one `var |-> value` line per variable in alphabetical order (the form of a TLC trace expression).  Field names of
// the message records in their interning order (first occurrence, VRST.tla:104-163), as vsr_format.hpp does for VSR.tla; the
// reference ships no printed trace of this model, so the exact TLC normal form is [TLC-RECALLED] here.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "vrst_actions.hpp"

namespace vsr {
namespace vrst {

inline std::string fmt_entry2(const std::vector<std::string>& vals, int v) { return "[operation |-> " + vals[v] + "]"; }
inline std::string fmt_bytes_log(const std::vector<std::string>& vals, u32 bytes, int from) {
  std::string s = "<<";
  bool first = true;
  for (int i = from; i <= 3; i++) {
    const int e = (int)((bytes >> (8 * (i - 1))) & 0xFF);
    if (!(e & 7)) continue;
    if (!first) s += ", ";
    s += fmt_entry2(vals, (e >> 3) & 3);
    first = false;
  }
  return s + ">>";
}
inline const char* fmt_type2(int t) {
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
inline std::string fmt_dest(int d) { return d == ANYDEST ? std::string("AnyDest") : std::to_string(d); }
inline std::string fmt_msg2(const std::vector<std::string>& vals, u64 w) {
  const int t = m_type(w);
  std::string s = "[type |-> " + std::string(fmt_type2(t)) + ", view_number |-> " + std::to_string(m_view(w));
  const u32 lg = m_lg(w) & 0xFFFFFF;
  if (t == T_PREPARE) s += ", message |-> " + fmt_entry2(vals, (int)((lg >> 3) & 3));
  if (t == T_DVC || t == T_SV) s += ", log |-> " + fmt_bytes_log(vals, lg, 1);
  if (t == T_NEWSTATE) {
    const int fo = m_first_op(w);
    if (fo == 1 || !(lg >> (8 * (fo - 1)))) {
      s += ", log |-> " + (fo == 1 ? fmt_bytes_log(vals, lg, 1) : std::string("<<>>"));
    } else {
      s += ", log |-> (";
      bool first = true;
      for (int i = fo; i <= 3; i++) {
        const int e = (int)((lg >> (8 * (i - 1))) & 0xFF);
        if (!(e & 7)) continue;
        s += std::string(first ? "" : " @@ ") + std::to_string(i) + " :> " + fmt_entry2(vals, (e >> 3) & 3);
        first = false;
      }
      s += ")";
    }
    s += ", first_op |-> " + std::to_string(fo);
  }
  if (t == T_DVC) s += ", last_normal_vn |-> " + std::to_string(m_lnv(w));
  if (t != T_SVC) s += ", op_number |-> " + std::to_string(m_op(w));
  if (t == T_PREPARE || t == T_DVC || t == T_SV || t == T_NEWSTATE) s += ", commit_number |-> " + std::to_string(m_commit(w));
  return s + ", dest |-> " + fmt_dest(m_dest(w)) + ", source |-> " + std::to_string(m_source(w)) + "]";
}

template <typename F>
inline std::string per_replica(const Model& M, F f) {
  std::string s = "<<";
  for (int r = 1; r <= M.R; r++) s += (r > 1 ? ", " : "") + f(r);
  return s + ">>";
}

inline std::string format_state_tlc(const Model& M, const std::vector<std::string>& vals, const u64* rec) {
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  auto A = [&](int r) { return rec[r]; };
  auto B = [](int b) { return std::string(b ? "TRUE" : "FALSE"); };
  std::vector<std::string> lines;
  {
    std::string s;
    int cnt = 0;
    for (int v = 0; v < M.n; v++)
      if (hdr_acked(hdr, v)) {
        s += (cnt ? " @@ " : "") + vals[v] + " :> " + B(hdr_acked(hdr, v) == 2);
        cnt++;
      }
    lines.push_back("aux_client_acked |-> " + (cnt ? "(" + s + ")" : std::string("<<>>")));
  }
  lines.push_back("aux_svc |-> " + std::to_string(hdr_aux_svc(hdr)));
  {
    std::vector<u64> ms(rec + M.h0, rec + M.h0 + nmsg);
    std::sort(ms.begin(), ms.end());
    std::string s;
    for (int j = 0; j < nmsg; j++) s += (j ? " @@ " : "") + fmt_msg2(vals, ms[j]) + " :> " + std::to_string(m_count(ms[j]));
    lines.push_back("messages |-> " + (nmsg ? "(" + s + ")" : std::string("<<>>")));
  }
  lines.push_back("no_progress |-> " + per_replica(M, [&](int r) { return B(b_noprog(A(r))); }));
  lines.push_back("no_progress_ctr |-> " + std::to_string((int)((hdr >> 20) & 7)));
  lines.push_back("rep_commit_number |-> " + per_replica(M, [&](int r) { return std::to_string(a_commit(A(r))); }));
  lines.push_back("rep_last_normal_view |-> " + per_replica(M, [&](int r) { return std::to_string(a_lnv(A(r))); }));
  lines.push_back("rep_log |-> " + per_replica(M, [&](int r) { return fmt_bytes_log(vals, blog_to_bytes(b_log(A(r))), 1); }));
  lines.push_back("rep_op_number |-> " + per_replica(M, [&](int r) { return std::to_string(a_op(A(r))); }));
  lines.push_back("rep_peer_op_number |-> " + per_replica(M, [&](int r) {
    std::string s = "<<";
    for (int p = 1; p <= M.R; p++) s += (p > 1 ? ", " : "") + std::to_string(b_peer(A(r), p));
    return s + ">>";
  }));
  lines.push_back("rep_sent_dvc |-> " + per_replica(M, [&](int r) { return B(a_sent_dvc(A(r))); }));
  lines.push_back("rep_sent_sv |-> " + per_replica(M, [&](int r) { return B(a_sent_sv(A(r))); }));
  lines.push_back("rep_status |-> " + per_replica(M, [&](int r) {
    const int st = a_status(A(r));
    return std::string(st == ST2_NORMAL ? "Normal" : st == ST2_VIEWCHANGE ? "ViewChange" : "StateTransfer");
  }));
  lines.push_back("rep_view_number |-> " + per_replica(M, [&](int r) { return std::to_string(a_view(A(r))); }));
  lines.push_back("replicas |-> 1.." + std::to_string(M.R));
  std::string out = "[\n";
  for (size_t i = 0; i < lines.size(); i++) out += lines[i] + (i + 1 < lines.size() ? ",\n" : "\n");
  return out + "]";
}

}  // namespace vrst
}  // namespace vsr

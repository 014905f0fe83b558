// vras_format.hpp — host-side printer of the THIRD model's states (VR_APP_STATE.tla:68-91): wire-layout record -> TLC's value
// syntax, one `var |-> value` line per variable in alphabetical order (the form of a TLC trace expression), as vrst_format.hpp
// does for the second model; the reference ships no printed trace of this model, so the exact TLC normal form is [TLC-RECALLED].
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "vras_actions.hpp"
#include "vrst_format.hpp"

namespace vsr {
namespace vras {

inline std::string format_state_tlc(const Model& M, const std::vector<std::string>& vals, const u64* rec) {
  using vrst::fmt_bytes_log;
  using vrst::fmt_entry2;
  using vrst::fmt_msg2;
  using vrst::per_replica;
  const u64 hdr = rec[0];
  const int nmsg = hdr_nmsg(hdr);
  auto A = [&](int r) { return rec[c_ia(r)]; };
  auto Bw = [&](int r) { return rec[c_ia(r) + 1]; };
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
  lines.push_back("aux_restart |-> 0");                          // VRAS.tla:91, never written
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
  lines.push_back("rep_app_state |-> " + per_replica(M, [&](int r) {
    std::string s = "<<";
    for (int i = 1; i <= a_commit(A(r)); i++) s += (i > 1 ? ", " : "") + fmt_entry2(vals, c_app(A(r), i));
    return s + ">>";
  }));
  lines.push_back("rep_commit_number |-> " + per_replica(M, [&](int r) { return std::to_string(a_commit(A(r))); }));
  lines.push_back("rep_last_normal_view |-> " + per_replica(M, [&](int r) { return std::to_string(a_lnv(A(r))); }));
  lines.push_back("rep_log |-> " + per_replica(M, [&](int r) { return fmt_bytes_log(vals, blog_to_bytes(b_log(A(r))), 1); }));
  lines.push_back("rep_op_number |-> " + per_replica(M, [&](int r) { return std::to_string(a_op(A(r))); }));
  lines.push_back("rep_peer_op_number |-> " + per_replica(M, [&](int r) {
    std::string s = "<<";
    for (int p = 1; p <= M.R; p++) s += (p > 1 ? ", " : "") + std::to_string(b_peer(A(r), p));
    return s + ">>";
  }));
  lines.push_back("rep_rec_number |-> " + per_replica(M, [&](int) { return std::string("0"); }));   // :83, never written
  lines.push_back("rep_rec_recv |-> " + per_replica(M, [&](int) { return std::string("{}"); }));    // :84, never written
  lines.push_back("rep_recv_dvc |-> " + per_replica(M, [&](int r) {
    std::string s = "{";
    bool first = true;
    for (int src = 1; src <= M.R; src++) {
      const u32 sl = d_slot(Bw(r), src);
      if (!(sl & 1)) continue;
      const u64 w = m_make(T_DVC, d_view(Bw(r)), r, src, (int)((sl >> 4) & 3), (int)((sl >> 6) & 3), (int)((sl >> 1) & 7), 0,
                           blog_to_bytes((sl >> 8) & 0x1FF));
      s += std::string(first ? "" : ", ") + fmt_msg2(vals, w);
      first = false;
    }
    return s + "}";
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

}  // namespace vras
}  // namespace vsr

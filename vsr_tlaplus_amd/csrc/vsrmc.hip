// vsrmc.hip — host side of libvsrmc.so: cfg reader, model lowering, FPSet / expand / checker handles (include/vsrmc.h).
// Mirrors tlc2.TLC (config reading), tlc2.tool.ModelChecker + Worker.run (level loop), StateQueue (frontier double
// buffer), FPSet (seen-set) and TLCTrace (parent/ordinal log) for VSR.tla — see SURVEY.md §3.1 for the TLC loop.
// All model semantics run on the GPU (vsr_kernels.hpp); there is no CPU fallback: without a HIP device every
// compute entry point fails with VSRMC_E_HIP.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/vsrmc.h"
#include "vsr_format.hpp"
#include "vrst_format.hpp"
#include "vras_format.hpp"
#include "vsr_parse.hpp"
#include "vras_parse.hpp"
#include "vsr_kernels.hpp"

// threads per block of the ordinary-level kernel (k_expand<.., PLAIN, BLK>): 64 = one wave per block with its own 16-record tiles
#define VSRMC_FP_VERSION 2          // fingerprint function of this build; checkpoints of another version are refused
#ifndef VSRMC_DEFAULT_BLK
#define VSRMC_DEFAULT_BLK 256
#endif

using namespace vsr;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t e_ = (expr);                                                                            \
    if (e_ != hipSuccess) return fail(VSRMC_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// ---------------------------------------------------------------------------------------------------------------
// SHA-256 (FIPS 180-4), used to refuse any module other than the VSR.tla this build lowers.
// ---------------------------------------------------------------------------------------------------------------
std::string sha256_hex(const std::string& data) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  std::string msg = data;
  uint64_t bitlen = (uint64_t)data.size() * 8;
  msg.push_back((char)0x80);
  while (msg.size() % 64 != 56) msg.push_back((char)0);
  for (int i = 7; i >= 0; i--) msg.push_back((char)((bitlen >> (8 * i)) & 0xFF));
  auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
  for (size_t blk = 0; blk < msg.size(); blk += 64) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
      w[i] = ((uint32_t)(uint8_t)msg[blk + 4 * i] << 24) | ((uint32_t)(uint8_t)msg[blk + 4 * i + 1] << 16) |
             ((uint32_t)(uint8_t)msg[blk + 4 * i + 2] << 8) | (uint32_t)(uint8_t)msg[blk + 4 * i + 3];
    for (int i = 16; i < 64; i++) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
      uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
      uint32_t ch = (e & f) ^ (~e & g);
      uint32_t t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
      uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
      uint32_t t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
  }
  char buf[65];
  for (int i = 0; i < 8; i++) std::snprintf(buf + 8 * i, 9, "%08x", h[i]);
  return std::string(buf, 64);
}

// SHA-256 of the one module this build lowers: /root/reference/vsr-revisited/paper/VSR.tla (970 lines)
const char* const VSR_TLA_SHA256 = "f37efb7b055316624c885e2805550097fa609864b1b7a782279c189f8e2dbf22";
// ... and of the second one: /root/reference/vsr-revisited/paper/analysis/03-state-transfer/VR_STATE_TRANSFER.tla (948 lines)
const char* const VRST_TLA_SHA256 = "e716e3e04a9f9284d9e2df039271a37d4974c1d32d19dbe54a8da5de46331eee";
// /root/reference/vsr-revisited/paper/analysis/04-application-state/VR_APP_STATE.tla (the third model, vras_actions.hpp)
const char* const VRAS_TLA_SHA256 = "6ef22989c86c9d5bb9c9c9829a09e9ee0e035c75e7b894c0bc6dbeba1f476bb6";

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------------------------------------------
struct vsrmc_model {
  Model M;
  int symmetry = 1;
  int check_deadlock = 0;
  std::vector<std::string> value_names;
};

namespace {

int build_model(int R, int C, int n, int L, int restart, int symmetry, int inv_mask, int assume_commit, vsrmc_model* out) {
  if (R < 2 || R > 5 || C < 1 || C > 2 || n < 1 || n > 3 || L < 0 || L > 6)
    return fail(VSRMC_E_CFG, "model constants outside the supported bounds (ReplicaCount 2..5, ClientCount 1..2, "
                             "|Values| 1..3, StartViewOnTimerLimit 0..6)");
  if (restart != 0)
    return fail(VSRMC_E_CFG, "RestartEmptyLimit > 0 is not supported: the recovery actions (VSR.tla:813-894) are not lowered");
  Model& M = out->M;
  std::memset(&M, 0, sizeof(M));
  M.R = R; M.C = C; M.n = n; M.L = L;
  M.wpr = 1 + (R + 2) / 2;
  M.h0 = 1 + R * M.wpr;
  int perms[6][3] = {{0, 1, 2}, {0, 2, 1}, {1, 0, 2}, {1, 2, 0}, {2, 0, 1}, {2, 1, 0}};
  int np = 0;
  for (int i = 0; i < 6; i++) {
    bool ok = true;                       // a permutation of {0..n-1}: fixes every index >= n
    for (int v = n; v < 3; v++) ok = ok && perms[i][v] == v;
    if (!ok) continue;
    if (!symmetry && np >= 1) break;
    M.pitab[np++] = (u32)perms[i][0] | ((u32)perms[i][1] << 2) | ((u32)perms[i][2] << 4);
  }
  M.np = np;
  M.fixed = M.h0 + M.np;
  M.assume_commit = assume_commit ? 1 : 0;
  M.inv_mask = inv_mask;
  // LDS stride of one staged record: 63 words (R <= 3) or 95 words (R >= 4: more replicas, larger bags); odd, so that the
  // columns the slot-major enumeration reads are bank-conflict free.  max_bag = stride - fixed.
  M.max_bag = (R <= 3 ? 63 : 95) - M.fixed;
  M.m0 = 4 * R + R * C * n;
  M.primtab = 0;
  for (int v = 1; v <= 7; v++) M.primtab |= (u32)(1 + ((v - 1) % R)) << (3 * v);   // Primary(v) == 1 + ((v - 1) % ReplicaCount), VSR.tla:287-288
  out->symmetry = symmetry ? 1 : 0;
  out->value_names.clear();
  for (int v = 0; v < n; v++) out->value_names.push_back("v" + std::to_string(v + 1));
  return 0;
}

// The second model (VR_STATE_TRANSFER.tla): one word per replica, no clients, no symmetry (vrst_actions.hpp)
int build_model2(int R, int n, int L, int no_progress_limit, int symmetry, int inv_mask, vsrmc_model* out) {
  if (R < 2 || R > 5 || n < 1 || n > 3 || L < 0 || L > 6)
    return fail(VSRMC_E_CFG, "model constants outside the supported bounds (ReplicaCount 2..5, |Values| 1..3, StartViewOnTimerLimit 0..6)");
  if (no_progress_limit != 0)
    return fail(VSRMC_E_CFG, "NoProgressChangeLimit > 0 is not supported: NoProgressChange (VR_STATE_TRANSFER.tla:765-776) is not lowered");
  if (symmetry)
    return fail(VSRMC_E_CFG, "SYMMETRY is not lowered for VR_STATE_TRANSFER.tla (VR_STATE_TRANSFER.cfg:25-27 keeps it commented out)");
  Model& M = out->M;
  std::memset(&M, 0, sizeof(M));
  M.model_id = 1;
  M.R = R; M.C = 0; M.n = n; M.L = L;
  M.wpr = 1;
  M.h0 = 1 + R;
  M.np = 1;
  M.pitab[0] = 0x24u;                                            // the identity
  M.fixed = M.h0 + 1;
  M.inv_mask = inv_mask;
  M.max_bag = 63 - M.fixed;
  M.m0 = 4 * R + R * n;
  for (int v = 1; v <= 7; v++) M.primtab |= (u32)(1 + ((v - 1) % R)) << (3 * v);   // Primary(v), VR_STATE_TRANSFER.tla:233-234
  out->symmetry = 0;
  out->value_names.clear();
  for (int v = 0; v < n; v++) out->value_names.push_back("v" + std::to_string(v + 1));
  return 0;
}

// The third model (VR_APP_STATE.tla): two words per replica (state + received DoViewChange set), no clients, no symmetry
// (vras_actions.hpp); ReplicaCount <= 3: the received-DoViewChange word holds three 17-bit slots
int build_model3(int R, int n, int L, int no_progress_limit, int symmetry, int inv_mask, vsrmc_model* out) {
  if (R < 2 || R > 3 || n < 1 || n > 3 || L < 0 || L > 6)
    return fail(VSRMC_E_CFG, "model constants outside the supported bounds (ReplicaCount 2..3, |Values| 1..3, StartViewOnTimerLimit 0..6)");
  if (no_progress_limit != 0)
    return fail(VSRMC_E_CFG, "NoProgressChangeLimit > 0 is not supported: NoProgressChange (VR_APP_STATE.tla:797-807) is not lowered");
  if (symmetry)
    return fail(VSRMC_E_CFG, "SYMMETRY is not lowered for VR_APP_STATE.tla (VR_APP_STATE.cfg:26-28 keeps it commented out)");
  Model& M = out->M;
  std::memset(&M, 0, sizeof(M));
  M.model_id = 2;
  M.R = R; M.C = 0; M.n = n; M.L = L;
  M.wpr = 2;
  M.h0 = 1 + 2 * R;
  M.np = 1;
  M.pitab[0] = 0x24u;                                            // the identity
  M.fixed = M.h0 + 1;
  M.inv_mask = inv_mask;
  M.max_bag = 63 - M.fixed;
  M.m0 = 4 * R + R * n;
  for (int v = 1; v <= 7; v++) M.primtab |= (u32)(1 + ((v - 1) % R)) << (3 * v);   // Primary(v), VR_APP_STATE.tla:238-239
  out->symmetry = 0;
  out->value_names.clear();
  for (int v = 0; v < n; v++) out->value_names.push_back(std::string(1, (char)('a' + v)));   // VR_APP_STATE.cfg:5 Values = {a, b}
  return 0;
}

std::string strip(const std::string& s) {
  size_t a = s.find_first_not_of(" \t\r\n"), b = s.find_last_not_of(" \t\r\n");
  return a == std::string::npos ? "" : s.substr(a, b - a + 1);
}

// wire layout -> device layout (insert np zero H words); returns device length
int wire_to_device(const Model& M, const u64* wire, u64* dev) {
  int nmsg = hdr_nmsg(wire[0]);
  for (int k = 0; k < M.h0; k++) dev[k] = wire[k];
  for (int i = 0; i < M.np; i++) dev[M.h0 + i] = 0;
  for (int j = 0; j < nmsg; j++) dev[M.fixed + j] = wire[M.h0 + j];
  return M.fixed + nmsg;
}
int device_to_wire(const Model& M, const u64* dev, u64* wire) {
  int nmsg = hdr_nmsg(dev[0]);
  for (int k = 0; k < M.h0; k++) wire[k] = dev[k];
  for (int j = 0; j < nmsg; j++) wire[M.h0 + j] = dev[M.fixed + j];
  return M.h0 + nmsg;
}

void init_record_wire(const Model& M, std::vector<u64>& rec) {   // Init, VSR.tla:323-348
  rec.assign(M.h0, 0);
  if (M.model_id == 1) {                                         // Init, VR_STATE_TRANSFER.tla:267-283
    for (int r = 1; r <= M.R; r++) rec[r] = a_set_lnv(a_set_view(a_set_status(0, vrst::ST2_NORMAL), 1), 1);   // view 1, last normal view 1
    return;
  }
  if (M.model_id == 2) {                                         // Init, VR_APP_STATE.tla:292-315 (rep_app_state, rep_recv_dvc empty)
    for (int r = 1; r <= M.R; r++) rec[vras::c_ia(r)] = a_set_lnv(a_set_view(a_set_status(0, vrst::ST2_NORMAL), 1), 1);
    return;
  }
  for (int r = 1; r <= M.R; r++) {
    u64 A = 0;
    A = a_set_status(A, ST_NORMAL);        // rep_status = Normal            :328
    A = a_set_view(A, 1);                  // rep_view_number = 1            :330
    for (int c = 1; c <= M.C; c++) A = a_set_ctrow(A, c, ct_make(0, 0, 1));   // EmptyClientTableRow :318-321
    rec[1 + (r - 1) * M.wpr] = A;          // everything else 0 / empty      :329-343
  }
}

// view hashes of a device-layout record on the host (pure arithmetic, the same functions the kernels run)
// identity of the fingerprint function in a checkpoint header: its version, and 23 bits of the seed's hash when the model carries one
int32_t fp_function_id(const Model& M) {
  return (int32_t)VSRMC_FP_VERSION | (M.fp_seed ? (int32_t)((fmix64(M.fp_seed) & 0x7FFFFF) << 8) : 0);
}
void hash_full_host(const Model& M, const u64* rec, u64* H) {
  if (M.model_id == 1) vrst::hash_full(M, rec, H);
  else if (M.model_id == 2) vras::hash_full(M, rec, H);
  else hash_full(M, rec, H);
}

}  // namespace

extern "C" {

const char* vsrmc_last_error(void) { return g_err.c_str(); }
int32_t vsrmc_version(void) { return 100; }
int32_t vsrmc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int32_t vsrmc_model_from_constants(int32_t R, int32_t C, int32_t n, int32_t L, int32_t restart, int32_t symmetry,
                                   int32_t inv_mask, int32_t assume_commit, vsrmc_model** out) {
  if (!out) return fail(VSRMC_E_ARG, "out is NULL");
  vsrmc_model* m = new vsrmc_model();
  int rc = build_model(R, C, n, L, restart, symmetry, inv_mask, assume_commit, m);
  if (rc) { delete m; return rc; }
  *out = m;
  return 0;
}

int32_t vsrmc_model2_from_constants(int32_t R, int32_t n, int32_t L, int32_t no_progress_limit, int32_t symmetry, int32_t inv_mask,
                                    vsrmc_model** out) {
  if (!out) return fail(VSRMC_E_ARG, "out is NULL");
  vsrmc_model* m = new vsrmc_model();
  int rc = build_model2(R, n, L, no_progress_limit, symmetry, inv_mask, m);
  if (rc) { delete m; return rc; }
  *out = m;
  return 0;
}

int32_t vsrmc_model3_from_constants(int32_t R, int32_t n, int32_t L, int32_t no_progress_limit, int32_t symmetry, int32_t inv_mask,
                                    vsrmc_model** out) {
  if (!out) return fail(VSRMC_E_ARG, "out is NULL");
  vsrmc_model* m = new vsrmc_model();
  int rc = build_model3(R, n, L, no_progress_limit, symmetry, inv_mask, m);
  if (rc) { delete m; return rc; }
  *out = m;
  return 0;
}

// The TLC cfg grammar as used by VSR.cfg:1-39: CONSTANTS (name = int | name = {mv, ...} | name = mv), INIT, NEXT,
// VIEW, SYMMETRY, INVARIANT[S] (multi-line list), CHECK_DEADLOCK, `\*` comments.  SPECIFICATION / PROPERTY are refused.
int32_t vsrmc_model_load(const char* tla_path, const char* cfg_path, vsrmc_model** out) {
  if (!cfg_path || !out) return fail(VSRMC_E_ARG, "cfg_path / out is NULL");
  int module = -1;                                               // 0 = VSR.tla, 1 = VR_STATE_TRANSFER.tla, 2 = VR_APP_STATE.tla, -1 = decided by the cfg
  if (tla_path) {
    module = 0;
    std::ifstream f(tla_path, std::ios::binary);
    if (!f) return fail(VSRMC_E_CFG, std::string("cannot read ") + tla_path);
    std::stringstream ss;
    ss << f.rdbuf();
    std::string dig = sha256_hex(ss.str());
    if (dig == VRST_TLA_SHA256) module = 1;
    else if (dig == VRAS_TLA_SHA256) module = 2;
    else if (dig != VSR_TLA_SHA256)
      return fail(VSRMC_E_CFG, std::string(tla_path) + ": sha256 " + dig + " is neither the VSR.tla (" + VSR_TLA_SHA256 +
                                   "), the VR_STATE_TRANSFER.tla (" + VRST_TLA_SHA256 + ") nor the VR_APP_STATE.tla (" + VRAS_TLA_SHA256 +
                                   ") this build lowers; refusing to check a module the action table was not derived from");
  }
  std::ifstream f(cfg_path);
  if (!f) return fail(VSRMC_E_CFG, std::string("cannot read ") + cfg_path);
  std::map<std::string, std::string> consts;
  std::vector<std::string> invariants;
  std::string init, next, view, symmetry, spec, line, section;
  int check_deadlock = 0;   // TLC's default is TRUE; the BASELINE runs use -deadlock, see DESIGN.md
  int lineno = 0;
  static const char* KW[] = {"CONSTANTS", "CONSTANT", "INIT", "NEXT", "VIEW", "SYMMETRY", "INVARIANTS", "INVARIANT",
                             "SPECIFICATION", "PROPERTIES", "PROPERTY", "CHECK_DEADLOCK", "CONSTRAINT", "CONSTRAINTS",
                             "ACTION_CONSTRAINT", "ACTION_CONSTRAINTS", "ALIAS", "POSTCONDITION"};
  while (std::getline(f, line)) {
    lineno++;
    size_t cpos = line.find("\\*");
    if (cpos != std::string::npos) line = line.substr(0, cpos);
    std::string rest = strip(line);
    while (!rest.empty()) {
      // leading keyword?
      std::string tok = rest.substr(0, rest.find_first_of(" \t"));
      bool is_kw = false;
      for (const char* k : KW)
        if (tok == k) is_kw = true;
      if (is_kw) {
        section = tok;
        rest = strip(rest.substr(tok.size()));
        if (section == "SPECIFICATION") continue;                  // VR_STATE_TRANSFER.cfg:21 `SPECIFICATION Spec`; checked below
        if (section == "PROPERTY" || section == "PROPERTIES" || section == "CONSTRAINT" ||
            section == "CONSTRAINTS" || section == "ACTION_CONSTRAINT" || section == "ACTION_CONSTRAINTS" ||
            section == "ALIAS" || section == "POSTCONDITION")
          return fail(VSRMC_E_CFG, std::string(cfg_path) + ":" + std::to_string(lineno) + ": " + section +
                                       " is not supported (only INIT/NEXT safety checking of VSR.tla is lowered)");
        continue;
      }
      if (section == "CONSTANTS" || section == "CONSTANT") {
        size_t eq = rest.find('=');
        if (eq == std::string::npos)
          return fail(VSRMC_E_CFG, std::string(cfg_path) + ":" + std::to_string(lineno) + ": expected `name = value`");
        std::string name = strip(rest.substr(0, eq)), val = strip(rest.substr(eq + 1));
        consts[name] = val;
        rest.clear();
      } else if (section == "SPECIFICATION") { spec = tok; rest = strip(rest.substr(tok.size())); }
      else if (section == "INIT") { init = tok; rest = strip(rest.substr(tok.size())); }
      else if (section == "NEXT") { next = tok; rest = strip(rest.substr(tok.size())); }
      else if (section == "VIEW") { view = tok; rest = strip(rest.substr(tok.size())); }
      else if (section == "SYMMETRY") { symmetry = tok; rest = strip(rest.substr(tok.size())); }
      else if (section == "INVARIANT" || section == "INVARIANTS") { invariants.push_back(tok); rest = strip(rest.substr(tok.size())); }
      else if (section == "CHECK_DEADLOCK") { check_deadlock = (tok == "TRUE"); rest = strip(rest.substr(tok.size())); }
      else
        return fail(VSRMC_E_CFG, std::string(cfg_path) + ":" + std::to_string(lineno) + ": unexpected text `" + rest + "`");
    }
  }
  auto need_int = [&](const char* name, int* v) -> bool {
    auto it = consts.find(name);
    if (it == consts.end()) return false;
    char* end = nullptr;
    long x = std::strtol(it->second.c_str(), &end, 10);
    if (end == it->second.c_str() || *end) return false;
    *v = (int)x;
    return true;
  };
  // no module given: the cfg's constants tell VSR.cfg from the analysis cfgs; of those two only VR_APP_STATE has NoAppStateDivergence
  if (module < 0) {
    module = consts.count("NoProgressChangeLimit") ? 1 : 0;
    for (const std::string& iv : invariants)
      if (module == 1 && iv == "NoAppStateDivergence") module = 2;
  }
  if (module == 1 || module == 2) {  // ---- VR_STATE_TRANSFER.cfg / VR_APP_STATE.cfg (the same constants and sections)
    int R2, L2, npl;
    if (!need_int("ReplicaCount", &R2) || !need_int("StartViewOnTimerLimit", &L2) || !need_int("NoProgressChangeLimit", &npl))
      return fail(VSRMC_E_CFG, std::string(cfg_path) + ": CONSTANTS must bind ReplicaCount, StartViewOnTimerLimit, "
                                   "NoProgressChangeLimit to integers (VR_STATE_TRANSFER.cfg:4-7)");
    std::vector<std::string> values2;
    auto itv = consts.find("Values");
    if (itv == consts.end() || itv->second.size() < 2 || itv->second.front() != '{' || itv->second.back() != '}')
      return fail(VSRMC_E_CFG, std::string(cfg_path) + ": CONSTANTS must bind Values to a set of model values (VR_STATE_TRANSFER.cfg:5)");
    {
      std::string body = itv->second.substr(1, itv->second.size() - 2), item;
      std::stringstream ss(body);
      while (std::getline(ss, item, ',')) {
        item = strip(item);
        if (!item.empty()) values2.push_back(item);
      }
    }
    static const char* SELF2[] = {"Normal", "ViewChange", "StateTransfer", "PrepareMsg", "PrepareOkMsg", "StartViewChangeMsg",
                                  "DoViewChangeMsg", "StartViewMsg", "GetStateMsg", "NewStateMsg", "Nil", "AnyDest"};
    for (const char* sname : SELF2) {
      auto it = consts.find(sname);
      if (it == consts.end() || it->second != sname)
        return fail(VSRMC_E_CFG, std::string(cfg_path) + ": constant " + sname + " must be bound to the model value " + sname +
                                     " (VR_STATE_TRANSFER.cfg:8-19)");
    }
    if (!((spec == "Spec" && init.empty() && next.empty()) || (spec.empty() && init == "Init" && next == "Next")))
      return fail(VSRMC_E_CFG, std::string(cfg_path) + ": expected SPECIFICATION Spec (VR_STATE_TRANSFER.cfg:21; LivenessSpec and "
                                   "PROPERTY checking are not lowered) or INIT Init / NEXT Next");
    if (view != "view") return fail(VSRMC_E_CFG, std::string(cfg_path) + ": expected VIEW view (VR_STATE_TRANSFER.cfg:23)");
    int mask2 = 0;
    for (const std::string& iv : invariants) {
      if (iv == "AcknowledgedWriteNotLost") mask2 |= 1;                  // VR_STATE_TRANSFER.tla:830-835
      else if (iv == "AcknowledgedWritesExistOnMajority") mask2 |= 2;   // :818-824
      else if (iv == "NoLogDivergence") mask2 |= 4;                     // :806-811
      else if (iv == "CommitNumberNeverHigherThanOpNumber") mask2 |= 8; // :845-847
      else if (iv == "NoAppStateDivergence" && module == 2) mask2 |= 16;   // VR_APP_STATE.tla:852-858
      else if (iv == "TestInv") mask2 |= 0;                             // :849 (TRUE)
      else return fail(VSRMC_E_CFG, std::string(cfg_path) + ": unknown INVARIANT " + iv);
    }
    vsrmc_model* m2 = new vsrmc_model();
    int rc2 = module == 2 ? build_model3(R2, (int)values2.size(), L2, npl, symmetry.empty() ? 0 : 1, mask2, m2)
                          : build_model2(R2, (int)values2.size(), L2, npl, symmetry.empty() ? 0 : 1, mask2, m2);
    if (rc2) { delete m2; return rc2; }
    m2->value_names = values2;
    m2->check_deadlock = check_deadlock;
    *out = m2;
    return 0;
  }
  if (!spec.empty())
    return fail(VSRMC_E_CFG, std::string(cfg_path) + ": SPECIFICATION is not supported for VSR.tla (VSR.cfg:26-27 uses INIT / NEXT)");
  int R, C, L, restart;
  if (!need_int("ReplicaCount", &R) || !need_int("ClientCount", &C) || !need_int("StartViewOnTimerLimit", &L) ||
      !need_int("RestartEmptyLimit", &restart))
    return fail(VSRMC_E_CFG, std::string(cfg_path) + ": CONSTANTS must bind ReplicaCount, ClientCount, "
                                 "StartViewOnTimerLimit, RestartEmptyLimit to integers (VSR.cfg:4-8)");
  std::vector<std::string> values;
  {
    auto it = consts.find("Values");
    if (it == consts.end() || it->second.size() < 2 || it->second.front() != '{' || it->second.back() != '}')
      return fail(VSRMC_E_CFG, std::string(cfg_path) + ": CONSTANTS must bind Values to a set of model values (VSR.cfg:6)");
    std::string body = it->second.substr(1, it->second.size() - 2), item;
    std::stringstream ss(body);
    while (std::getline(ss, item, ',')) {
      item = strip(item);
      if (!item.empty()) values.push_back(item);
    }
  }
  // the self-named model values of VSR.cfg:9-24
  static const char* SELF[] = {"Normal", "ViewChange", "Recovering", "RequestMsg", "ReplyMsg", "PrepareMsg", "PrepareOkMsg",
                               "CommitMsg", "StartViewChangeMsg", "DoViewChangeMsg", "StartViewMsg", "GetStateMsg",
                               "NewStateMsg", "RecoveryMsg", "RecoveryResponseMsg", "Nil"};
  for (const char* s : SELF) {
    auto it = consts.find(s);
    if (it == consts.end() || it->second != s)
      return fail(VSRMC_E_CFG, std::string(cfg_path) + ": constant " + s + " must be bound to the model value " + s +
                                   " (VSR.cfg:9-24)");
  }
  if (init != "Init" || next != "Next")
    return fail(VSRMC_E_CFG, std::string(cfg_path) + ": expected INIT Init / NEXT Next (VSR.cfg:26-27)");
  if (view != "view")
    return fail(VSRMC_E_CFG, std::string(cfg_path) + ": expected VIEW view (VSR.cfg:29); state identity without the view is not lowered");
  if (!symmetry.empty() && symmetry != "symmValues")
    return fail(VSRMC_E_CFG, std::string(cfg_path) + ": SYMMETRY must be symmValues (VSR.cfg:31)");
  int inv_mask = 0;
  for (const std::string& iv : invariants) {
    if (iv == "AcknowledgedWriteNotLost") inv_mask |= 1;            // VSR.tla:945-950
    else if (iv == "AcknowledgedWritesExistOnMajority") inv_mask |= 2;   // VSR.tla:937-943
    else if (iv == "NoLogDivergence") inv_mask |= 4;                // VSR.tla:926-931 (vacuous)
    else if (iv == "TestInv") inv_mask |= 8;                        // VSR.tla:952
    else return fail(VSRMC_E_CFG, std::string(cfg_path) + ": unknown INVARIANT " + iv);
  }
  vsrmc_model* m = new vsrmc_model();
  int rc = build_model(R, C, (int)values.size(), L, restart, symmetry.empty() ? 0 : 1, inv_mask, 0, m);
  if (rc) { delete m; return rc; }
  m->value_names = values;
  m->check_deadlock = check_deadlock;
  *out = m;
  return 0;
}

int32_t vsrmc_model_set_fp_seed(vsrmc_model* m, uint64_t seed) {
  if (!m) return fail(VSRMC_E_ARG, "NULL argument");
  m->M.fp_seed = seed;
  return 0;
}
uint64_t vsrmc_model_fp_seed(const vsrmc_model* m) { return m ? m->M.fp_seed : 0; }

int32_t vsrmc_model_info(const vsrmc_model* m, vsrmc_layout* out) {
  if (!m || !out) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(out, 0, sizeof(*out));
  const Model& M = m->M;
  out->replica_count = M.R; out->client_count = M.C; out->value_count = M.n; out->start_view_on_timer_limit = M.L;
  out->symmetry = m->symmetry; out->invariant_mask = M.inv_mask; out->assume_commit_number = M.assume_commit;
  out->check_deadlock = m->check_deadlock;
  out->words_per_replica = M.wpr; out->fixed_words = M.h0; out->permutations = M.np; out->max_bag = M.max_bag;
  out->max_record_words = 256;   // wire-layout upper bound (8-bit length); BFS records are bounded by max_bag
  out->module = M.model_id;
  return 0;
}

int32_t vsrmc_model_init_state(const vsrmc_model* m, uint64_t* rec, int32_t cap, int32_t* n_words) {
  if (!m || !rec || !n_words) return fail(VSRMC_E_ARG, "NULL argument");
  std::vector<u64> r;
  init_record_wire(m->M, r);
  if ((int)r.size() > cap) return fail(VSRMC_E_ARG, "buffer too small");
  std::copy(r.begin(), r.end(), rec);
  *n_words = (int32_t)r.size();
  return 0;
}

int32_t vsrmc_model_format_state(const vsrmc_model* m, const uint64_t* rec, char* buf, int64_t cap, int64_t* n) {
  if (!m || !rec || !n) return fail(VSRMC_E_ARG, "NULL argument");
  std::string s = m->M.model_id == 1   ? vrst::format_state_tlc(m->M, m->value_names, rec)
                  : m->M.model_id == 2 ? vras::format_state_tlc(m->M, m->value_names, rec)
                                       : format_state_tlc(m->M, m->value_names, rec);
  *n = (int64_t)s.size() + 1;
  if (buf && cap >= *n) std::memcpy(buf, s.c_str(), s.size() + 1);
  else if (buf && cap > 0) return fail(VSRMC_E_ARG, "buffer too small");
  return 0;
}

const char* vsrmc_action_name(int32_t a) {
  static const char* const NAMES[16] = {"Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC",
                                        "SendDVC", "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV",
                                        "ReceiveClientRequest", "ReceivePrepareMsg", "ReceivePrepareOkMsg", "ExecuteOp",
                                        "SendGetState", "ReceiveGetState", "ReceiveNewState"};   // VSR.tla:896-913
  return (a >= 0 && a < 16) ? NAMES[a] : "?";
}

void vsrmc_model_destroy(vsrmc_model* m) { delete m; }

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// FPSet
// ---------------------------------------------------------------------------------------------------------------
struct vsrmc_fpset {
  int device = 0;
  u64 slots = 0;
  Slot* table = nullptr;
  u64* d_size = nullptr;
  u32* d_err = nullptr;
};

namespace {
int check_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
    return fail(VSRMC_E_HIP, "no HIP device: libvsrmc has no CPU fallback (the GPU path is the product)");
  if (device < 0 || device >= n) return fail(VSRMC_E_ARG, "device ordinal out of range");
  HIPCHK(hipSetDevice(device));
  return 0;
}
}  // namespace

extern "C" {

int32_t vsrmc_fpset_create(int32_t device, int32_t log2_slots, vsrmc_fpset** out) {
  if (!out || log2_slots < 4 || log2_slots > 36) return fail(VSRMC_E_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  vsrmc_fpset* s = new vsrmc_fpset();
  s->device = device;
  s->slots = (u64)1 << log2_slots;
  hipError_t e = hipMalloc((void**)&s->table, s->slots * sizeof(Slot));
  if (e == hipSuccess) e = hipMalloc((void**)&s->d_size, 16);
  if (e != hipSuccess) { delete s; return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  s->d_err = (u32*)(s->d_size + 1);
  HIPCHK(hipMemset(s->table, 0, s->slots * sizeof(Slot)));
  HIPCHK(hipMemset(s->d_size, 0, 16));
  *out = s;
  return 0;
}

int32_t vsrmc_fpset_put_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_was, void* stream) {
  if (!s) return fail(VSRMC_E_ARG, "NULL handle");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(k_fpset_put, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->table, s->slots - 1,
                     d_fps, n, d_was, s->d_size, s->d_err);
  HIPCHK(hipGetLastError());
  return 0;
}
int32_t vsrmc_fpset_contains_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_present, void* stream) {
  if (!s) return fail(VSRMC_E_ARG, "NULL handle");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  hipLaunchKernelGGL(k_fpset_contains, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s->table,
                     s->slots - 1, d_fps, n, d_present);
  HIPCHK(hipGetLastError());
  return 0;
}

static int fpset_host_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* res, bool put) {
  if (!s || (n && (!fps || !res))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(s->device));
  u64* d_fps = nullptr;
  uint8_t* d_res = nullptr;
  HIPCHK(hipMalloc((void**)&d_fps, n * 8));
  hipError_t e = hipMalloc((void**)&d_res, n);
  if (e != hipSuccess) { (void)hipFree(d_fps); return fail(VSRMC_E_HIP, "hipMalloc"); }
  int rc = 0;
  e = hipMemcpy(d_fps, fps, n * 8, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    rc = put ? vsrmc_fpset_put_batch_device(s, d_fps, n, d_res, nullptr) : vsrmc_fpset_contains_batch_device(s, d_fps, n, d_res, nullptr);
    if (!rc) e = hipMemcpy(res, d_res, n, hipMemcpyDeviceToHost);
  }
  u32 err = 0;
  if (e == hipSuccess) e = hipMemcpy(&err, s->d_err, 4, hipMemcpyDeviceToHost);
  (void)hipFree(d_fps);
  (void)hipFree(d_res);
  if (rc) return rc;
  if (e != hipSuccess) return fail(VSRMC_E_HIP, std::string("hipMemcpy: ") + hipGetErrorString(e));
  if (err) return fail(VSRMC_E_REP, "fingerprint set is full");
  return 0;
}
int32_t vsrmc_fpset_put_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* was_present) {
  return fpset_host_batch(s, fps, n, was_present, true);
}
int32_t vsrmc_fpset_contains_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* present) {
  return fpset_host_batch(s, fps, n, present, false);
}
int32_t vsrmc_fpset_size(vsrmc_fpset* s, uint64_t* size) {
  if (!s || !size) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(s->device));
  HIPCHK(hipMemcpy(size, s->d_size, 8, hipMemcpyDeviceToHost));
  return 0;
}
void vsrmc_fpset_destroy(vsrmc_fpset* s) {
  if (!s) return;
  if (s->table) (void)hipFree(s->table);
  if (s->d_size) (void)hipFree(s->d_size);
  delete s;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// expand_batch / fingerprint_batch
// ---------------------------------------------------------------------------------------------------------------
namespace {

// upload n wire records as device-layout records with their H words filled in
int upload_records(const Model& M, const u64* words, const u64* off, u64 n, u64** d_words, u64** d_off, u64* total_words) {
  std::vector<u64> dev, doff(n + 1);
  dev.reserve((size_t)(off[n] + n * M.np));
  std::vector<u64> tmp(512);
  for (u64 i = 0; i < n; i++) {
    doff[i] = dev.size();
    const u64* w = words + off[i];
    int nmsg = hdr_nmsg(w[0]);
    if ((u64)(M.h0 + nmsg) != off[i + 1] - off[i]) return fail(VSRMC_E_ARG, "record length does not match its header");
    if (nmsg > M.max_bag) return fail(VSRMC_E_REP, "record bag larger than max_bag");
    int len = wire_to_device(M, w, tmp.data());
    dev.insert(dev.end(), tmp.begin(), tmp.begin() + len);
  }
  doff[n] = dev.size();
  *total_words = dev.size();
  HIPCHK(hipMalloc((void**)d_words, std::max<size_t>(dev.size(), 1) * 8));
  HIPCHK(hipMalloc((void**)d_off, (n + 1) * 8));
  HIPCHK(hipMemcpy(*d_words, dev.data(), dev.size() * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(*d_off, doff.data(), (n + 1) * 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((M.model_id == 1 ? k_hash_records<1> : M.model_id == 2 ? k_hash_records<2> : k_hash_records<0>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, M, *d_words, *d_off, n);
  HIPCHK(hipGetLastError());
  return 0;
}

}  // namespace

extern "C" {

int32_t vsrmc_expand_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                           uint64_t* out_words, uint64_t out_words_cap, uint64_t* out_meta, uint64_t out_cap,
                           uint64_t* n_out, uint64_t* words_out) {
  if (!m || !words || !off || !out_words || !out_meta || !n_out || !words_out) return fail(VSRMC_E_ARG, "NULL argument");
  int rc = check_device(device);
  if (rc) return rc;
  const Model& M = m->M;
  *n_out = 0;
  *words_out = 0;
  if (n == 0) return 0;
  u64 *d_words = nullptr, *d_off = nullptr, *d_ow = nullptr, *d_om = nullptr, *d_cnt = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  u64 dev_words_cap = out_words_cap + out_cap * (u64)M.np;
  HIPCHK(hipMalloc((void**)&d_ow, std::max<u64>(dev_words_cap, 1) * 8));
  HIPCHK(hipMalloc((void**)&d_om, std::max<u64>(out_cap, 1) * 64));
  HIPCHK(hipMalloc((void**)&d_cnt, 32));
  HIPCHK(hipMemset(d_cnt, 0, 32));
  hipLaunchKernelGGL((M.model_id == 1 ? k_successors<1> : M.model_id == 2 ? k_successors<2> : k_successors<0>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, M, d_words, d_off, n, d_ow, dev_words_cap,
                     d_om, out_cap, d_cnt);
  HIPCHK(hipGetLastError());
  u64 cnt[4];
  HIPCHK(hipMemcpy(cnt, d_cnt, 32, hipMemcpyDeviceToHost));
  int ret = 0;
  if (cnt[2]) {
    ret = fail(VSRMC_E_ARG, "successor buffers too small");
  } else {
    std::vector<u64> hw(cnt[1]), hm(cnt[0] * 8);
    HIPCHK(hipMemcpy(hw.data(), d_ow, cnt[1] * 8, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(hm.data(), d_om, cnt[0] * 64, hipMemcpyDeviceToHost));
    // deterministic (parent, ordinal) order
    std::vector<u64> order(cnt[0]);
    for (u64 k = 0; k < cnt[0]; k++) order[k] = k;
    std::sort(order.begin(), order.end(), [&](u64 a, u64 b) {
      if (hm[8 * a] != hm[8 * b]) return hm[8 * a] < hm[8 * b];
      return hm[8 * a + 1] < hm[8 * b + 1];
    });
    u64 wpos = 0;
    for (u64 k = 0; k < cnt[0] && !ret; k++) {
      const u64* mm = &hm[8 * order[k]];
      const u64* dev = &hw[mm[7]];
      int evalerr = (int)mm[6];
      int wl = evalerr ? 0 : M.h0 + hdr_nmsg(dev[0]);
      if (wpos + (u64)wl > out_words_cap) { ret = fail(VSRMC_E_ARG, "successor word buffer too small"); break; }
      if (!evalerr) device_to_wire(M, dev, out_words + wpos);
      for (int q = 0; q < 7; q++) out_meta[8 * k + q] = mm[q];
      out_meta[8 * k + 7] = wpos;
      wpos += (u64)wl;
    }
    *n_out = cnt[0];
    *words_out = wpos;
  }
  (void)hipFree(d_words); (void)hipFree(d_off); (void)hipFree(d_ow); (void)hipFree(d_om); (void)hipFree(d_cnt);
  return ret;
}

// ---- TLC trace / state import: text in TLC's value syntax -> wire records ---------------------------------
int32_t vsrmc_model_parse_states(const vsrmc_model* m, const char* text, uint64_t* words, uint64_t cap_words, uint64_t* off,
                                 int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!m || !text || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  std::vector<ParsedState> st;
  std::string err;
  if (!parse_states_tlc(m->M, m->value_names, text, &st, &err)) return fail(VSRMC_E_CFG, "TLC state text: " + err);
  *n_states = st.size();
  u64 total = 0;
  for (const ParsedState& s : st) total += s.rec.size();
  if (!words || !off) return 0;                                 // size query
  if (cap_states < st.size() + 1 || cap_words < total) return fail(VSRMC_E_ARG, "buffer too small");
  u64 pos = 0;
  for (size_t i = 0; i < st.size(); i++) {
    off[i] = pos;
    std::copy(st[i].rec.begin(), st[i].rec.end(), words + pos);
    pos += st[i].rec.size();
    if (actions) {
      actions[i] = -1;
      for (int a = 0; a < 16; a++)
        if (st[i].action == vsrmc_action_name(a)) actions[i] = a;
    }
  }
  off[st.size()] = pos;
  return 0;
}

// Is the sequence of states a behaviour of the model?  State 0 must be Init; every later state must be among the successors
// the GPU generates for its predecessor (k_successors, the same gen() the BFS kernels run).  ords[i] / actions[i + 1] = the
// (action, binding) ordinal and action id of the step into state i + 1; *first_bad = index of the first state that does not
// follow (or -1); *inv_mask_last = invariants violated by the last state (when the whole sequence is legal).
int32_t vsrmc_model_check_trace(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n_states,
                                uint32_t* ords, int32_t* actions, int64_t* first_bad, int32_t* inv_mask_last) {
  if (!m || !words || !off || !first_bad) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  *first_bad = -1;
  if (inv_mask_last) *inv_mask_last = 0;
  if (n_states == 0) return 0;
  auto normal = [&](const u64* rec, u64 len) {
    std::vector<u64> v(rec, rec + len);
    if (len > (u64)M.h0) std::sort(v.begin() + M.h0, v.end());
    return v;
  };
  std::vector<u64> init;
  init_record_wire(M, init);
  if (normal(words + off[0], off[1] - off[0]) != init) {
    *first_bad = 0;
    return 0;
  }
  if (actions) actions[0] = 0;
  if (n_states == 1) return 0;
  // Walk like vsrmc_model_replay does: the state that is expanded next is the successor as the GPU produced it (its bag order
  // defines the ordinals of the message-bound actions), the text's states are only compared against.
  const u64 cap = 2048, capw = cap * 64;
  std::vector<u64> ow(capw), om(cap * 8), cur = init;
  for (u64 i = 0; i + 1 < n_states; i++) {
    const std::vector<u64> want = normal(words + off[i + 1], off[i + 2] - off[i + 1]);
    const u64 coff[2] = {0, cur.size()};
    u64 n_out = 0, w_out = 0;
    int rc = vsrmc_expand_batch(m, device, cur.data(), coff, 1, ow.data(), capw, om.data(), cap, &n_out, &w_out);
    if (rc) return rc;
    bool found = false;
    for (u64 k = 0; k < n_out && !found; k++) {
      if (om[8 * k + 6]) continue;                             // an instance that raises an evaluation error has no successor
      const u64 wo = om[8 * k + 7];
      const u64 len = (u64)M.h0 + (u64)hdr_nmsg(ow[wo]);
      if (normal(&ow[wo], len) != want) continue;
      found = true;
      if (ords) ords[i] = (uint32_t)om[8 * k + 1];
      if (actions) actions[i + 1] = (int32_t)om[8 * k + 2];
      if (i + 2 == n_states && inv_mask_last) *inv_mask_last = (int32_t)om[8 * k + 5];
      cur.assign(&ow[wo], &ow[wo] + len);
    }
    if (!found) {
      *first_bad = (int64_t)(i + 1);
      return 0;
    }
  }
  return 0;
}

int32_t vsrmc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                                uint64_t* fps, uint32_t* auxkeys) {
  if (!m || !words || !off || !fps) return fail(VSRMC_E_ARG, "NULL argument");
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return 0;
  const Model& M = m->M;
  u64 *d_words = nullptr, *d_off = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  std::vector<u64> dev(total), doff(n + 1);
  HIPCHK(hipMemcpy(dev.data(), d_words, total * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(doff.data(), d_off, (n + 1) * 8, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < n; i++) {   // the hashes were computed on the GPU (k_hash_records); only the final min is here
    u64 fp;
    u32 ak;
    canonical_fp(M, dev[doff[i]], &dev[doff[i] + M.h0], &fp, &ak);
    fps[i] = fp;
    if (auxkeys) auxkeys[i] = ak;
  }
  (void)hipFree(d_words); (void)hipFree(d_off);
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// StateQueue: a ring of record words + a ring of (offset, length) refs, both in HBM
// ---------------------------------------------------------------------------------------------------------------
struct vsrmc_queue {
  int device = 0;
  u64 cap_words = 0, cap_states = 0;
  u64* words = nullptr;
  std::vector<std::pair<u64, u32>> refs;   // (word position in the ring, length) of every queued record, oldest at `head`
  u64 head = 0;
  u64 wpos = 0;                            // next write position in the word ring
  u64 wtail() const { return head < refs.size() ? refs[head].first : wpos; }   // position of the oldest record
};

extern "C" {

int32_t vsrmc_queue_create(int32_t device, uint64_t capacity_words, uint64_t capacity_states, vsrmc_queue** out) {
  if (!out || capacity_words < 256 || capacity_states < 1) return fail(VSRMC_E_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  vsrmc_queue* q = new vsrmc_queue();
  q->device = device;
  q->cap_words = capacity_words;
  q->cap_states = capacity_states;
  hipError_t e = hipMalloc((void**)&q->words, capacity_words * 8);
  if (e != hipSuccess) { delete q; return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *out = q;
  return 0;
}

int32_t vsrmc_queue_enqueue_batch(vsrmc_queue* q, const uint64_t* words, const uint64_t* off, uint64_t n) {
  if (!q || (n && (!words || !off))) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(q->device));
  if (q->refs.size() - q->head + n > q->cap_states) return fail(VSRMC_E_REP, "state queue full (states)");
  for (u64 i = 0; i < n; i++) {
    const u64 len = off[i + 1] - off[i];
    if (len == 0 || len > 255) return fail(VSRMC_E_ARG, "bad record length");
    if (q->head == q->refs.size()) { q->refs.clear(); q->head = 0; q->wpos = 0; }   // empty: start over at the front of the ring
    const u64 tail = q->wtail();
    u64 at;
    if (q->wpos >= tail) {                                    // data in [tail, wpos): append, or wrap to the front
      if (q->wpos + len <= q->cap_words) at = q->wpos;
      else if (len < tail) at = 0;                            // records never straddle the end of the ring
      else return fail(VSRMC_E_REP, "state queue full (words)");
    } else {                                                  // wrapped: data in [tail, cap) and [0, wpos)
      if (q->wpos + len < tail) at = q->wpos;
      else return fail(VSRMC_E_REP, "state queue full (words)");
    }
    HIPCHK(hipMemcpy(q->words + at, words + off[i], len * 8, hipMemcpyHostToDevice));
    q->refs.emplace_back(at, (u32)len);
    q->wpos = at + len;
  }
  return 0;
}

int32_t vsrmc_queue_dequeue_batch(vsrmc_queue* q, uint64_t max_states, uint64_t* words, uint64_t cap_words, uint64_t* off, uint64_t* n) {
  if (!q || !words || !off || !n) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(q->device));
  u64 k = 0, pos = 0;
  off[0] = 0;
  while (k < max_states && q->head < q->refs.size()) {
    const u64 wp = q->refs[q->head].first;
    const u32 len = q->refs[q->head].second;
    if (pos + len > cap_words) break;
    HIPCHK(hipMemcpy(words + pos, q->words + wp, (u64)len * 8, hipMemcpyDeviceToHost));
    pos += len;
    off[++k] = pos;
    q->head++;
  }
  if (q->head > 4096 && q->head * 2 > q->refs.size()) {       // drop the consumed prefix of the ref list now and then
    q->refs.erase(q->refs.begin(), q->refs.begin() + (long)q->head);
    q->head = 0;
  }
  *n = k;
  return 0;
}

int32_t vsrmc_queue_size(vsrmc_queue* q, uint64_t* n_states) {
  if (!q || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  *n_states = q->refs.size() - q->head;
  return 0;
}

void vsrmc_queue_destroy(vsrmc_queue* q) {
  if (!q) return;
  (void)hipSetDevice(q->device);
  if (q->words) (void)hipFree(q->words);
  delete q;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------
// simulation mode
// ---------------------------------------------------------------------------------------------------------------
extern "C" int32_t vsrmc_simulate(const vsrmc_model* m, int32_t device, uint32_t n_walkers, int32_t max_depth, uint64_t seed,
                                  double max_seconds, vsrmc_sim_result* out) {
  if (!m || !out || n_walkers == 0 || max_depth < 1 || max_depth > 512) return fail(VSRMC_E_ARG, "bad argument (max_depth 1..512)");
  int rc = check_device(device);
  if (rc) return rc;
  Model M = m->M;
  M.max_bag = 255 - M.fixed;     // walkers live in HBM, not in LDS tiles: the bag may grow to what the 8-bit count can hold
  std::memset(out, 0, sizeof(*out));
  const int stride = M.fixed + M.max_bag;
  std::vector<u64> wire, dev(512);
  init_record_wire(M, wire);
  const int len = wire_to_device(M, wire.data(), dev.data());   // the H words stay 0: simulation never fingerprints
  u64 *d_init = nullptr, *d_words = nullptr, *d_rng = nullptr;
  u32* d_depth = nullptr;
  u16* d_ords = nullptr;
  SimCtl* d_ctl = nullptr;
  HIPCHK(hipMalloc((void**)&d_init, 512 * 8));
  HIPCHK(hipMalloc((void**)&d_words, (u64)n_walkers * stride * 8));
  HIPCHK(hipMalloc((void**)&d_rng, (u64)n_walkers * 8));
  HIPCHK(hipMalloc((void**)&d_depth, (u64)n_walkers * 4));
  HIPCHK(hipMalloc((void**)&d_ords, (u64)n_walkers * max_depth * 2));
  HIPCHK(hipMalloc((void**)&d_ctl, sizeof(SimCtl)));
  HIPCHK(hipMemcpy(d_init, dev.data(), len * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_depth, 0xFF, (u64)n_walkers * 4));
  HIPCHK(hipMemset(d_ctl, 0, sizeof(SimCtl)));
  std::vector<u64> rng(n_walkers);
  u64 x = seed;
  for (u32 i = 0; i < n_walkers; i++) {   // splitmix64 stream: one non-zero xorshift state per walker
    x += 0x9E3779B97F4A7C15ULL;
    u64 z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z ^= z >> 31;
    rng[i] = z ? z : 1;
  }
  HIPCHK(hipMemcpy(d_rng, rng.data(), (u64)n_walkers * 8, hipMemcpyHostToDevice));
  SimCtl h;
  typedef void (*SimKernel)(Model, const u64*, int, u64*, int, u32*, u16*, u64*, u32, int, int, SimCtl*);
  SimKernel sim_kernel = k_simulate<0>;
  if (M.model_id == 1) sim_kernel = k_simulate<1000>;
  if (M.model_id == 2) sim_kernel = k_simulate<2000>;
  else
  switch (M.R * 100 + M.C * 10 + M.n) {                        // the same per-configuration instantiations as k_expand
    case 312: sim_kernel = k_simulate<312>; break;
    case 313: sim_kernel = k_simulate<313>; break;
    case 512: sim_kernel = k_simulate<512>; break;
    default: break;
  }
  double t0 = now_s();
  while (true) {
    hipLaunchKernelGGL(sim_kernel, dim3((n_walkers + 63) / 64), dim3(64), 0, 0, M, d_init, len, d_words, stride, d_depth, d_ords, d_rng,
                       n_walkers, max_depth, 64, d_ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpy(&h, d_ctl, sizeof(h), hipMemcpyDeviceToHost));
    if (h.found || now_s() - t0 > max_seconds) break;
  }
  out->seconds = now_s() - t0;
  out->found = (int32_t)h.found;
  out->steps = h.steps;
  out->walks = h.walks;
  if (h.found) {
    out->viol_mask = (int32_t)(h.viol_mask & 0x7FFFFFFFu);
    out->viol_steps = (int32_t)h.viol_depth;
    for (u32 k = 0; k < h.viol_depth && k < 512; k++) out->ords[k] = h.ords[k];
  }
  (void)hipFree(d_init); (void)hipFree(d_words); (void)hipFree(d_rng); (void)hipFree(d_depth); (void)hipFree(d_ords); (void)hipFree(d_ctl);
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// checker
// ---------------------------------------------------------------------------------------------------------------
struct PassDst {            // where a pass writes (records, refs, fingerprints)
  u64* words = nullptr;
  u64 words_cap = 0;
  u64* off = nullptr;
  u64* fp = nullptr;
  u64 cap = 0;
};
struct DeepLevelRec { u64 n_new = 0, n_local = 0, generated = 0, max_bag = 0, frontier = 0; };   // n_local: this rank's share (unsharded: all)   // a level that exists in the seen-set only (vsr_deep.hpp)
struct vsrmc_checker {
  vsrmc_model model;
  vsrmc_options opt;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  Slot* table = nullptr;
  u64 tmask = 0;
  u64* words[2] = {nullptr, nullptr};
  u64* off[2] = {nullptr, nullptr};
  u64* lvl_fp = nullptr;
  u64* pending = nullptr;
  LevelCtl* ctl = nullptr;
  u64* d_find = nullptr;
  int cur = 0;
  int level = 0;
  u64 n_frontier = 0;
  u64 distinct = 0, total_generated = 0;
  int num_cus = 256;
  int lds_stride = 65;
  int failed = 0;
  // TLCTrace: there is no separate log — a state's slot in the seen-set names its parent:
  // 45 bits of its parent's fingerprint (meta word, vsr_model.hpp); traces are walked through the table (k_trace_walk)
  // state of the level in flight (between the phases)
  LevelCtl h;
  double t_level0 = 0, expand_ms = 0, materialize_ms = 0;
  u64 nx_n = 0, nx_w = 0;
  u64 n_valid = 1;                       // states in the newest level (n_frontier is its index range, holes included)
  u64 cur_w = 0;                         // words of the current frontier buffer in use (chunk slack included)
  u64* rslot = nullptr;                  // sharded: slot of every received candidate
  u64 rslot_cap = 0;
  u64* filter = nullptr;                 // sharded single-pass levels: this rank's sent-filter (vsr_kernels.hpp, k_expand)
  u64 fmask = 0;
  u64* cand_idx = nullptr;               // ... and where each announced candidate was written (world x cand_cap)
  u64 cand_idx_cap = 0;
  bool level_fused = false;              // the level in flight is a single-pass level
  int failed_code = 0;                   // device ERR_* that stopped the search (failed == 1)
  // the single-pass kernel of this model (a specialised instantiation when there is one) and its launch shape
  void* fused_kernel = nullptr;
  void* plain_kernel = nullptr;          // the same without modes / sharding, when the configuration has one (ordinary unsharded levels)
  void* modes_kernel = nullptr;          // the same without sharding (expand_pass: the passes of vsrmc_checker_probe / _probe2 / _probe3), or null
  int plain_blk = VSR_BLOCK;             // threads per block of plain_kernel: 256, or 64 = one wave per block with a 16-record tile of its own
  u64 cur_max_bag = 0;                   // largest bag among the records of the newest level (LDS slot size of the next launch)
  bool bag_known = true;                 // false after a checkpoint was loaded or records arrived from other ranks: use the capacity
  // vsrmc_checker_probe / _probe2: where the reported violator's counter-example is walked from — the fingerprint of the deepest
  // state of the path that is IN the seen-set, its level, and the fingerprint of the one probed state beyond it (0: none)
  u64 probe_fp = 0, probe_extra_fp = 0;
  int probe_level = 0;
  int host_frontier = 0;                 // bit b: record buffer b lives in pinned host memory (zero-copy over PCIe)
  bool saw_violation = false;            // a committed level held a violating state (the caller went on): probe passes apply every action
  // levels beyond the record buffers (vsr_deep.hpp): `deep` levels above `level` are complete in the seen-set and have no frontier
  int deep = 0;
  std::vector<DeepLevelRec> deep_lv;     // [i] = level + 1 + i
  u64 deep_g = 2;                        // successors generated per expanded state, rounded up, the largest any level showed (worst-case slice sizes)
  u64 deep_distinct = 0, deep_generated = 0;
  bool deep_regen_done = false;          // a descent has set taken bits in the levels beyond the base: cleared before the next one
  std::vector<PassDst> scratch;          // scratch buffers of the descent, each a quarter of the one above down to a floor (allocated on first use)
  u64 hist_new[2] = {0, 0};              // new states of the last two levels (growth estimate of vsrmc_checker_advance)
  u64 g_last = 16, cur_rec_w = 0;        // successors generated per expanded state of the last level (rounded up, + 1); words of the newest level's records
  u64 words_cap(int b) const { return (b == 1 && opt.frontier_words_b) ? opt.frontier_words_b : opt.frontier_words; }
};

namespace {
typedef void (*ExpandKernel)(Model, const u64*, const u64*, u64, int, int, Slot*, u64, u64*, u64, LevelCtl*, int, int, u64*, u64, u32, u64*,
                             u64, u64*, u64, u64*, u32, u32, int, u32, u64*, u64, u64*, u32, int, u64);
// k_expand<true, SPEC>: the configurations of BASELINE.json (and their small neighbours used by the tests) have their own
// instantiation with the model constants folded in; anything else runs the generic one.
ExpandKernel exact_kernel_for(const Model& M) {               // two-kernel levels: k_expand<false, SPEC>
  if (M.model_id == 1) return k_expand<false, 1000>;
  if (M.model_id == 2) return k_expand<false, 2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_expand<false, 211>;
    case 312: return k_expand<false, 312>;
    case 313: return k_expand<false, 313>;
    case 512: return k_expand<false, 512>;
    default: return k_expand<false, 0>;
  }
}
typedef void (*MaterializeKernel)(Model, const u64*, const u64*, const u64*, u64, Slot*, u64*, u64, u64*, u64, u64*, LevelCtl*,
                                  const uint8_t*, u64*, u64*, int, u32, u32, int, const u64*);
MaterializeKernel materialize_kernel_for(const Model& M) {
  if (M.model_id == 1) return k_materialize<1000>;
  if (M.model_id == 2) return k_materialize<2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_materialize<211>;
    case 312: return k_materialize<312>;
    case 313: return k_materialize<313>;
    case 512: return k_materialize<512>;
    default: return k_materialize<0>;
  }
}
ExpandKernel plain_kernel_for(const Model& M, int blk) {      // unsharded ordinary levels: modes and sharding compiled out
  if (blk != VSR_BLOCK) return nullptr;
  if (M.model_id == 1) return (M.R == 3 && M.n == 2) ? k_expand<true, 1302, true> : nullptr;   // the shipped VR_STATE_TRANSFER.cfg
  if (M.model_id == 2) return (M.R == 3 && M.n == 2) ? k_expand<true, 2302, true> : nullptr;   // the shipped VR_APP_STATE.cfg
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return k_expand<true, 312, true>;
    case 313: return k_expand<true, 313, true>;
    case 512: return k_expand<true, 512, true>;
    default: return nullptr;
  }
}
ExpandKernel modes_kernel_for(const Model& M) {               // unsharded passes with a mode (probe / virtual / regenerated / streamed levels)
  if (M.model_id != 0) return nullptr;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 312: return k_expand<true, 312, 2>;
    case 313: return k_expand<true, 313, 2>;
    case 512: return k_expand<true, 512, 2>;
    default: return nullptr;
  }
}
ExpandKernel fused_kernel_for(const Model& M) {
  if (M.model_id == 1) return (M.R == 3 && M.n == 2) ? k_expand<true, 1302> : k_expand<true, 1000>;
  if (M.model_id == 2) return (M.R == 3 && M.n == 2) ? k_expand<true, 2302> : k_expand<true, 2000>;
  switch (M.R * 100 + M.C * 10 + M.n) {
    case 211: return k_expand<true, 211>;
    case 212: return k_expand<true, 212>;
    case 311: return k_expand<true, 311>;
    case 312: return k_expand<true, 312>;
    case 313: return k_expand<true, 313>;
    case 323: return k_expand<true, 323>;
    case 412: return k_expand<true, 412>;
    case 512: return k_expand<true, 512>;
    default: return k_expand<true, 0>;
  }
}
// Launch shape of the single-pass kernel for one launch.  The LDS slot of a record only has to hold the longest record of the
// level that is being expanded (stride = fixed words + its largest bag, made odd: conflict-free columns), not the format's
// worst case, so deep levels of small bags leave room for more resident blocks.  64-record tiles when that gives at least
// three blocks per CU (registers and LDS, asked from the runtime), else 128-record tiles (R <= 3) at two.
#ifndef VSR_CCAP64          // work-list entries of a 64-record tile, R <= 3 (24 per record; an overflow is ERR_FRONTIER_FULL, never silent)
#define VSR_CCAP64 1536
#endif
struct FusedShape {
  int blk;
  int tile;
  u32 ccap;
  int stride;
  size_t lds;
  unsigned blocks_per_cu;
};
FusedShape fused_shape(vsrmc_checker* c, u64 max_bag_of_source, bool plain = false) {
  const Model& M = c->model.M;
  FusedShape f;
  f.stride = (int)std::min<u64>((u64)c->lds_stride, (u64)((M.fixed + (int)std::min<u64>(max_bag_of_source, 255)) | 1));
  const void* kernel = (plain && c->plain_kernel) ? c->plain_kernel : c->fused_kernel;
  const int blk = (plain && c->plain_kernel) ? c->plain_blk : VSR_BLOCK;
  auto occupancy = [&](int tile, u32 ccap, size_t* lds) {
    *lds = (size_t)tile * f.stride * 8 + 2 * (size_t)ccap * 4;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, blk, *lds) != hipSuccess) nb = 0;
    return nb;
  };
  f.blk = blk;
  if (blk == 512) {
    f.tile = 128;
    f.ccap = 3072u;
    f.blocks_per_cu = (unsigned)std::max(1, std::min(occupancy(128, f.ccap, &f.lds), 2));
    return f;
  }
  if (blk == 64) {                                              // one wave, 16 records, 24 (R <= 3) or 32 work-list entries per record
    f.tile = 16;
    f.ccap = M.R <= 3 ? 384u : 512u;
    f.blocks_per_cu = (unsigned)std::max(1, std::min(occupancy(16, f.ccap, &f.lds), 16));
    return f;
  }
  size_t lds64 = 0, lds128 = 0;
  const u32 ccap64 = M.R <= 3 ? (u32)VSR_CCAP64 : (u32)VSR_CAND_CAP;      // work-list entries per tile (24 resp. 32 per record)
  const int occ64 = occupancy(64, ccap64, &lds64);
  const int occ128 = M.R <= 3 ? occupancy(128, 1536u, &lds128) : 0;
  if (M.R <= 3 && occ64 < 3 && occ128 >= 1) {
    f.tile = 128; f.ccap = 1536u; f.lds = lds128; f.blocks_per_cu = (unsigned)std::min(occ128, 2);
  } else {
    f.tile = 64; f.ccap = ccap64; f.lds = lds64; f.blocks_per_cu = (unsigned)std::max(1, std::min(occ64, VSR_OCC));
  }
  if (const char* e = std::getenv("VSRMC_MAX_BPC"))            // diagnostic: fewer resident blocks per CU (occupancy sweeps)
    f.blocks_per_cu = (unsigned)std::max(1, std::min<int>((int)f.blocks_per_cu, std::atoi(e)));
  return f;
}

// Put the checker in its initial state (ModelChecker.doInit): empty seen-set, Init in frontier 0 and in the set.
int checker_seed(vsrmc_checker* c) {
  const Model& M = c->model.M;
  c->saw_violation = false;
  c->deep = 0;
  c->deep_lv.clear();
  c->deep_g = 2;
  c->deep_distinct = c->deep_generated = 0;
  c->deep_regen_done = false;
  c->hist_new[0] = c->hist_new[1] = 0;
  c->g_last = 16;
  c->cur_rec_w = 0;
  c->failed = 0;
  c->failed_code = 0;
  HIPCHK(hipSetDevice(c->opt.device));
  hipLaunchKernelGGL(k_table_init, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1);
  HIPCHK(hipGetLastError());
  std::vector<u64> wire, dev(512);
  init_record_wire(M, wire);
  int len = wire_to_device(M, wire.data(), dev.data());
  u64 H[6];
  hash_full_host(M, (const u64*)dev.data(), H);   // pure arithmetic on the constant Init record (same code as the kernels)
  for (int i = 0; i < M.np; i++) dev[M.h0 + i] = H[i];
  u64 zero = (u64)len, init_fp = 0;                            // ref of record 0: offset 0, length len
  u32 init_ak = 0;
  canonical_fp(M, dev[0], &dev[M.h0], &init_fp, &init_ak);
  // sharded: every rank starts with Init (replicated phase: the small early levels are explored by every rank on its own,
  // vsrmc_shard_local_step); vsrmc_shard_partition then leaves each state with its owner
  const bool mine = true;
  if (c->filter) HIPCHK(hipMemsetAsync(c->filter, 0, (c->fmask + 1) * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->ctl, 0, sizeof(LevelCtl), c->stream));
  if (mine) {
    HIPCHK(hipMemcpyAsync(c->words[0], dev.data(), len * 8, hipMemcpyDefault, c->stream));   // the buffer may be pinned host memory
    HIPCHK(hipMemcpyAsync(c->off[0], &zero, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_seed, dim3(1), dim3(64), 0, c->stream, M, c->words[0], c->table, c->tmask, c->lvl_fp, c->ctl);
    HIPCHK(hipGetLastError());
  }
  HIPCHK(hipStreamSynchronize(c->stream));
  c->cur = 0;
  c->level = 1;
  c->n_frontier = mine ? 1 : 0;
  c->n_valid = c->n_frontier;
  c->cur_w = (u64)len;
  c->distinct = mine ? 1 : 0;
  c->total_generated = 0;
  c->failed = 0;
  c->cur_max_bag = 0;
  c->bag_known = true;
  c->failed_code = 0;
  c->probe_fp = 0;
  c->probe_level = 0;
  c->probe_extra_fp = 0;
  return 0;
}
}  // namespace

extern "C" {

void vsrmc_options_default(vsrmc_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->device = 0;
  o->table_log2 = 26;
  o->frontier_words = (uint64_t)1 << 27;
  o->frontier_states = (uint64_t)1 << 22;
  o->pending_entries = (uint64_t)1 << 23;
  o->keep_trace = 1;
  o->trace_entries = 0;
  o->rank = 0;
  o->world = 1;
}

// vsrmc_options with table_log2 == 0 and / or frontier_words == 0: sized from the free memory of the device.  Seen-set: the largest power
// of two of 16-byte slots within 30 % of what is free (1.7e9 states at load 0.4 on an empty MI355X); sharded runs on ONE device (tests)
// take their share.  Records: what is left after the seen-set, the index arrays (24 B per state index), the sent-filter and a reserve
// for the scratch buffers of the deep search (vsr_deep.hpp: 1/4 + 1/16 + .. of one record buffer) and the exchange buffers of a sharded
// run, in two equal buffers — the last two levels differ by the growth factor, but which of the two buffers holds the last one is not
// known in advance; pending list: only the exact scheme needs one worth the name.
static int autosize_options(vsrmc_options* o, const Model& M) {
  if (o->table_log2 != 0 && o->frontier_words != 0) return 0;
  size_t free_b = 0, total_b = 0;
  HIPCHK(hipSetDevice(o->device));
  HIPCHK(hipMemGetInfo(&free_b, &total_b));
  const char* share_env = std::getenv("VSRMC_AUTOSIZE_SHARE");     // several checkers on one device (tests: ranks sharing a GPU): 1 / share each
  const double share = share_env ? std::max(1.0, std::atof(share_env)) : 1.0;
  double avail = ((double)free_b - 3.0e9) / share;                 // runtime, code objects, small allocations
  if (avail < 256e6) return fail(VSRMC_E_HIP, "less than 256 MB of free device memory to size the checker from");
  if (o->table_log2 == 0) {
    int lg = 8;
    while (lg < 36 && (double)((u64)1 << (lg + 1)) * 16.0 <= 0.30 * avail) lg++;
    o->table_log2 = lg;
  }
  avail -= (double)((u64)1 << o->table_log2) * 16.0;
  if (o->world > 1 && !o->exact_ties) avail -= (double)((u64)1 << (o->filter_log2 > 0 ? o->filter_log2 : o->table_log2)) * 8.0;
  if (o->pending_entries == 0) o->pending_entries = o->exact_ties ? (u64)1 << 24 : (u64)1 << 16;
  avail -= (double)o->pending_entries * 24.0;
  if (o->frontier_words == 0) {
    if (o->world > 1) avail *= 0.80;                               // candidate / verdict / rebalancing buffers of the level loop
    // per record word: 8 B in each of two buffers, 1/24 state index (3 arrays of 8 B), a third of one buffer for the scratch buffers
    const double per_word = 2.0 * 8.0 + 8.0 / 3.0 + 24.0 / 24.0;
    const double words = avail / per_word;
    if (words < 4096.0 * (M.fixed + M.max_bag)) return fail(VSRMC_E_HIP, "not enough free device memory for the record buffers");
    o->frontier_words = (u64)words;
    o->frontier_words_b = 0;
    if (o->frontier_states == 0) o->frontier_states = std::max<u64>((u64)1 << 16, o->frontier_words / 24);
  }
  if (o->frontier_states == 0) o->frontier_states = std::max<u64>((u64)1 << 16, o->frontier_words / 24);
  return 0;
}

int32_t vsrmc_checker_create(const vsrmc_model* m, const vsrmc_options* o_in, vsrmc_checker** out) {
  if (!m || !o_in || !out) return fail(VSRMC_E_ARG, "NULL argument");
  vsrmc_options sized = *o_in;
  {
    const int rc0 = autosize_options(&sized, m->M);
    if (rc0) return rc0;
  }
  const vsrmc_options* o = &sized;
  if (o->table_log2 < 8 || o->table_log2 > 36 || o->frontier_states < 1 || o->frontier_words < 256 ||
      o->pending_entries < 4 * (uint64_t)VSR_CAND_CAP)
    return fail(VSRMC_E_ARG, "bad options");
  if (o->world < 1 || o->world > 8 || o->rank < 0 || o->rank >= o->world) return fail(VSRMC_E_ARG, "bad rank / world (1..8 ranks)");
  if (o->frontier_states > ((uint64_t)1 << 40)) return fail(VSRMC_E_ARG, "frontier_states > 2^40");   // origin word: parent index | ordinal << 40
  // a block reserves frontier indices in chunks of at least VSR_CAND_CAP (one tile's successors) and clears the unused tail of its
  // chunk: a frontier smaller than one chunk would be written past its end
  if (o->frontier_states < (uint64_t)VSR_CAND_CAP) return fail(VSRMC_E_ARG, "frontier_states must be at least 2048 (one index chunk)");
  int rc = check_device(o->device);
  if (rc) return rc;
  vsrmc_checker* c = new vsrmc_checker();
  c->model = *m;
  c->opt = *o;
  const Model& M = c->model.M;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, o->device));
  c->num_cus = prop.multiProcessorCount;
  c->lds_stride = (M.fixed + M.max_bag) | 1;
  HIPCHK(hipStreamCreate(&c->stream));
  for (int i = 0; i < 4; i++) HIPCHK(hipEventCreate(&c->ev[i]));
  u64 slots = (u64)1 << o->table_log2;
  c->tmask = slots - 1;
  hipError_t e = hipMalloc((void**)&c->table, slots * sizeof(Slot));
  c->host_frontier = o->host_frontier & 3;
  for (int b = 0; b < 2 && e == hipSuccess; b++) {
    // host_frontier: the records stay in pinned host memory and the kernels read / write them over PCIe (zero-copy); the
    // refs, fingerprints, trace log and the seen-set stay in HBM.  For state spaces whose frontier outgrows the 288 GB.
    if ((c->host_frontier >> b) & 1) e = hipHostMalloc((void**)&c->words[b], c->words_cap(b) * 8, hipHostMallocMapped | hipHostMallocPortable);
    else e = hipMalloc((void**)&c->words[b], c->words_cap(b) * 8);
    if (e == hipSuccess) e = hipMalloc((void**)&c->off[b], (o->frontier_states + 1) * 8);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&c->lvl_fp, o->frontier_states * 8);
  if (e == hipSuccess && o->world > 1 && !o->exact_ties) {
    const int fl = o->filter_log2 > 0 ? o->filter_log2 : o->table_log2;
    c->fmask = ((u64)1 << fl) - 1;
    e = hipMalloc((void**)&c->filter, (c->fmask + 1) * 8);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&c->pending, o->pending_entries * 24);   // (slot, key, parent index) entries of the exact scheme
  if (e == hipSuccess) e = hipMalloc((void**)&c->ctl, sizeof(LevelCtl));
  if (e == hipSuccess) e = hipMalloc((void**)&c->d_find, 8);
  if (e != hipSuccess) {
    vsrmc_checker_destroy(c);
    return fail(VSRMC_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e));
  }
  c->fused_kernel = (void*)fused_kernel_for(M);
  c->modes_kernel = (void*)modes_kernel_for(M);
  {
    // VSRMC_BLK=64 / 512 select the experimental block shapes of a -DVSRMC_EXPERIMENTAL_BLK=1 build (A/B runs); default 256
    const char* e = std::getenv("VSRMC_BLK");
    const int want = e ? std::atoi(e) : VSRMC_DEFAULT_BLK;
    c->plain_blk = ((want == 64 || want == 512) && plain_kernel_for(M, want)) ? want : VSR_BLOCK;
    c->plain_kernel = (void*)plain_kernel_for(M, c->plain_blk);
  }
  rc = checker_seed(c);
  if (rc) { vsrmc_checker_destroy(c); return rc; }
  *out = c;
  return 0;
}

int32_t vsrmc_checker_options(const vsrmc_checker* c, vsrmc_options* out) {
  if (!c || !out) return fail(VSRMC_E_ARG, "NULL argument");
  *out = c->opt;
  return 0;
}

int32_t vsrmc_checker_reset(vsrmc_checker* c) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  return checker_seed(c);
}

}  // extern "C"

namespace {

// ---- the phases of one BFS level (shared by the single-GPU step and the sharded protocol) --------------------------
int level_error(vsrmc_checker* c, const LevelCtl& h, int new_level) {
  c->failed = 1;
  c->failed_code = (int)h.err;
  char buf[256];
  std::snprintf(buf, sizeof(buf), "device error %u at frontier index %llu ordinal %llu (level %d)", h.err,
                (unsigned long long)(h.err_info >> 16), (unsigned long long)(h.err_info & 0xFFFF), new_level);
  std::string msg = buf;
  if (h.err == ERR_EVAL_421) msg = "VSR.tla:421: record has no field 'commit' (ReceivePrepareMsg, ClientCount >= 2); " + msg;
  return fail(h.err < ERR_REP_RANGE ? VSRMC_E_EVAL : VSRMC_E_REP, msg);
}

// phase 1: k_expand over the current frontier.  io == nullptr: unsharded.
int phase_expand(vsrmc_checker* c, const vsrmc_shard_io* io, int mode = MODE_NORMAL) {
  const Model& M = c->model.M;
  HIPCHK(hipSetDevice(c->opt.device));
  if (c->level + 1 >= 511) return fail(VSRMC_E_REP, "more than 510 BFS levels");
  c->t_level0 = now_s();
  c->expand_ms = c->materialize_ms = 0;
  std::memset(&c->h, 0, sizeof(c->h));
  c->h.viol_fp = ~(u64)0;
  HIPCHK(hipMemcpyAsync(c->ctl, &c->h, sizeof(c->h), hipMemcpyHostToDevice, c->stream));
  c->level_fused = !c->opt.exact_ties;
  if (c->n_frontier > 0) {
    // 128 records per tile when the work list has room for them (about 4 successors per record at R <= 3), else 64
    const bool fused = !c->opt.exact_ties;                       // sharded (io != nullptr) or not
    // sharded: records arrive from other ranks (rebalancing), the local maximum says nothing -> the format's capacity
    const bool use_plain = fused && !io && mode == MODE_NORMAL && c->plain_kernel;
    const FusedShape fs = fused_shape(c, c->bag_known ? c->cur_max_bag : (u64)M.max_bag, use_plain);
    const int cdiv = std::max(1, VSR_BLOCK / fs.blk);            // one-wave blocks: four times the blocks, a quarter of the chunk sizes
    const int tile = fused ? fs.tile : (M.R <= 3 ? 128 : 64);
    const int stride = fused ? fs.stride : c->lds_stride;
    u64 ntiles = (c->n_frontier + tile - 1) / tile;
    // every block reserves pending-list room in chunks: the list must hold one chunk per block beyond the real entries
    const u32 pchunk = c->opt.pending_entries >= ((u64)1 << 24) ? 8192u : (u32)VSR_CAND_CAP;
    if (io && c->cand_idx_cap < (u64)c->opt.world * io->cand_cap) {   // per announced candidate: where it was written (fused) / its parent (exact)
      if (c->cand_idx) (void)hipFree(c->cand_idx);
      c->cand_idx = nullptr;
      c->cand_idx_cap = 0;
      HIPCHK(hipMalloc((void**)&c->cand_idx, (u64)c->opt.world * io->cand_cap * 8));
      c->cand_idx_cap = (u64)c->opt.world * io->cand_cap;
    }
    unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)c->num_cus * 3);
    if (!fused) grid = (unsigned)std::min<u64>(grid, std::max<u64>(1, c->opt.pending_entries / (4 * (u64)pchunk)));   // the pending list is only used by the two-kernel scheme
    const u32 ccap = fused ? fs.ccap : (tile == 128 ? 1536u : (u32)VSR_CAND_CAP);   // 128-record tiles: two blocks per CU in LDS
    size_t lds = (size_t)tile * stride * 8 + 2 * (size_t)ccap * 4;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    // fused single-pass mode (unsharded, not exact_ties): the lane that inserts a fingerprint writes the successor itself
    const int nxt = c->cur ^ 1;
    const u64 nx_cap = c->opt.frontier_states;
    u32 ichunk = 0, wchunk = 0, cchunk = 0;
    if (fused && io)   // candidate entries a block reserves per owner at a time: <= 1/4 of a bucket in total over all blocks
      cchunk = (u32)std::max<u64>(16, std::min<u64>(512, io->cand_cap / (4 * (u64)c->num_cus * 2)));
    if (fused) {
      // persistent blocks (2 resident per CU: 225 VGPRs, 79 KB LDS): every block leaves one partly used index chunk and
      // one word chunk behind per level, so fewer blocks = fewer unused slots in the next frontier
      grid = (unsigned)std::min<u64>((u64)ntiles, (u64)c->num_cus * fs.blocks_per_cu);
      grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, std::min<u64>(nx_cap / (4 * (u64)VSR_CAND_CAP / cdiv), c->words_cap(nxt) / (4 * 16384))));
      // a tile's successors (at most ccap records of at most stride + 5 words each) must fit one word chunk, and every block
      // may leave one partly used chunk behind: fewer blocks if the buffer is too small for that
      // (a buffer too small even for one such chunk keeps going with what it has: the kernel refuses a tile that does not fit
      // its chunk with ERR_FRONTIER_FULL instead of writing past it)
      const u64 wmin = std::max<u64>(16384, (u64)ccap * (u64)(stride + 5));
      grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, c->words_cap(nxt) / (4 * wmin)));
      ichunk = (u32)std::max<u64>(VSR_CAND_CAP / cdiv, std::min<u64>(8192 / cdiv, nx_cap / (4 * (u64)grid)));
      wchunk = (u32)std::max<u64>(std::min<u64>(wmin, c->words_cap(nxt) / 2), std::min<u64>(262144 / cdiv, c->words_cap(nxt) / (4 * (u64)grid)));
    }
    if (fused)
      hipLaunchKernelGGL((ExpandKernel)(use_plain ? c->plain_kernel : c->fused_kernel), dim3(grid),
                         dim3(use_plain ? fs.blk : VSR_BLOCK), lds, c->stream, M, c->words[c->cur], c->off[c->cur],
                         c->n_frontier, c->level + 1, c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl,
                         stride, io ? c->opt.world : 1, io ? io->cand_send : nullptr, io ? io->cand_cap : 0, pchunk, c->words[nxt],
                         c->words_cap(nxt), c->off[nxt], nx_cap, c->lvl_fp, ichunk,
                         wchunk, tile, ccap, c->filter, c->fmask, c->cand_idx, cchunk,
                         mode | ((mode == MODE_PROBE && c->saw_violation) ? (int)MODE_NO_FOOTPRINT : 0), (u64)0);
    else
      hipLaunchKernelGGL(exact_kernel_for(M), dim3(grid), dim3(VSR_BLOCK), lds, c->stream, M, c->words[c->cur], c->off[c->cur],
                         c->n_frontier, c->level + 1, c->opt.rank, c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl,
                         c->lds_stride, io ? c->opt.world : 1, io ? io->cand_send : nullptr, io ? io->cand_cap : 0, pchunk, nullptr,
                         0, nullptr, 0, nullptr, 0, 0, tile, ccap, nullptr, 0, io ? c->cand_idx : nullptr, 0, 0, (u64)0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
  }
  HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (c->n_frontier > 0) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->expand_ms = ms;
  }
  c->nx_n = c->nx_w = 0;
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  if (!c->opt.exact_ties) {                                    // fused: the level is already materialised (sharded: speculatively)
    c->nx_n = c->h.n_new;
    c->nx_w = c->h.words_new;
    if (c->h.ties) {
      c->failed = 1;
      return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables "
                                 "(SURVEY F2); the single-pass scheme cannot arbitrate: create the checker with "
                                 "vsrmc_options.exact_ties = 1");
    }
  }
  return 0;
}

// phase 2: k_materialize over a list of (slot-or-fp, key) entries into one target (next frontier or a peer's bucket)
// (src_words / src_off: where the parents are read from — default: the current frontier; a slice of another buffer in the deep search)
int phase_materialize(vsrmc_checker* c, const u64* entries, u64 n, const uint8_t* verdict, u64* t_words, u64 t_words_cap,
                      u64* t_off, u64 t_cap, u64* t_fp, u64* cnt_n, u64* cnt_w, int entry_words, const u64* pidx_arr,
                      const u64* src_words = nullptr, const u64* src_off = nullptr) {
  if (n == 0) return 0;
  if (!src_words) { src_words = c->words[c->cur]; src_off = c->off[c->cur]; }
  const Model& M = c->model.M;
  // persistent waves: each keeps private output chunks, so the grid is sized to what is resident (LDS: 5 waves / CU)
  u64 grid64 = std::min<u64>((n + VSR_MAT_BLOCK - 1) / VSR_MAT_BLOCK, (u64)c->num_cus * 5);
  const u64 min_wchunk = (u64)VSR_MAT_BLOCK * c->lds_stride;
  grid64 = std::max<u64>(1, std::min<u64>(grid64, std::min<u64>(t_words_cap / (4 * min_wchunk), t_cap / (4 * 64))));
  const u32 ichunk = (u32)std::min<u64>(1024, std::max<u64>(64, t_cap / (4 * grid64)));
  const u32 wchunk = (u32)std::min<u64>(65536, std::max<u64>(min_wchunk, t_words_cap / (4 * grid64)));
  unsigned grid = (unsigned)grid64;
  size_t lds = (size_t)VSR_MAT_BLOCK * c->lds_stride * 8;
  HIPCHK(hipEventRecord(c->ev[2], c->stream));
  hipLaunchKernelGGL(materialize_kernel_for(M), dim3(grid), dim3(VSR_MAT_BLOCK), lds, c->stream, M, src_words, src_off, entries, n,
                     c->table, t_words, t_words_cap, t_off, t_cap, t_fp, c->ctl, verdict, cnt_n, cnt_w, c->lds_stride, ichunk, wchunk,
                     entry_words, pidx_arr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipEventRecord(c->ev[3], c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  float ms = 0;
  HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
  c->materialize_ms += ms;
  return 0;
}

// materialise the local pending list straight into the next frontier (self bucket)
int phase_materialize_local(vsrmc_checker* c) {
  u64 n_pending = std::min<u64>(c->h.n_pending, c->opt.pending_entries);
  const u64 nx_cap = c->opt.frontier_states;
  const int nxt = c->cur ^ 1;
  int rc = phase_materialize(c, c->pending, n_pending, nullptr, c->words[nxt], c->words_cap(nxt), c->off[nxt], nx_cap,
                             c->lvl_fp, &c->ctl->n_new, &c->ctl->words_new, 3, nullptr);
  if (rc) return rc;
  HIPCHK(hipMemcpy(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost));
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  c->nx_n = c->h.n_new;
  c->nx_w = c->h.words_new;
  return 0;
}

// phase 3: the level is complete: swap the frontiers, fill in the local statistics
int phase_commit(vsrmc_checker* c, vsrmc_level_info* info) {
  std::memset(info, 0, sizeof(*info));
  const LevelCtl& h = c->h;
  info->frontier = c->n_frontier;
  info->generated = h.generated;
  info->deadlocks = h.deadlocks;
  info->pending = h.n_pending;
  info->probes = h.probes;
  info->max_bag = h.max_bag;
  for (int a = 0; a < 16; a++) info->act_generated[a] = h.act_generated[a];
  for (int a = 0; a < 8; a++) info->phase_cycles[a] = h.phase_cycles[a];
  info->viol_fp = ~(u64)0;
  info->viol_index = ~(u64)0;
  info->expand_ms = c->expand_ms;
  info->materialize_ms = c->materialize_ms;
  // nx_n is an index RANGE: waves allocate indices in chunks and publish unused ones as invalid refs (0)
  u64 n_new = 0;
  if (c->nx_n > 0 && c->opt.world == 1 && !c->opt.exact_ties) {
    n_new = h.n_written;                                         // single-pass, unsharded: every record written is a new state
  } else if (c->nx_n > 0) {
    u64 zero = 0;
    HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_count_valid, dim3(1024), dim3(256), 0, c->stream, c->off[c->cur ^ 1], c->nx_n, c->d_find);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(&n_new, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  c->total_generated += h.generated;
  info->n_new = n_new;
  info->words_new = c->nx_w;
  info->record_words = h.rec_words;
  if (n_new > 0 || c->opt.world > 1) {   // sharded: levels stay aligned across ranks even when this shard got nothing
    c->cur ^= 1;
    c->level += 1;
    c->distinct += n_new;
    c->n_frontier = c->nx_n;
    c->n_valid = n_new;
    c->cur_w = c->nx_w;
    c->cur_max_bag = h.max_bag;
    c->bag_known = c->opt.world == 1;
    c->hist_new[0] = c->hist_new[1];
    c->hist_new[1] = n_new;
    c->g_last = (h.generated + std::max<u64>(1, info->frontier) - 1) / std::max<u64>(1, info->frontier) + 1;
    c->cur_rec_w = h.rec_words;
  } else {
    c->n_frontier = 0;
    c->n_valid = 0;
  }
  if (h.viol_fp != ~(u64)0) {
    info->viol_fp = h.viol_fp;
    info->viol_mask = (int32_t)h.viol_mask;
    c->saw_violation = true;
  }
  info->level = c->level;
  info->distinct = c->distinct;
  info->total_generated = c->total_generated;
  info->seconds = now_s() - c->t_level0;
  return 0;
}

int find_fp_newest(vsrmc_checker* c, u64 fp, u64* idx) {
  *idx = ~(u64)0;
  if (c->n_frontier == 0) return 0;
  u64 big = ~(u64)0;
  HIPCHK(hipMemcpy(c->d_find, &big, 8, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_find_fp, dim3((unsigned)((c->n_frontier + 255) / 256)), dim3(256), 0, c->stream, c->lvl_fp, c->n_frontier, fp,
                     c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipMemcpy(idx, c->d_find, 8, hipMemcpyDeviceToHost));
  return 0;
}

}  // namespace

extern "C" {

static int32_t step_local(vsrmc_checker* c, vsrmc_level_info* info) {
  int rc = phase_expand(c, nullptr);
  if (!rc && c->opt.exact_ties) rc = phase_materialize_local(c);
  if (rc) {   // like a TLC evaluation error: the run aborts, the partial level is not committed
    std::memset(info, 0, sizeof(*info));
    info->level = c->level;
    info->distinct = c->distinct;
    info->error_code = (int32_t)c->h.err;
    return rc;
  }
  rc = phase_commit(c, info);
  if (rc) return rc;
  if (info->viol_mask) return find_fp_newest(c, info->viol_fp, &info->viol_index);
  return 0;
}

namespace {
// One single-pass launch over an arbitrary source (a slice of the newest level, or the partial next frontier a MODE_REGEN
// slice just wrote), unsharded.  Resets the level counters, returns them in c->h.  Destination = the next-frontier buffers.
// io != nullptr: a pass of a sharded run (vsr_deep.hpp) — successors owned by other ranks are announced into io's buckets
int expand_pass(vsrmc_checker* c, const u64* src_words, const u64* src_off, u64 n_parents, u64 p_offset, int level, int mode,
                u64 src_max_bag, const PassDst* dst = nullptr, const vsrmc_shard_io* io = nullptr) {
  const Model& M = c->model.M;
  std::memset(&c->h, 0, sizeof(c->h));
  c->h.viol_fp = ~(u64)0;
  HIPCHK(hipMemcpyAsync(c->ctl, &c->h, sizeof(c->h), hipMemcpyHostToDevice, c->stream));
  if (n_parents > 0) {
    // an ordinary level into other buffers (the streamed level's sub-slices) runs the plain instantiation: the code of a stored level
    static const bool plain_normal = std::getenv("VSRMC_STREAM_MODES_KERNEL") == nullptr;
    const bool use_plain = !io && mode == MODE_NORMAL && plain_normal && c->plain_kernel && c->plain_blk == VSR_BLOCK;
    const FusedShape fs = fused_shape(c, src_max_bag, use_plain);
    u32 cchunk = 0;
    if (io) {
      if (c->cand_idx_cap < (u64)c->opt.world * io->cand_cap) {   // per announced candidate: where it was written / what regenerates it
        if (c->cand_idx) (void)hipFree(c->cand_idx);
        c->cand_idx = nullptr;
        c->cand_idx_cap = 0;
        HIPCHK(hipMalloc((void**)&c->cand_idx, (u64)c->opt.world * io->cand_cap * 8));
        c->cand_idx_cap = (u64)c->opt.world * io->cand_cap;
      }
      cchunk = (u32)std::max<u64>(16, std::min<u64>(512, io->cand_cap / (4 * (u64)c->num_cus * 2)));
    }
    const int tile = fs.tile;
    const u64 ntiles = (n_parents + tile - 1) / tile;
    const u32 ccap = fs.ccap;
    const size_t lds = fs.lds;
    const int nxt = c->cur ^ 1;
    u64* const d_words = dst ? dst->words : c->words[nxt];
    u64* const d_off = dst ? dst->off : c->off[nxt];
    u64* const d_fp = dst ? dst->fp : c->lvl_fp;
    const u64 d_wcap = dst ? dst->words_cap : c->words_cap(nxt);
    const u64 nx_cap = dst ? dst->cap : c->opt.frontier_states;
    unsigned grid = (unsigned)std::min<u64>(ntiles, (u64)c->num_cus * fs.blocks_per_cu);
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, std::min<u64>(nx_cap / (4 * (u64)VSR_CAND_CAP), d_wcap / (4 * 16384))));
    const u64 wmin = std::max<u64>(16384, (u64)ccap * (u64)(fs.stride + 5));   // see phase_expand
    grid = (unsigned)std::max<u64>(1, std::min<u64>(grid, d_wcap / (4 * wmin)));
    // index chunks: a block leaves the unused tail of its last chunk behind as invalid refs, and the NEXT pass stages those holes like records.
    // A whole level (2.6e8 states) loses 1-3 % to 8192-index chunks; a sub-slice of a streamed level (7e6 states from 1024 blocks) lost a third
    // of its index range, and the probe pass over it 16 % of its time (VSRMC_ICHUNK=8192: the old size, for A/B runs).
    static const u64 ichunk_small = std::getenv("VSRMC_ICHUNK") ? (u64)std::atoll(std::getenv("VSRMC_ICHUNK")) : 2048;
    const u64 ichunk_max = (dst || mode == MODE_REGEN) ? std::max<u64>(VSR_CAND_CAP, ichunk_small) : 8192;
    const u32 ichunk = (u32)std::max<u64>(VSR_CAND_CAP, std::min<u64>(ichunk_max, nx_cap / (4 * (u64)grid)));
    const u32 wchunk = (u32)std::max<u64>(std::min<u64>(wmin, d_wcap / 2), std::min<u64>(262144, d_wcap / (4 * (u64)grid)));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    const void* kern = io ? c->fused_kernel : use_plain ? c->plain_kernel : (c->modes_kernel ? c->modes_kernel : c->fused_kernel);
    hipLaunchKernelGGL((ExpandKernel)kern, dim3(grid), dim3(VSR_BLOCK), lds, c->stream, M, src_words, src_off, n_parents, level, c->opt.rank,
                       c->table, c->tmask, c->pending, c->opt.pending_entries, c->ctl, fs.stride, io ? c->opt.world : 1,
                       io ? io->cand_send : nullptr, io ? io->cand_cap : (u64)0, (u32)VSR_CAND_CAP,
                       d_words, d_wcap, d_off, nx_cap, d_fp,
                       ichunk, wchunk, tile, ccap, io ? c->filter : nullptr, io ? c->fmask : (u64)0, io ? c->cand_idx : nullptr, cchunk,
                       mode | ((mode == MODE_PROBE && c->saw_violation) ? (int)MODE_NO_FOOTPRINT : 0), p_offset);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
  }
  HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (n_parents > 0) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    c->expand_ms += ms;
  }
  if (c->h.err) return level_error(c, c->h, level);
  if (c->h.ties) {
    c->failed = 1;
    return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2)");
  }
  return 0;
}

// one step of a trace walk through the seen-set (k_table_lookup): by_low_bits = 0: the slot of fingerprint `key`; 1: the slot of
// the level-`level` state whose fingerprint ends in the 45 bits `key` (what a child's meta word knows of its parent)
int table_lookup(vsrmc_checker* c, u64 key, int level, int by_low_bits, int* found, u64* fp, u64* meta) {   // *found = matching states
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 24));
  hipLaunchKernelGGL(k_table_lookup, dim3(1), dim3(64), 0, c->stream, c->table, c->tmask, key, level, by_low_bits, d);
  u64 h[3] = {0, 0, 0};
  const bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
                  hipMemcpy(h, d, 24, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return fail(VSRMC_E_HIP, "k_table_lookup failed");
  *found = (int32_t)std::min<u64>(h[0], 0x7FFFFFFF);          // by_low_bits: the number of matching states (> 1: ambiguous)
  *fp = h[1];
  *meta = h[2];
  return 0;
}
// TLCTrace.getTrace, backwards half: the fingerprints of the path Init -> the level-`level` state with fingerprint `fp`
int walk_trace(vsrmc_checker* c, u64 fp, int level, std::vector<u64>* fps) {
  if (level < 1) return fail(VSRMC_E_ARG, "no such level");
  fps->assign((size_t)level + 1, 0);
  u64* d_fps = nullptr;
  HIPCHK(hipMalloc((void**)&d_fps, ((u64)level + 1) * 8));
  bool ok = hipMemsetAsync(d_fps, 0, ((u64)level + 1) * 8, c->stream) == hipSuccess;
  hipLaunchKernelGGL(k_trace_walk, dim3(1), dim3(64), 0, c->stream, c->table, c->tmask, fp, level, d_fps);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
       hipMemcpy(fps->data(), d_fps, ((u64)level + 1) * 8, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d_fps);
  if (!ok) return fail(VSRMC_E_HIP, "k_trace_walk failed");
  const u64 status = fps->back();
  fps->pop_back();
  if ((status & 0xFF) == 2) {
    char buf[256];
    std::snprintf(buf, sizeof(buf), "ambiguous predecessor pointer: %llu states of level %llu share the 45 fingerprint bits a successor keeps of its "
                  "parent (expected about once in 2^45 / level size steps); the counter-example cannot be walked through the seen-set",
                  (unsigned long long)(status >> 16), (unsigned long long)((status >> 8) & 0xFF));
    return fail(VSRMC_E_STATE, buf);
  }
  if (status != 0 || (*fps)[0] == 0) return fail(VSRMC_E_STATE, "the seen-set holds no path from Init to this state at this level");
  return 0;
}

// smallest-fingerprint violator of the (fp, key) list a PROBE / INSERT pass left in c->pending; among equal fps the smallest key
int min_violator(vsrmc_checker* c, u64 fp_min, u64* key) {
  *key = ~(u64)0;
  const u64 n = std::min<u64>(c->h.n_pending, std::min<u64>(c->opt.pending_entries, (u64)1 << 20));
  std::vector<u64> list(2 * n);
  if (n) HIPCHK(hipMemcpy(list.data(), c->pending, 16 * n, hipMemcpyDeviceToHost));
  for (u64 i = 0; i < n; i++)
    if (list[2 * i] == fp_min && list[2 * i + 1] < *key) *key = list[2 * i + 1];
  return 0;
}
}  // namespace

// ---- levels beyond the record buffers: virtual / regenerated / streamed / probed levels, vsrmc_checker_deepen, _probe2, _probe3
#include "vsr_deep.hpp"

// Probe level: expand the newest level WITHOUT storing its successors — every successor that is not a state of an earlier
// level gets its invariants checked, nothing is inserted into the seen-set, no frontier is written.  The search cannot
// continue afterwards (the level does not exist), but a violation one level beyond what memory can hold is found and its
// counter-example reconstructed (vsrmc_checker_probe_trace).  Also valid right after a step that failed with "frontier full".
int32_t vsrmc_checker_probe(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  // sharded: the rank probes its part of the newest level against ITS part of the seen-set; the violating successors it could
  // not find there (vsrmc_checker_probe_candidates) still have to be shown to their owners (sharded.py: ShardedChecker.probe)
  if (c->opt.exact_ties) return fail(VSRMC_E_STATE, "probe levels need a single-pass checker");
  if (c->failed && c->failed_code != ERR_FRONTIER_FULL) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  c->failed = 0;
  c->probe_fp = 0;
  c->probe_level = 0;
  c->probe_extra_fp = 0;
  int rc = phase_expand(c, nullptr, MODE_PROBE);
  if (rc) return rc;
  std::memset(info, 0, sizeof(*info));
  info->level = c->level + 1;
  info->frontier = c->n_frontier;
  info->generated = c->h.generated;
  info->deadlocks = c->h.deadlocks;
  info->probes = c->h.probes;
  info->pending = c->h.n_pending;                               // violating successors seen (duplicates included)
  info->distinct = c->distinct;
  info->total_generated = c->total_generated + c->h.generated;
  info->expand_ms = c->expand_ms;
  info->seconds = now_s() - c->t_level0;
  info->viol_fp = ~(u64)0;
  info->viol_index = ~(u64)0;
  for (int a = 0; a < 16; a++) info->act_generated[a] = c->h.act_generated[a];
  for (int a = 0; a < 8; a++) info->phase_cycles[a] = c->h.phase_cycles[a];
  if (c->h.viol_fp != ~(u64)0) {
    info->viol_fp = c->h.viol_fp;
    info->viol_mask = (int32_t)c->h.viol_mask;
    // the violator that is reported: smallest fingerprint; among its (fp, key) entries the smallest key
    u64 key = ~(u64)0;
    rc = min_violator(c, c->h.viol_fp, &key);
    if (rc) return rc;
    if (key != ~(u64)0 && c->opt.world <= 1) {                  // its parent: the newest level's state with these fingerprint bits
      int found = 0;
      u64 pfp = 0, pmeta = 0;
      rc = table_lookup(c, meta_pfp(key), c->level, 1, &found, &pfp, &pmeta);
      if (rc) return rc;
      if (found > 1) return fail(VSRMC_E_STATE, "ambiguous predecessor pointer: several states of the parent's level share the 45 fingerprint bits the violating successor keeps of its parent");
      if (found) {
        c->probe_fp = pfp;
        c->probe_level = c->level;
        c->probe_extra_fp = c->h.viol_fp;
      }
    }
  }
  return 0;
}

int32_t vsrmc_checker_probe_candidates(vsrmc_checker* c, uint64_t* pairs, uint64_t cap_pairs, uint64_t* n) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  *n = c->h.n_pending;
  if (c->h.n_pending > c->opt.pending_entries) return fail(VSRMC_E_REP, "more violating successors than the pending list holds (pending_entries)");
  if (c->h.n_pending == 0 || !pairs) return 0;                 // pairs == NULL: only the number is asked for
  if (cap_pairs < c->h.n_pending) return fail(VSRMC_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  HIPCHK(hipMemcpy(pairs, c->pending, 16 * c->h.n_pending, hipMemcpyDeviceToHost));
  return 0;
}

int32_t vsrmc_checker_seen_batch(vsrmc_checker* c, const uint64_t* fps, uint64_t n, int32_t level, uint8_t* seen) {
  if (!c || (n && (!fps || !seen))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 16 * n));
  std::vector<u64> flags(n, 0);
  hipError_t e = hipMemcpy(d, fps, 8 * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_table_seen, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->table, c->tmask, d, (u64)n, (int)level, d + n);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(flags.data(), d + n, 8 * n, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return fail(VSRMC_E_HIP, std::string("vsrmc_checker_seen_batch: ") + hipGetErrorString(e));
  for (u64 i = 0; i < n; i++) seen[i] = flags[i] ? 1 : 0;
  return 0;
}

int32_t vsrmc_checker_probe_trace(vsrmc_checker* c, uint64_t* words, uint64_t cap_words, uint64_t* off, int32_t* actions,
                                  uint64_t cap_states, uint64_t* n_states) {
  if (!c || !words || !off || !actions || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->probe_fp == 0) return fail(VSRMC_E_STATE, "no violation recorded by vsrmc_checker_probe");
  HIPCHK(hipSetDevice(c->opt.device));
  std::vector<u64> fps;
  int rc = walk_trace(c, c->probe_fp, c->probe_level, &fps);    // Init .. the deepest state of the path that is in the seen-set
  if (rc) return rc;
  if (c->probe_extra_fp) fps.push_back(c->probe_extra_fp);      // ... and the probed state beyond it
  return vsrmc_model_replay_fps(&c->model, c->opt.device, fps.data(), (int32_t)fps.size(), words, cap_words, off, actions, cap_states, n_states);
}

int32_t vsrmc_checker_step(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  if (c->deep) return fail(VSRMC_E_STATE, "levels beyond the record buffers exist in the seen-set (vsrmc_checker_deepen): the search goes on with vsrmc_checker_deepen / _advance");
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: drive the level with the vsrmc_shard_* phases");
  return step_local(c, info);
}

int32_t vsrmc_shard_local_step(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  return step_local(c, info);
}

// Sharded runs: the largest bag among the records of the newest level over ALL ranks (records move between ranks when the
// frontiers are rebalanced, so a rank's own maximum is not enough).  Lets the next k_expand size its LDS record slots for the
// level instead of the format's worst case; without this call the worst case is used.
int32_t vsrmc_shard_set_max_bag(vsrmc_checker* c, uint64_t max_bag) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  c->cur_max_bag = max_bag;
  c->bag_known = true;
  return 0;
}

int32_t vsrmc_shard_partition(vsrmc_checker* c, uint64_t* n_kept) {
  if (!c || !n_kept) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  *n_kept = c->n_valid;
  if (c->opt.world <= 1 || c->n_frontier == 0) return 0;
  u64 zero = 0;
  HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_partition, dim3((unsigned)((c->n_frontier + 255) / 256)), dim3(256), 0, c->stream, c->off[c->cur], c->lvl_fp,
                     c->n_frontier, c->opt.rank, c->opt.world, c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(n_kept, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->n_valid = *n_kept;
  return 0;
}

// Will the next level fit the idle record buffer?  Level sizes of these models grow by a factor that FALLS from level to level once the
// search is past its first few levels (tests/golden/oracle_levels_*.json: r(l+1) / r(l) is 0.92 .. 0.99 everywhere beyond level 10), so
// the last level's factor bounds the next one's; records grow by at most one bag entry per level.  Small levels are bounded by the
// successors generated per state instead.  A wrong "yes" ends in ERR_FRONTIER_FULL, a wrong "no" only costs a level of re-expansion.
static bool next_level_fits(const vsrmc_checker* c) {
  const u64 n = c->n_valid;
  if (n == 0) return true;
  double pred;
  if (n < 32768 || c->hist_new[0] == 0) pred = (double)n * (double)std::max<u64>(2, c->g_last) * 1.25;
  else pred = (double)n * std::min((double)c->g_last, (double)c->hist_new[1] / (double)c->hist_new[0] * 1.02);
  const double wbar = (double)std::max<u64>(c->cur_rec_w, (u64)c->model.M.fixed * n) / (double)n + 1.0;
  const int nxt = c->cur ^ 1;
  const double blocks = 4.0 * c->num_cus;                        // every resident block leaves a partly used word and index chunk behind
  const double cap_w = (double)c->words_cap(nxt), cap_n = (double)c->opt.frontier_states;
  return pred * wbar + std::min(blocks * 262144.0, cap_w / 4) <= cap_w && pred * 1.09 + std::min(blocks * 8192.0, cap_n / 4) <= cap_n;
}

// One unit of progress of the automatic level scheme (no level numbers, no sizes from the caller): an ordinary BFS level while the
// next one is predicted to fit the record buffers (*what = 1: a = that level), otherwise one pass of the deep search — the next level
// inserted into the seen-set only, the one after it probed (*what = 2: a = the inserted level, b = the probed one, b->level == 0 when the
// pass probed nothing).  a->n_new == 0: the search is exhausted.
int32_t vsrmc_checker_advance(vsrmc_checker* c, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what) {
  if (!c || !a || !b || !what) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(b, 0, sizeof(*b));
  b->viol_fp = b->viol_index = ~(u64)0;
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: vsrmc_shard_loop_advance");
  if (!c->deep && (c->opt.exact_ties || next_level_fits(c))) {
    *what = 1;
    return vsrmc_checker_step(c, a);
  }
  *what = 2;
  return vsrmc_checker_deepen(c, a, b);
}

// ≙ ModelChecker.run: stop_reason 0 = exhausted, 1 = invariant violated (*last = the level it was found in; a probed level: see
// vsrmc_checker_probe_trace), 2 = max_depth, 3 = max_seconds, 4 = the seen-set is 85 % full (the search is incomplete: depth reached
// = last->level)
int32_t vsrmc_check(vsrmc_checker* c, int32_t max_depth, double max_seconds, int32_t* stop_reason, vsrmc_level_info* last) {
  if (!c || !stop_reason || !last) return fail(VSRMC_E_ARG, "NULL argument");
  const double t0 = now_s();
  std::memset(last, 0, sizeof(*last));
  last->level = c->level;
  last->distinct = c->distinct;
  vsrmc_level_info a, b;
  while (true) {
    if (max_depth > 0 && c->level + c->deep >= max_depth) { *stop_reason = 2; return 0; }
    if (max_seconds > 0 && now_s() - t0 > max_seconds) { *stop_reason = 3; return 0; }
    if ((double)(c->deep ? c->deep_distinct : c->distinct) > 0.85 * (double)(c->tmask + 1)) { *stop_reason = 4; return 0; }
    int32_t what = 0;
    int rc = vsrmc_checker_advance(c, &a, &b, &what);
    if (rc) return rc;
    *last = a;
    if (a.viol_mask) { *stop_reason = 1; return 0; }
    if (a.n_new == 0) { *stop_reason = 0; return 0; }
    if (what == 2 && b.level && b.viol_mask) { *last = b; *stop_reason = 1; return 0; }
  }
}

// ---- sharded protocol: one level = expand -> [exchange] -> claim -> [exchange] -> materialize -> [exchange] -> append -> commit
int32_t vsrmc_shard_expand(vsrmc_checker* c, const vsrmc_shard_io* io, uint64_t* cand_counts) {
  if (!c || !io || !cand_counts) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  int rc = phase_expand(c, io);
  for (int o = 0; o < c->opt.world; o++) cand_counts[o] = c->h.cand_cnt[o];
  return rc;
}

int32_t vsrmc_shard_claim(vsrmc_checker* c, const uint64_t* d_cand_recv, uint64_t n, uint8_t* d_verdict) {
  if (!c || (n && (!d_cand_recv || !d_verdict))) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  if (c->opt.exact_ties && n > c->rslot_cap) {
    if (c->rslot) (void)hipFree(c->rslot);
    c->rslot = nullptr;
    c->rslot_cap = 0;
    HIPCHK(hipMalloc((void**)&c->rslot, n * 8));
    c->rslot_cap = n;
  }
  unsigned grid = (unsigned)((n + 255) / 256);
  if (!c->opt.exact_ties) {   // single-pass level: the inserting candidate wins, the verdict is known at once
    hipLaunchKernelGGL(k_claim_batch_fused, dim3(grid), dim3(256), 0, c->stream, c->table, c->tmask, d_cand_recv, n, c->level + 1, d_verdict,
                       c->ctl);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
  }
  hipLaunchKernelGGL(k_claim_batch, dim3(grid), dim3(256), 0, c->stream, c->table, c->tmask, d_cand_recv, n, c->level + 1, c->rslot, c->ctl);
  hipLaunchKernelGGL(k_verdict, dim3(grid), dim3(256), 0, c->stream, c->table, d_cand_recv, c->rslot, n, d_verdict);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

// NOTE: every rank must have finished vsrmc_shard_claim for ALL its received candidates before any verdict is used:
// the verdict of a slot is final only when every claim of the level has landed (the orchestrator's exchange is the barrier).
int32_t vsrmc_shard_materialize(vsrmc_checker* c, const vsrmc_shard_io* io, const uint8_t* d_verdict_in) {
  if (!c || !io) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  // every winner — local owner or remote verdict — is written into THIS rank's next frontier: records stay with their
  // generator, only 16-byte candidates and verdict bytes cross ranks (rebalancing moves records in bulk when needed)
  if (c->level_fused) {
    // single-pass level: k_expand wrote the announced successors speculatively; withdraw the ones whose owner said no
    HIPCHK(hipSetDevice(c->opt.device));
    const int nxt = c->cur ^ 1;
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    for (int o = 0; o < c->opt.world; o++) {
      if (o == c->opt.rank) continue;
      const u64 n = std::min<u64>(c->h.cand_cnt[o], io->cand_cap);
      if (n == 0) continue;
      if (!d_verdict_in) return fail(VSRMC_E_ARG, "verdicts missing");
      hipLaunchKernelGGL(k_apply_verdict, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, io->cand_send + 2 * (u64)o * io->cand_cap,
                         c->cand_idx + (u64)o * io->cand_cap, d_verdict_in + (u64)o * io->cand_cap, n, c->off[nxt], c->lvl_fp, c->ctl);
      HIPCHK(hipGetLastError());
    }
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    HIPCHK(hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[2], c->ev[3]));
    c->materialize_ms += ms;
    if (c->h.err) return level_error(c, c->h, c->level + 1);
    if (c->h.ties) {
      c->failed = 1;
      return fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2): "
                                 "create the checker with vsrmc_options.exact_ties = 1");
    }
    c->nx_n = c->h.n_new;
    c->nx_w = c->h.words_new;
    return 0;
  }
  int rc = phase_materialize_local(c);
  const int nxt = c->cur ^ 1;
  const u64 nx_cap = c->opt.frontier_states;
  for (int o = 0; o < c->opt.world && !rc; o++) {
    if (o == c->opt.rank) continue;
    u64 n = std::min<u64>(c->h.cand_cnt[o], io->cand_cap);
    if (n && !d_verdict_in) return fail(VSRMC_E_ARG, "verdicts missing");
    rc = phase_materialize(c, io->cand_send + 2 * (u64)o * io->cand_cap, n, d_verdict_in + (u64)o * io->cand_cap, c->words[nxt],
                           c->words_cap(nxt), c->off[nxt], nx_cap, c->lvl_fp, &c->ctl->n_new, &c->ctl->words_new, 2,
                           c->cand_idx + (u64)o * io->cand_cap);
  }
  if (rc) return rc;
  HIPCHK(hipMemcpy(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost));
  if (c->h.err) return level_error(c, c->h, c->level + 1);
  c->nx_n = c->h.n_new;
  c->nx_w = c->h.words_new;
  return 0;
}

int32_t vsrmc_shard_count(vsrmc_checker* c, uint64_t* n_valid, uint64_t* n_range) {
  if (!c || !n_valid || !n_range) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  *n_range = c->nx_n;
  *n_valid = 0;
  if (c->nx_n == 0) return 0;
  u64 zero = 0;
  HIPCHK(hipMemcpyAsync(c->d_find, &zero, 8, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_count_valid, dim3(1024), dim3(256), 0, c->stream, c->off[c->cur ^ 1], c->nx_n, c->d_find);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(n_valid, c->d_find, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return 0;
}

int32_t vsrmc_shard_export(vsrmc_checker* c, uint64_t first, uint64_t n, uint64_t* d_words, uint64_t words_cap, uint64_t* d_off,
                           uint64_t* d_fp, uint64_t cap, uint64_t* n_out, uint64_t* words_out) {
  if (!c || !n_out || !words_out) return fail(VSRMC_E_ARG, "NULL argument");
  *n_out = *words_out = 0;
  if (n == 0) return 0;
  if (!d_words || !d_off || !d_fp || first + n > c->nx_n) return fail(VSRMC_E_ARG, "bad export window");
  HIPCHK(hipSetDevice(c->opt.device));
  const int nxt = c->cur ^ 1;
  u64* d_cnt = nullptr;
  HIPCHK(hipMalloc((void**)&d_cnt, 32));
  HIPCHK(hipMemsetAsync(d_cnt, 0, 32, c->stream));
  hipLaunchKernelGGL(k_export, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream, c->words[nxt], c->off[nxt] + 0, c->lvl_fp,
                     first, n, d_words, words_cap, d_off, d_fp, cap, d_cnt, (u32*)(d_cnt + 2));
  HIPCHK(hipGetLastError());
  u64 h[4];
  HIPCHK(hipMemcpyAsync(h, d_cnt, 32, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  (void)hipFree(d_cnt);
  if ((u32)h[2]) return fail(VSRMC_E_REP, "export buffers too small");
  *n_out = h[0];
  *words_out = h[1];
  return 0;
}

int32_t vsrmc_shard_append(vsrmc_checker* c, const uint64_t* d_words, uint64_t n_words, const uint64_t* d_off,
                           const uint64_t* d_fp, uint64_t n) {
  if (!c) return fail(VSRMC_E_ARG, "NULL argument");
  if (n == 0) return 0;
  if (!d_words || !d_off || !d_fp) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  const u64 nx_cap = c->opt.frontier_states;
  if (c->nx_n + n > nx_cap || c->nx_w + n_words > c->words_cap(c->cur ^ 1)) {
    c->failed = 1;
    return fail(VSRMC_E_REP, "frontier buffers full while appending received records");
  }
  const int nxt = c->cur ^ 1;
  HIPCHK(hipMemcpyAsync(c->words[nxt] + c->nx_w, d_words, n_words * 8, hipMemcpyDefault, c->stream));
  hipLaunchKernelGGL(k_append_fixup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->off[nxt] + c->nx_n,
                     c->lvl_fp + c->nx_n, d_off, d_fp, n, c->nx_w);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  c->nx_n += n;
  c->nx_w += n_words;
  return 0;
}

int32_t vsrmc_shard_commit(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  return phase_commit(c, info);
}

// ---- checkpoint / recover ≙ TLC's checkpoints (ModelChecker.checkpoint: FPSet.beginChkpt/commitChkpt, StateQueue, TLCTrace) ----
namespace {
struct ChkHeader {
  char magic[8];                 // "VSRMCCK3" (1 = the format with a separate trace log and index-based meta words; 2 = without the module in the header)
  int32_t consts[12];            // R, C, n, L, symmetry, inv_mask, assume_commit, np, module (model_id), words per replica, fixed words,
                                 // version of the fingerprint function: records and fingerprints mean nothing under another layout or hash
  int32_t level, shard;          // shard: 0 = unsharded, else world << 16 | rank (each rank writes and reads its own file)
  u64 n_frontier, n_valid, cur_w, distinct, total_generated, n_levels, table_entries, trace_entries;
};
bool dev_to_file(FILE* f, const void* d_ptr, u64 bytes, std::vector<char>& buf) {
  for (u64 pos = 0; pos < bytes; pos += buf.size()) {
    const u64 k = std::min<u64>(buf.size(), bytes - pos);
    if (hipMemcpy(buf.data(), (const char*)d_ptr + pos, k, hipMemcpyDefault) != hipSuccess) return false;
    if (std::fwrite(buf.data(), 1, k, f) != k) return false;
  }
  return true;
}
bool file_to_dev(FILE* f, void* d_ptr, u64 bytes, std::vector<char>& buf) {
  for (u64 pos = 0; pos < bytes; pos += buf.size()) {
    const u64 k = std::min<u64>(buf.size(), bytes - pos);
    if (std::fread(buf.data(), 1, k, f) != k) return false;
    if (hipMemcpy((char*)d_ptr + pos, buf.data(), k, hipMemcpyDefault) != hipSuccess) return false;
  }
  return true;
}
}  // namespace

int32_t vsrmc_checker_save(vsrmc_checker* c, const char* path) {
  if (!c || !path) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->failed) return fail(VSRMC_E_STATE, "the checker stopped on an error");
  if (c->opt.world > 1 && c->opt.exact_ties) return fail(VSRMC_E_STATE, "checkpoints of sharded exact-mode checkers are not supported");
  if (c->deep) return fail(VSRMC_E_STATE, "the seen-set holds levels beyond the newest materialised one (vsrmc_checker_deepen): no checkpoint can describe that state");
  HIPCHK(hipSetDevice(c->opt.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const Model& M = c->model.M;
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(VSRMC_E_CFG, "cannot write " + tmp);
  ChkHeader h;
  std::memset(&h, 0, sizeof(h));
  std::memcpy(h.magic, "VSRMCCK3", 8);
  const int32_t consts[12] = {M.R, M.C, M.n, M.L, c->model.symmetry, M.inv_mask, M.assume_commit, M.np, M.model_id, M.wpr, M.fixed, fp_function_id(M)};
  std::memcpy(h.consts, consts, sizeof(consts));
  h.level = c->level;
  h.shard = c->opt.world > 1 ? (c->opt.world << 16 | c->opt.rank) : 0;
  h.n_frontier = c->n_frontier;
  h.n_valid = c->n_valid;
  h.cur_w = c->cur_w;
  h.distinct = c->distinct;
  h.total_generated = c->total_generated;
  h.n_levels = (u64)c->level;
  h.trace_entries = 0;                                          // the predecessor pointers travel inside the seen-set slots
  std::vector<char> buf((size_t)64 << 20);
  bool ok = true;
  // the seen-set: occupied slots only, window by window (the export buffer holds one window)
  const u64 slots = c->tmask + 1, win = std::min<u64>(slots, (u64)1 << 26);
  Slot* d_out = nullptr;
  u64* d_cnt = nullptr;
  if (hipMalloc((void**)&d_out, win * sizeof(Slot)) != hipSuccess || hipMalloc((void**)&d_cnt, 8) != hipSuccess) {
    if (d_out) (void)hipFree(d_out);
    std::fclose(f);                                             // no early return leaves the file open or the .tmp behind
    std::remove(tmp.c_str());
    return fail(VSRMC_E_HIP, "hipMalloc of the checkpoint export window failed");
  }
  ok = std::fwrite(&h, sizeof(h), 1, f) == 1;                   // rewritten at the end with table_entries
  u64 total = 0;
  for (u64 first = 0; first < slots && ok; first += win) {
    u64 cnt = 0;
    ok = hipMemset(d_cnt, 0, 8) == hipSuccess;
    hipLaunchKernelGGL(k_table_export, dim3((unsigned)((win + 255) / 256)), dim3(256), 0, c->stream, c->table, first, win, d_out, win, d_cnt);
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(&cnt, d_cnt, 8, hipMemcpyDeviceToHost) == hipSuccess;
    ok = ok && dev_to_file(f, d_out, cnt * sizeof(Slot), buf);
    total += cnt;
  }
  (void)hipFree(d_out);
  (void)hipFree(d_cnt);
  ok = ok && dev_to_file(f, c->words[c->cur], c->cur_w * 8, buf);
  ok = ok && dev_to_file(f, c->off[c->cur], c->n_frontier * 8, buf);
  ok = ok && dev_to_file(f, c->lvl_fp, c->n_frontier * 8, buf);
  h.table_entries = total;
  ok = ok && std::fseek(f, 0, SEEK_SET) == 0 && std::fwrite(&h, sizeof(h), 1, f) == 1;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) {
    std::remove(tmp.c_str());
    return fail(VSRMC_E_CFG, std::string("writing the checkpoint ") + path + " failed");
  }
  return 0;
}

int32_t vsrmc_checker_load(const vsrmc_model* m, const vsrmc_options* o, const char* path, vsrmc_checker** out) {
  if (!m || !o || !path || !out) return fail(VSRMC_E_ARG, "NULL argument");
  if (o->world > 1 && o->exact_ties) return fail(VSRMC_E_STATE, "checkpoints of sharded exact-mode checkers are not supported");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(VSRMC_E_CFG, std::string("cannot read ") + path);
  ChkHeader h;
  bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "VSRMCCK3", 8) == 0;
  // header invariants (a truncated or foreign file must not become an inconsistent checker)
  if (ok) ok = h.level >= 1 && h.level < 511 && (u64)h.level == h.n_levels && h.n_valid <= h.n_frontier && h.trace_entries == 0;
  if (!ok) {
    std::fclose(f);
    return fail(VSRMC_E_CFG, std::string(path) + " is not a (consistent) vsrmc checkpoint");
  }
  const Model& M = m->M;
  const int32_t consts[12] = {M.R, M.C, M.n, M.L, m->symmetry, M.inv_mask, M.assume_commit, M.np, M.model_id, M.wpr, M.fixed, fp_function_id(M)};
  if (std::memcmp(h.consts, consts, sizeof(consts)) != 0) {
    std::fclose(f);
    return fail(VSRMC_E_CFG, "the checkpoint was written for another module, other model constants or another fingerprint function");
  }
  if (h.shard != (o->world > 1 ? (o->world << 16 | o->rank) : 0)) {   // the seen-set is partitioned by owner_of(fp, world)
    std::fclose(f);
    return fail(VSRMC_E_CFG, "the checkpoint was written by another rank or for another world size");
  }
  const int buf_of_level = (h.level - 1) & 1;                   // level L lives in record buffer (L - 1) mod 2, also after recovery
  const u64 cap_of_buf = (buf_of_level == 1 && o->frontier_words_b) ? o->frontier_words_b : o->frontier_words;
  if (h.n_frontier > o->frontier_states || h.cur_w > cap_of_buf || 2 * h.table_entries > ((u64)1 << o->table_log2)) {
    std::fclose(f);
    return fail(VSRMC_E_ARG, "the options are too small for this checkpoint (frontier, table)");
  }
  vsrmc_checker* c = nullptr;
  int rc = vsrmc_checker_create(m, o, &c);
  if (rc) {
    std::fclose(f);
    return rc;
  }
  std::vector<char> buf((size_t)64 << 20);
  // the seen-set: empty it (create seeded Init), re-insert the saved slots
  hipLaunchKernelGGL(k_table_init, dim3(4096), dim3(256), 0, c->stream, c->table, c->tmask + 1);
  const u64 win = (u64)1 << 24;
  Slot* d_in = nullptr;
  u32* d_err = nullptr;
  ok = hipMalloc((void**)&d_in, win * sizeof(Slot)) == hipSuccess && hipMalloc((void**)&d_err, 4) == hipSuccess &&
       hipMemset(d_err, 0, 4) == hipSuccess;
  for (u64 done = 0; done < h.table_entries && ok; done += win) {
    const u64 k = std::min<u64>(win, h.table_entries - done);
    ok = file_to_dev(f, d_in, k * sizeof(Slot), buf);
    hipLaunchKernelGGL(k_table_import, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, c->table, c->tmask, d_in, k, d_err);
    ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
  }
  u32 terr = 0;
  if (ok) ok = hipMemcpy(&terr, d_err, 4, hipMemcpyDeviceToHost) == hipSuccess && terr == 0;
  if (d_in) (void)hipFree(d_in);
  if (d_err) (void)hipFree(d_err);
  c->cur = buf_of_level;
  ok = ok && file_to_dev(f, c->words[c->cur], h.cur_w * 8, buf);
  ok = ok && file_to_dev(f, c->off[c->cur], h.n_frontier * 8, buf);
  ok = ok && file_to_dev(f, c->lvl_fp, h.n_frontier * 8, buf);
  std::fclose(f);
  if (!ok) {
    vsrmc_checker_destroy(c);
    return fail(VSRMC_E_CFG, std::string("reading the checkpoint ") + path + " failed");
  }
  c->bag_known = false;                                        // the header does not carry it: LDS slots at the format's capacity
  c->level = h.level;
  c->n_frontier = h.n_frontier;
  c->n_valid = h.n_valid;
  c->cur_w = h.cur_w;
  c->distinct = h.distinct;
  c->total_generated = h.total_generated;
  *out = c;
  return 0;
}

int32_t vsrmc_checker_status(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(info, 0, sizeof(*info));
  info->level = c->level;
  info->n_new = c->n_valid;
  info->distinct = c->distinct;
  info->total_generated = c->total_generated;
  info->words_new = c->cur_w;
  info->viol_fp = ~(u64)0;
  info->viol_index = ~(u64)0;
  return 0;
}

int32_t vsrmc_checker_find_fp(vsrmc_checker* c, uint64_t fp, uint64_t* index) {
  if (!c || !index) return fail(VSRMC_E_ARG, "NULL argument");
  return find_fp_newest(c, fp, index);
}

int32_t vsrmc_checker_lookup(vsrmc_checker* c, uint64_t key, int32_t level, int32_t by_low_bits, int32_t* found, uint64_t* fp,
                             uint64_t* meta) {
  if (!c || !found || !fp || !meta) return fail(VSRMC_E_ARG, "NULL argument");
  HIPCHK(hipSetDevice(c->opt.device));
  int f = 0;
  int rc = table_lookup(c, key, level, by_low_bits, &f, fp, meta);
  *found = f;
  return rc;
}

int32_t vsrmc_checker_level_fps(vsrmc_checker* c, uint64_t* out, uint64_t cap, uint64_t* n) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  *n = c->n_valid;
  if (!out || cap < c->n_valid) return fail(VSRMC_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  std::vector<u64> all(c->n_frontier);
  HIPCHK(hipMemcpy(all.data(), c->lvl_fp, c->n_frontier * 8, hipMemcpyDeviceToHost));
  u64 k = 0;
  for (u64 v : all)
    if (v != 0 && k < cap) out[k++] = v;        // 0 = unused index of a wave's chunk
  *n = k;
  std::sort(out, out + k);
  return 0;
}

int32_t vsrmc_checker_level_checksum(vsrmc_checker* c, uint64_t* fp_xor, uint64_t* fp_sum, uint64_t* n_states) {
  if (!c || !fp_xor || !fp_sum || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  *fp_xor = *fp_sum = *n_states = 0;
  if (c->n_frontier == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  u64* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, 24));
  u64 h[3] = {0, 0, 0};
  bool ok = hipMemsetAsync(d, 0, 24, c->stream) == hipSuccess;
  hipLaunchKernelGGL(k_level_checksum, dim3(2048), dim3(256), 0, c->stream, c->lvl_fp, c->n_frontier, d);
  ok = ok && hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
       hipMemcpy(h, d, 24, hipMemcpyDeviceToHost) == hipSuccess;
  (void)hipFree(d);
  if (!ok) return fail(VSRMC_E_HIP, "k_level_checksum failed");
  *fp_xor = h[0];
  *fp_sum = h[1];
  *n_states = h[2];
  return 0;
}

int32_t vsrmc_checker_frontier(vsrmc_checker* c, uint64_t* words, uint64_t cap_words, uint64_t* off, uint64_t cap_states,
                               uint64_t* n) {
  if (!c || !n || !words || !off) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = c->model.M;
  *n = c->n_valid;
  if (cap_states < c->n_valid + 1) return fail(VSRMC_E_ARG, "offset buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  std::vector<u64> doff(c->n_frontier);
  HIPCHK(hipMemcpy(doff.data(), c->off[c->cur], c->n_frontier * 8, hipMemcpyDeviceToHost));
  u64 hi = 0;
  std::vector<char> valid(c->n_frontier);
  for (u64 i = 0; i < c->n_frontier; i++) {                    // refs are (word offset << 8 | length); 0 = unused index
    valid[i] = doff[i] != 0;
    doff[i] >>= 8;
    hi = std::max(hi, doff[i]);
  }
  std::vector<u64> dev(hi + (u64)M.fixed + 256);
  u64 take = std::min<u64>(dev.size(), c->words_cap(c->cur));
  HIPCHK(hipMemcpy(dev.data(), c->words[c->cur], take * 8, hipMemcpyDefault));
  u64 pos = 0, k = 0;
  for (u64 i = 0; i < c->n_frontier; i++) {
    if (!valid[i]) continue;
    const u64* r = &dev[doff[i]];
    u64 wl = (u64)M.h0 + hdr_nmsg(r[0]);
    if (pos + wl > cap_words || k >= cap_states) return fail(VSRMC_E_ARG, "buffers too small");
    off[k++] = pos;
    device_to_wire(M, r, words + pos);
    pos += wl;
  }
  off[k] = pos;
  *n = k;
  return 0;
}

// The states of the newest level in which an action of `action_mask` is enabled (wire layout), at most max_states of them;
// *n_matching = how many there are in all.  ≙ TLC's action coverage, used as a filter (directed parity tests, debugging).
int32_t vsrmc_checker_select(vsrmc_checker* c, uint32_t action_mask, uint64_t max_states, uint64_t* words, uint64_t cap_words,
                             uint64_t* off, uint64_t* n_states, uint64_t* n_matching) {
  if (!c || !words || !off || !n_states || !n_matching) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = c->model.M;
  *n_states = *n_matching = 0;
  off[0] = 0;
  if (c->n_frontier == 0 || max_states == 0) return 0;
  HIPCHK(hipSetDevice(c->opt.device));
  u64 *d_idx = nullptr, *d_cnt = nullptr;
  HIPCHK(hipMalloc((void**)&d_idx, max_states * 8));
  if (hipMalloc((void**)&d_cnt, 8) != hipSuccess || hipMemset(d_cnt, 0, 8) != hipSuccess) {
    (void)hipFree(d_idx);
    return fail(VSRMC_E_HIP, "hipMalloc failed");
  }
  hipLaunchKernelGGL((M.model_id == 1 ? k_select<1> : M.model_id == 2 ? k_select<2> : k_select<0>), dim3((unsigned)((c->n_frontier + 255) / 256)), dim3(256), 0, c->stream, M, c->words[c->cur], c->off[c->cur],
                     c->n_frontier, action_mask, d_idx, max_states, d_cnt);
  u64 cnt = 0;
  bool ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess &&
            hipMemcpy(&cnt, d_cnt, 8, hipMemcpyDeviceToHost) == hipSuccess;
  const u64 k = std::min<u64>(cnt, max_states);
  std::vector<u64> idx(k);
  ok = ok && (k == 0 || hipMemcpy(idx.data(), d_idx, k * 8, hipMemcpyDeviceToHost) == hipSuccess);
  (void)hipFree(d_idx);
  (void)hipFree(d_cnt);
  if (!ok) return fail(VSRMC_E_HIP, "k_select failed");
  std::sort(idx.begin(), idx.end());
  std::vector<u64> rec(256);
  u64 pos = 0;
  for (u64 q = 0; q < k; q++) {
    u64 ref = 0;
    HIPCHK(hipMemcpy(&ref, c->off[c->cur] + idx[q], 8, hipMemcpyDeviceToHost));
    const u64 len = ref & 255;
    HIPCHK(hipMemcpy(rec.data(), c->words[c->cur] + (ref >> 8), len * 8, hipMemcpyDefault));
    const u64 wl = (u64)M.h0 + hdr_nmsg(rec[0]);
    if (pos + wl > cap_words) return fail(VSRMC_E_ARG, "buffers too small");
    device_to_wire(M, rec.data(), words + pos);
    pos += wl;
    off[q + 1] = pos;
  }
  *n_states = k;
  *n_matching = cnt;
  return 0;
}

// ≙ the forward half of TLCTrace.getTrace: re-execute `nsteps` ordinals from Init on the GPU (k_replay)
// re-execute a path from Init: by ordinals (fps == nullptr) or by the fingerprints of its states (fps[0 .. nsteps], fps[0] = Init)
static int32_t replay_path(const vsrmc_model* m, int32_t device, const uint32_t* ords, const uint64_t* fps, int32_t nsteps, uint64_t* words,
                           uint64_t cap_words, uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!m || !words || !off || !actions || !n_states || nsteps < 0 || (nsteps && !ords && !fps)) return fail(VSRMC_E_ARG, "bad argument");
  int rc = check_device(device);
  if (rc) return rc;
  Model M = m->M;
  M.max_bag = 255 - M.fixed;     // replay is not bound by the LDS tile stride (simulation walks carry larger bags)
  const int level = nsteps + 1;
  if (cap_states < (u64)level + 1) return fail(VSRMC_E_ARG, "state buffers too small");
  u64 maxw = (u64)(M.fixed + M.max_bag + 8) * (u64)level;
  u64 *d_w = nullptr, *d_o = nullptr, *d_m = nullptr, *d_fps = nullptr;
  u32* d_ords = nullptr;
  HIPCHK(hipMalloc((void**)&d_w, maxw * 8));
  HIPCHK(hipMalloc((void**)&d_o, ((u64)level + 1) * 8));
  HIPCHK(hipMalloc((void**)&d_m, (u64)std::max(nsteps, 1) * 32));
  HIPCHK(hipMalloc((void**)&d_ords, (u64)std::max(nsteps, 1) * 4));
  if (fps) {
    HIPCHK(hipMalloc((void**)&d_fps, (u64)level * 8));
    HIPCHK(hipMemcpy(d_fps, fps, (u64)level * 8, hipMemcpyHostToDevice));
  }
  std::vector<u64> wire, dev(512);
  init_record_wire(M, wire);
  int len = wire_to_device(M, wire.data(), dev.data());
  u64 H[6];
  hash_full_host(M, (const u64*)dev.data(), H);
  for (int i = 0; i < M.np; i++) dev[M.h0 + i] = H[i];
  HIPCHK(hipMemcpy(d_w, dev.data(), len * 8, hipMemcpyHostToDevice));
  HIPCHK(hipMemset(d_m, 0, (u64)std::max(nsteps, 1) * 32));
  if (nsteps > 0 && ords) HIPCHK(hipMemcpy(d_ords, ords, (u64)nsteps * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL((M.model_id == 1 ? k_replay<1> : M.model_id == 2 ? k_replay<2> : k_replay<0>), dim3(1), dim3(64), 0, 0, M, d_w, d_o, d_ords, nsteps, d_m, d_fps,
                     (u32*)nullptr);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  std::vector<u64> hw(maxw), ho(level + 1), hm((size_t)std::max(nsteps, 1) * 4);
  HIPCHK(hipMemcpy(ho.data(), d_o, ((u64)level + 1) * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hw.data(), d_w, maxw * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(hm.data(), d_m, (u64)std::max(nsteps, 1) * 32, hipMemcpyDeviceToHost));
  (void)hipFree(d_w); (void)hipFree(d_o); (void)hipFree(d_m); (void)hipFree(d_ords);
  if (d_fps) (void)hipFree(d_fps);
  u64 pos = 0;
  for (int t = 0; t < level; t++) {
    const u64* r = &hw[ho[t]];
    u64 wl = (u64)M.h0 + hdr_nmsg(r[0]);
    if (pos + wl > cap_words) return fail(VSRMC_E_ARG, "word buffer too small");
    off[t] = pos;
    device_to_wire(M, r, words + pos);
    pos += wl;
    actions[t] = t == 0 ? 0 : (int32_t)hm[4 * (t - 1)];
    if (t > 0 && hm[4 * (t - 1) + 3])
      return fail(VSRMC_E_STATE, fps ? "trace replay: a state of the path has no successor with the next fingerprint"
                                     : "trace replay hit a disabled or failing step");
  }
  off[level] = pos;
  *n_states = (u64)level;
  return 0;
}

int32_t vsrmc_model_replay(const vsrmc_model* m, int32_t device, const uint32_t* ords, int32_t nsteps, uint64_t* words,
                           uint64_t cap_words, uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (nsteps && !ords) return fail(VSRMC_E_ARG, "bad argument");
  return replay_path(m, device, ords, nullptr, nsteps, words, cap_words, off, actions, cap_states, n_states);
}

// ≙ the forward half of TLCTrace.getTrace for a path given by the fingerprints of its states (fps[0] = Init's, n_fps >= 1): what a
// walk through the seen-set yields — at every step the successor with the next fingerprint is taken
int32_t vsrmc_model_replay_fps(const vsrmc_model* m, int32_t device, const uint64_t* fps, int32_t n_fps, uint64_t* words,
                               uint64_t cap_words, uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!fps || n_fps < 1) return fail(VSRMC_E_ARG, "bad argument");
  return replay_path(m, device, nullptr, fps, n_fps - 1, words, cap_words, off, actions, cap_states, n_states);
}

int32_t vsrmc_checker_trace_fp(vsrmc_checker* c, int32_t level, uint64_t fp, uint64_t* words, uint64_t cap_words, uint64_t* off,
                               int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!c || !words || !off || !actions || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  if (c->opt.world > 1) return fail(VSRMC_E_STATE, "sharded checker: walk the predecessor pointers with vsrmc_checker_lookup on every rank");
  if (level < 1 || level > c->level) return fail(VSRMC_E_ARG, "no such level");
  HIPCHK(hipSetDevice(c->opt.device));
  std::vector<u64> fps;
  int rc = walk_trace(c, fp, level, &fps);                     // through the seen-set, back to Init, on the device
  if (rc) return rc;
  return vsrmc_model_replay_fps(&c->model, c->opt.device, fps.data(), (int32_t)fps.size(), words, cap_words, off, actions, cap_states, n_states);
}

int32_t vsrmc_checker_trace(vsrmc_checker* c, int32_t level, uint64_t index, uint64_t* words, uint64_t cap_words, uint64_t* off,
                            int32_t* actions, uint64_t cap_states, uint64_t* n_states) {
  if (!c || !words || !off || !actions || !n_states) return fail(VSRMC_E_ARG, "NULL argument");
  if (level != c->level || index >= c->n_frontier)
    return fail(VSRMC_E_ARG, "no such state: states are addressed by index in the newest level only (older ones: vsrmc_checker_trace_fp)");
  HIPCHK(hipSetDevice(c->opt.device));
  u64 fp = 0;
  HIPCHK(hipMemcpy(&fp, c->lvl_fp + index, 8, hipMemcpyDeviceToHost));
  if (fp == 0) return fail(VSRMC_E_ARG, "no such state: the index is an unused slot of the level's index range");
  return vsrmc_checker_trace_fp(c, level, fp, words, cap_words, off, actions, cap_states, n_states);
}

void vsrmc_checker_destroy(vsrmc_checker* c) {
  if (!c) return;
  (void)hipSetDevice(c->opt.device);
  if (c->table) (void)hipFree(c->table);
  for (int b = 0; b < 2; b++) {
    if (c->words[b]) (void)(((c->host_frontier >> b) & 1) ? hipHostFree(c->words[b]) : hipFree(c->words[b]));
    if (c->off[b]) (void)hipFree(c->off[b]);
  }
  if (c->lvl_fp) (void)hipFree(c->lvl_fp);
  for (PassDst& B : c->scratch) {
    if (B.words) (void)hipFree(B.words);
    if (B.off) (void)hipFree(B.off);
    if (B.fp) (void)hipFree(B.fp);
  }
  if (c->pending) (void)hipFree(c->pending);
  if (c->ctl) (void)hipFree(c->ctl);
  if (c->d_find) (void)hipFree(c->d_find);
  if (c->rslot) (void)hipFree(c->rslot);
  if (c->filter) (void)hipFree(c->filter);
  if (c->cand_idx) (void)hipFree(c->cand_idx);
  for (int i = 0; i < 4; i++)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"

#include "vsr_shard_loop.hpp"

// ---------------------------------------------------------------------------------------------------------------
// TLC-style fingerprints (FP64 over TLC's serialisation of the view): a characterisation mode, never the identity the seen-set uses
// ---------------------------------------------------------------------------------------------------------------
#include "vsr_tlcfp.hpp"

extern "C" {

int32_t vsrmc_tlc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n, uint64_t* fps) {
  if (!m || !words || !off || !fps) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  int rc = check_device(device);
  if (rc) return rc;
  if (n == 0) return 0;
  for (u64 i = 0; i < n; i++)
    if (hdr_nmsg(words[off[i]]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  u64 *d_words = nullptr, *d_off = nullptr, *d_out = nullptr, total = 0;
  rc = upload_records(M, words, off, n, &d_words, &d_off, &total);
  if (rc) return rc;
  bool ok = hipMalloc((void**)&d_out, n * 8) == hipSuccess;
  if (ok) {
    hipLaunchKernelGGL(vsr::tlcfp::k_tlc_fingerprints, dim3((unsigned)std::min<u64>((n + 255) / 256, 4096)), dim3(256), 0, 0, M, d_words, d_off, (const u64*)nullptr, n, d_out);
    ok = hipGetLastError() == hipSuccess && hipMemcpy(fps, d_out, n * 8, hipMemcpyDeviceToHost) == hipSuccess;
  }
  (void)hipFree(d_words); (void)hipFree(d_off); (void)hipFree(d_out);
  return ok ? 0 : fail(VSRMC_E_HIP, "k_tlc_fingerprints failed");
}

int32_t vsrmc_tlc_view_bytes(const vsrmc_model* m, const uint64_t* record, int32_t permutation, uint8_t* out, uint64_t cap, uint64_t* n_bytes) {
  if (!m || !record || !n_bytes || (!out && cap)) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  if (permutation < 0 || permutation >= M.np) return fail(VSRMC_E_ARG, "no such permutation");
  if (hdr_nmsg(record[0]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  vsr::tlcfp::ByteSink s{out, cap, 0};
  vsr::tlcfp::put_view(M, s, record, record + M.h0, M.pitab[permutation]);     // wire layout: the bag follows the replica blocks
  *n_bytes = s.n;
  return 0;
}

int32_t vsrmc_tlc_min_permutation(const vsrmc_model* m, const uint64_t* record, int32_t* permutation) {
  if (!m || !record || !permutation) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = m->M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  if (hdr_nmsg(record[0]) > vsr::tlcfp::MAX_MSGS) return fail(VSRMC_E_REP, "record bag larger than the TLC fingerprint mode orders");
  *permutation = vsr::tlcfp::min_permutation(M, record, record + M.h0);
  return 0;
}

uint64_t vsrmc_fp64_extend(uint64_t fp, const uint8_t* bytes, uint64_t n) {
  vsr::tlcfp::FpSink s{fp, vsr::tlcfp::H_TABLE.t};
  for (u64 k = 0; k < n; k++) s.byte(bytes[k]);
  return s.fp;
}

uint64_t vsrmc_fp64_new(void) { return vsr::tlcfp::IRRED_POLY; }

int32_t vsrmc_checker_tlc_level_fps(vsrmc_checker* c, uint64_t* out, uint64_t cap, uint64_t* n, double* kernel_ms) {
  if (!c || !n) return fail(VSRMC_E_ARG, "NULL argument");
  const Model& M = c->model.M;
  if (M.model_id != 0) return fail(VSRMC_E_ARG, "the TLC fingerprint mode covers VSR.tla only");
  *n = c->n_valid;
  if (c->n_frontier == 0) return 0;
  if (out && cap < c->n_valid) return fail(VSRMC_E_ARG, "buffer too small");
  HIPCHK(hipSetDevice(c->opt.device));
  u64* d_out = nullptr;
  HIPCHK(hipMalloc((void**)&d_out, c->n_frontier * 8));
  hipEvent_t e0 = nullptr, e1 = nullptr;
  bool ok = hipEventCreate(&e0) == hipSuccess && hipEventCreate(&e1) == hipSuccess && hipEventRecord(e0, c->stream) == hipSuccess;
  hipLaunchKernelGGL(vsr::tlcfp::k_tlc_fingerprints, dim3((unsigned)std::min<u64>((c->n_frontier + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, M,
                     (const u64*)c->words[c->cur], (const u64*)nullptr, (const u64*)c->off[c->cur], c->n_frontier, d_out);
  ok = ok && hipGetLastError() == hipSuccess && hipEventRecord(e1, c->stream) == hipSuccess && hipStreamSynchronize(c->stream) == hipSuccess;
  float ms = 0;
  if (ok) (void)hipEventElapsedTime(&ms, e0, e1);
  if (kernel_ms) *kernel_ms = ms;
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (ok && out) {
    std::vector<u64> all(c->n_frontier);
    ok = hipMemcpy(all.data(), d_out, c->n_frontier * 8, hipMemcpyDeviceToHost) == hipSuccess;
    std::vector<u64> refs(c->n_frontier);
    ok = ok && hipMemcpy(refs.data(), c->off[c->cur], c->n_frontier * 8, hipMemcpyDeviceToHost) == hipSuccess;
    u64 k = 0;
    for (u64 i = 0; i < c->n_frontier && ok; i++)
      if (refs[i] != 0 && k < cap) out[k++] = all[i];
    *n = k;
    std::sort(out, out + k);
  }
  (void)hipFree(d_out);
  return ok ? 0 : fail(VSRMC_E_HIP, "k_tlc_fingerprints failed");
}

}  // extern "C"

// host_model.hpp — the lowered models: constants -> vsr::Model, the VSR.cfg reader (≙ tlc2.TLC -config), printing / parsing of states (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

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

// TEST HOOK, compiled only with -DVSRMC_TEST_HOOKS (vsr_tlaplus_amd/libvsrmc_hooks.so, which build.py makes beside the product library and only
// tests/test_sharded_gloo.py loads): VSRMC_TEST_FORCE_BAD=<hex fingerprint>:<mask> makes the state with that fingerprint fail the invariants of
// <mask> in the sharded and two-kernel paths (Model::test_bad_fp) — how a test puts a violator of masks 4 / 8 / 16 on a rank that does not
// own it.  The product library reads no such variable and evaluates no such compare.
void apply_test_hooks(Model& M) {
  M.test_bad_fp = 0;
  M.test_bad_mask = 0;
#ifdef VSRMC_TEST_HOOKS
  if (const char* e = std::getenv("VSRMC_TEST_FORCE_BAD")) {
    char* end = nullptr;
    const u64 fp = std::strtoull(e, &end, 16);
    if (end && *end == ':') { M.test_bad_fp = fp; M.test_bad_mask = (u32)std::strtoul(end + 1, nullptr, 0) & 31u; }
  }
  // VSRMC_TEST_MAX_BAG=<n>: a smaller bag capacity — how tests/test_probe_footprint.py brings the records of a small space to a representation limit
  // (LevelCtl::limit_unchecked: a probe pass must then be run again with every action applied)
  if (const char* e = std::getenv("VSRMC_TEST_MAX_BAG")) {
    const int n = std::atoi(e);
    if (n >= 4 && n < M.max_bag) M.max_bag = n;
  }
#endif
}

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
  M.fixed = M.h0 + M.np + VSR_PAD_WORDS;
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
  apply_test_hooks(out->M);
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
  M.fixed = M.h0 + 1 + VSR_PAD_WORDS;
  M.inv_mask = inv_mask;
  M.max_bag = 63 - M.fixed;
  M.m0 = 4 * R + R * n;
  for (int v = 1; v <= 7; v++) M.primtab |= (u32)(1 + ((v - 1) % R)) << (3 * v);   // Primary(v), VR_STATE_TRANSFER.tla:233-234
  out->symmetry = 0;
  out->value_names.clear();
  for (int v = 0; v < n; v++) out->value_names.push_back("v" + std::to_string(v + 1));
  apply_test_hooks(out->M);
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
  M.fixed = M.h0 + 1 + VSR_PAD_WORDS;
  M.inv_mask = inv_mask;
  M.max_bag = 63 - M.fixed;
  M.m0 = 4 * R + R * n;
  for (int v = 1; v <= 7; v++) M.primtab |= (u32)(1 + ((v - 1) % R)) << (3 * v);   // Primary(v), VR_APP_STATE.tla:238-239
  out->symmetry = 0;
  out->value_names.clear();
  for (int v = 0; v < n; v++) out->value_names.push_back(std::string(1, (char)('a' + v)));   // VR_APP_STATE.cfg:5 Values = {a, b}
  apply_test_hooks(out->M);
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
  for (int i = M.h0; i < M.fixed; i++) dev[i] = 0;
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


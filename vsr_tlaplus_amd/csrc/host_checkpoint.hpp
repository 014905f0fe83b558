// host_checkpoint.hpp — checkpoint / recover (≙ TLC checkpoints), status, frontier / level accessors, trace reconstruction, destroy (included by vsrmc.hip: one translation unit, the sections share its anonymous-namespace helpers).
#pragma once

extern "C" {
// ---- checkpoint / recover ≙ TLC's checkpoints (ModelChecker.checkpoint: FPSet.beginChkpt/commitChkpt, StateQueue, TLCTrace) ----
namespace {
struct ChkHeader {
  char magic[8];                 // "VSRMCCK4" (1 = the format with a separate trace log and index-based meta words; 2 = without the module in the header;
                                 // 3 = without the deep-search section: no checkpoint once a level existed in the seen-set only)
  int32_t consts[12];            // R, C, n, L, symmetry, inv_mask, assume_commit, np, module (model_id), words per replica, fixed words,
                                 // version of the fingerprint function: records and fingerprints mean nothing under another layout or hash
  int32_t level, shard;          // shard: 0 = unsharded, else world << 16 | rank (each rank writes and reads its own file)
  u64 n_frontier, n_valid, cur_w, distinct, total_generated, n_levels, table_entries, trace_entries;
  // what the automatic level scheme has learnt (a recovered run then makes the decisions the uninterrupted one would)
  u64 hist_new[2], g_last, cur_rec_w, cur_max_bag;
  // the search beyond the record buffers (vsr_deep.hpp): `deep` levels above `level` are complete in the seen-set and have no frontier; their
  // descriptors (DeepLevelRec: 5 words each) follow the header.  The frontier below is the BASE level every descent starts from.
  u64 deep, deep_g, deep_distinct, deep_generated;
  u64 cur_buf;                   // which of the two record buffers holds the frontier (level L lives in buffer (L - 1) mod 2 until a re-basing moves it)
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
  // (a sharded search beyond its record buffers: the rank's winner set travels in a section of its own behind the frontier, the level loop's own state —
  // the run's totals, the violation — in a sidecar the loop writes: vsrmc_shard_loop_save)
  HIPCHK(hipSetDevice(c->opt.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const Model& M = c->model.M;
  const std::string tmp = std::string(path) + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return fail(VSRMC_E_CFG, "cannot write " + tmp);
  ChkHeader h;
  std::memset(&h, 0, sizeof(h));
  std::memcpy(h.magic, "VSRMCCK4", 8);
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
  h.hist_new[0] = c->hist_new[0]; h.hist_new[1] = c->hist_new[1];
  h.g_last = c->g_last; h.cur_rec_w = c->cur_rec_w; h.cur_max_bag = c->bag_known ? c->cur_max_bag : ~(u64)0;
  h.cur_buf = (u64)c->cur;
  h.deep = (u64)c->deep; h.deep_g = c->deep_g; h.deep_distinct = c->deep_distinct; h.deep_generated = c->deep_generated;
  if ((size_t)c->deep != c->deep_lv.size()) return fail(VSRMC_E_STATE, "inconsistent deep-search state");
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
  static_assert(sizeof(DeepLevelRec) == 40, "a deep-level descriptor is five words");
  if (ok && c->deep) ok = std::fwrite(c->deep_lv.data(), sizeof(DeepLevelRec), (size_t)c->deep, f) == (size_t)c->deep;
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
  // optional section: the generator-side winner set of a sharded deep search, as (fingerprint, level) pairs
  if (ok && c->opt.world > 1 && c->deep && c->d_wset) {
    const u64 wslots = c->h_wset.mask + 1, wwin = std::min<u64>(wslots, (u64)1 << 26);
    Slot* d_w = nullptr;
    u64* d_wc = nullptr;
    ok = hipMalloc((void**)&d_w, wwin * sizeof(Slot)) == hipSuccess && hipMalloc((void**)&d_wc, 8) == hipSuccess;
    const u64 magic = 0x5445535754455357ull;                    // "WSETWSET"
    const long at = std::ftell(f);
    u64 head[2] = {magic, 0};
    ok = ok && std::fwrite(head, 8, 2, f) == 2;
    u64 wtotal = 0;
    for (u64 first = 0; first < wslots && ok; first += wwin) {
      u64 cnt = 0;
      const u64 n = std::min<u64>(wwin, wslots - first);
      ok = hipMemset(d_wc, 0, 8) == hipSuccess;
      hipLaunchKernelGGL(k_wset_export, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64*)c->h_wset.fp, (const u32*)c->h_wset.epoch, first, n, d_w, wwin, d_wc);
      ok = ok && hipStreamSynchronize(c->stream) == hipSuccess && hipMemcpy(&cnt, d_wc, 8, hipMemcpyDeviceToHost) == hipSuccess;
      ok = ok && dev_to_file(f, d_w, cnt * sizeof(Slot), buf);
      wtotal += cnt;
    }
    if (d_w) (void)hipFree(d_w);
    if (d_wc) (void)hipFree(d_wc);
    head[1] = wtotal;
    ok = ok && std::fseek(f, at, SEEK_SET) == 0 && std::fwrite(head, 8, 2, f) == 2 && std::fseek(f, 0, SEEK_END) == 0;
  }
  h.table_entries = total;
  ok = ok && std::fseek(f, 0, SEEK_SET) == 0 && std::fwrite(&h, sizeof(h), 1, f) == 1;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok || std::rename(tmp.c_str(), path) != 0) {
    std::remove(tmp.c_str());
    return fail(VSRMC_E_CFG, std::string("writing the checkpoint ") + path + " failed");
  }
  return 0;
}

int32_t vsrmc_checker_load(const vsrmc_model* m, const vsrmc_options* o_in, const char* path, vsrmc_checker** out) {
  if (!m || !o_in || !path || !out) return fail(VSRMC_E_ARG, "NULL argument");
  if (o_in->world > 1 && o_in->exact_ties) return fail(VSRMC_E_STATE, "checkpoints of sharded exact-mode checkers are not supported");
  FILE* f = std::fopen(path, "rb");
  if (!f) return fail(VSRMC_E_CFG, std::string("cannot read ") + path);
  ChkHeader h;
  bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "VSRMCCK4", 8) == 0;
  // header invariants (a truncated or foreign file must not become an inconsistent checker)
  if (ok) ok = h.level >= 1 && h.level < 511 && (u64)h.level == h.n_levels && h.n_valid <= h.n_frontier && h.trace_entries == 0 &&
               h.deep < 511 && (u64)h.level + h.deep < 511 && (h.deep == 0 || h.deep_distinct >= h.distinct) && h.cur_buf <= 1;
  std::vector<DeepLevelRec> deep_lv((size_t)(ok ? h.deep : 0));
  if (ok && h.deep) ok = std::fread(deep_lv.data(), sizeof(DeepLevelRec), (size_t)h.deep, f) == (size_t)h.deep;
  if (ok && h.deep) {                                           // the descriptors against the header: every level non-empty, this handle's shares add up to its total
    u64 sum_local = 0;
    for (const DeepLevelRec& d : deep_lv) { ok = ok && d.n_new > 0 && d.n_local <= d.n_new; sum_local += d.n_local; }
    ok = ok && sum_local == h.deep_distinct - h.distinct;
  }
  if (!ok) {
    std::fclose(f);
    return fail(VSRMC_E_CFG, std::string(path) + " is not a (consistent) vsrmc checkpoint");
  }
  // options with zeros (the CLI's and ModelChecker.auto()'s defaults) are sized from the free device memory FIRST: the checkpoint is held
  // against the sizes the checker will really have, not against the zeros (round-4 advice: `vsrmc -recover ck` without explicit sizes failed).
  // A seen-set that had grown beyond the automatic size (vsrmc_checker_room) is sized from the checkpoint BEFORE the record buffers are sized from
  // what is left (round-5 advice: the table was bumped after the buffers had taken the memory it needs).
  vsrmc_options sized = *o_in;
  if (o_in->table_log2 == 0) {
    vsrmc_options probe = *o_in;
    const int rc0 = autosize_options(&probe, m->M);
    if (rc0) { std::fclose(f); return rc0; }
    int lg = (int)probe.table_log2;
    while (lg < 36 && (double)h.table_entries > 0.6 * (double)((u64)1 << lg)) lg++;
    if (lg != (int)probe.table_log2) sized.table_log2 = lg;     // (pinned: autosize_options takes it off the memory it shares out)
  }
  {
    const int rc0 = autosize_options(&sized, m->M);
    if (rc0) { std::fclose(f); return rc0; }
  }
  const vsrmc_options* o = &sized;
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
  const int buf_of_level = (int)h.cur_buf;                      // the record buffer the frontier was in: its level keeps alternating from there
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
  // optional section: the winner set of a sharded deep search (vsrmc_checker_save)
  u64 whead[2] = {0, 0};
  const bool has_wset = ok && std::fread(whead, 8, 2, f) == 2 && whead[0] == 0x5445535754455357ull;
  if (ok && h.deep && o->world > 1 && !has_wset) ok = false;    // (a sharded search beyond its buffers cannot regenerate anything without it)
  if (ok && has_wset) {
    c->deep = (int)h.deep;                                       // (wset_ensure sizes nothing from it, but a set only exists for a deep search)
    int wrc = wset_ensure(c);
    u64 need = whead[1];
    while (!wrc && (double)need > 0.6 * (double)(c->h_wset.mask + 1) && wset_grow(c) == 0) {}
    Slot* d_w = nullptr;
    u32* d_werr = nullptr;
    const u64 wwin = (u64)1 << 24;
    ok = !wrc && hipMalloc((void**)&d_w, wwin * sizeof(Slot)) == hipSuccess && hipMalloc((void**)&d_werr, 4) == hipSuccess && hipMemset(d_werr, 0, 4) == hipSuccess;
    for (u64 done = 0; done < whead[1] && ok; done += wwin) {
      const u64 k = std::min<u64>(wwin, whead[1] - done);
      ok = file_to_dev(f, d_w, k * sizeof(Slot), buf);
      hipLaunchKernelGGL(k_wset_import, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, (const WSet*)c->d_wset, (const Slot*)d_w, k, d_werr);
      ok = ok && hipStreamSynchronize(c->stream) == hipSuccess;
    }
    u32 werr = 0;
    if (ok) ok = hipMemcpy(&werr, d_werr, 4, hipMemcpyDeviceToHost) == hipSuccess && werr == 0;
    if (d_w) (void)hipFree(d_w);
    if (d_werr) (void)hipFree(d_werr);
    c->wset_used = true;
    c->wset_dirty = false;
    c->wepoch = 0;
  }
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
  c->hist_new[0] = h.hist_new[0]; c->hist_new[1] = h.hist_new[1];
  c->g_last = std::max<u64>(2, h.g_last); c->cur_rec_w = h.cur_rec_w;
  if (h.cur_max_bag != ~(u64)0 && o->world == 1) { c->cur_max_bag = h.cur_max_bag; c->bag_known = true; }
  if (h.deep) {                                                // the levels beyond the base: in the seen-set (restored above), described here
    c->deep = (int)h.deep;
    c->deep_lv = deep_lv;
    c->deep_g = std::max<u64>(2, h.deep_g);
    c->deep_distinct = h.deep_distinct;
    c->deep_generated = h.deep_generated;
    c->deep_regen_done = true;                                 // taken bits of earlier descents travelled with the slots: cleared before the next one;
                                                               // the base level's lvl_fp array had been reused as scratch: states are addressed by fingerprint
  }
  *out = c;
  return 0;
}

int32_t vsrmc_checker_status(vsrmc_checker* c, vsrmc_level_info* info) {
  if (!c || !info) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(info, 0, sizeof(*info));
  info->level = c->level;                                      // the newest STORED level
  info->n_new = c->n_valid;
  info->reserved0 = c->deep;                                   // levels beyond it that are complete in the seen-set only (vsrmc_checker_deepen)
  info->frontier = c->deep ? c->deep_lv.back().n_new : c->n_valid;   // states of the deepest complete level
  info->distinct = c->deep ? c->deep_distinct : c->distinct;
  info->total_generated = c->deep ? c->deep_generated : c->total_generated;
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
  if (c->deep_regen_done) return fail(VSRMC_E_STATE, "the fingerprint array of the newest stored level was reused by a descent of the deep search (vsrmc_checker_deepen): states are addressed by fingerprint from here on (vsrmc_checker_trace_fp, _lookup)");
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
  if (c->deep_regen_done) return fail(VSRMC_E_STATE, "the fingerprint array of the newest stored level was reused by a descent of the deep search (vsrmc_checker_deepen): states are addressed by fingerprint from here on (vsrmc_checker_trace_fp, _lookup)");
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
  if (c->deep_regen_done) return fail(VSRMC_E_STATE, "the fingerprint array of the newest stored level was reused by a descent of the deep search (vsrmc_checker_deepen): states are addressed by fingerprint from here on (vsrmc_checker_trace_fp, _lookup)");
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
    if (B.words && B.own_words) (void)hipFree(B.words);
    if (B.off && B.own_index) (void)hipFree(B.off);
    if (B.fp && B.own_index) (void)hipFree(B.fp);
  }
  if (c->pending) (void)hipFree(c->pending);
  if (c->ctl) (void)hipFree(c->ctl);
  if (c->d_find) (void)hipFree(c->d_find);
  for (u64* q : c->redo_buf) if (q) (void)hipFree(q);
  if (c->rslot) (void)hipFree(c->rslot);
  if (c->filter) (void)hipFree(c->filter);
  if (c->cand_idx) (void)hipFree(c->cand_idx);
  wset_free(c);
  if (c->claim_bits) (void)hipFree(c->claim_bits);
  for (int i = 0; i < 4; i++)
    if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

}  // extern "C"

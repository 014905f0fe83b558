// vsr_shard_loop.hpp — the level loop of a sharded run in C++ (included by vsrmc.hip: it drives the vsrmc_shard_* phases of one rank's
// checker and talks to the other ranks through a vsrmc_comm).
//
// TLC role replaced: the distributed mode's TLCServer / TLCWorker round trip (fingerprints shipped to the FPSet servers that own them,
// SURVEY §8e) — here one rank per GPU, owner(fp) = high fingerprint bits mod world, 16-byte (fp, key) candidates to the owner and one
// verdict byte back per BFS level.  Round 2 ran this protocol as a Python loop over torch.distributed (vsr_tlaplus_amd/sharded.py,
// kept: it is what the CPU gloo tests drive against a stand-in engine); this is the same protocol without the interpreter between
// the phases: per sharded level one count all-gather, two all-to-all-v (candidates out, verdict bytes back), one all-gather to
// compare frontier sizes (+ three all-to-all-v when records have to move), one all-gather with the level's figures.
//
// Transport = vsrmc_comm (include/vsrmc.h): (a) RCCL — grouped ncclSend / ncclRecv on the checker's stream, small all-gathers through
// a device staging buffer; librccl is dlopen'ed when such a comm is created, libvsrmc.so itself does not link it; (b) any caller-
// supplied pair of functions moving HOST buffers (the tests run gloo through ctypes callbacks with every rank on one GPU): the loop
// then stages the buckets through host memory.
#pragma once
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

namespace {

// ---- RCCL transport ------------------------------------------------------------------------------------------------------------
struct RcclId128 { char b[128]; };   // ncclUniqueId: 128 opaque bytes, passed by value
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, RcclId128, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
typedef decltype(RcclApi::CommInitRank) RcclInitFn;

RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {   // a librccl a framework loaded already serves this too
      api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (api.lib) break;
    }
    if (api.lib) {
      api.GetUniqueId = (int (*)(void*))dlsym(api.lib, "ncclGetUniqueId");
      api.CommInitRank = (RcclInitFn)dlsym(api.lib, "ncclCommInitRank");
      api.CommDestroy = (int (*)(void*))dlsym(api.lib, "ncclCommDestroy");
      api.GroupStart = (int (*)())dlsym(api.lib, "ncclGroupStart");
      api.GroupEnd = (int (*)())dlsym(api.lib, "ncclGroupEnd");
      api.Send = (int (*)(const void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclSend");
      api.Recv = (int (*)(void*, size_t, int, int, void*, hipStream_t))dlsym(api.lib, "ncclRecv");
      api.AllGather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))dlsym(api.lib, "ncclAllGather");
      api.GetErrorString = (const char* (*)(int))dlsym(api.lib, "ncclGetErrorString");
      if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.GroupStart || !api.GroupEnd || !api.Send || !api.Recv || !api.AllGather)
        api.lib = nullptr;
    }
  }
  return api.lib ? &api : nullptr;
}

struct RcclCtx {
  void* comm = nullptr;
  int rank = 0, world = 1, device = 0;
  char* d_stage = nullptr;       // (world + 1) * STAGE bytes: small all-gathers
  hipStream_t stream = nullptr;  // for the small all-gathers; all-to-all-v runs on the stream the loop passes
  enum { STAGE = 512 };
  bool always_call = false;      // issue the collectives even at world 1 (latency measurements)
};
const int NCCL_CHAR = 0;         // ncclInt8 / ncclChar

int rccl_alltoallv(void* vctx, const void* send, const uint64_t* scnt, const uint64_t* soff, void* recv, const uint64_t* rcnt,
                   const uint64_t* roff, uint32_t eb, void* stream) {
  RcclCtx* x = (RcclCtx*)vctx;
  RcclApi* a = rccl_api();
  if (x->world == 1 && !x->always_call) return 0;                // nobody to talk to
  int rc = a->GroupStart();
  for (int p = 0; p < x->world && rc == 0; p++) {
    if (p == x->rank) continue;
    if (scnt[p]) rc = a->Send((const char*)send + soff[p] * eb, scnt[p] * eb, NCCL_CHAR, p, x->comm, (hipStream_t)stream);
    if (rc == 0 && rcnt[p]) rc = a->Recv((char*)recv + roff[p] * eb, rcnt[p] * eb, NCCL_CHAR, p, x->comm, (hipStream_t)stream);
  }
  const int rc2 = a->GroupEnd();
  if (rc == 0) rc = rc2;
  if (rc == 0 && hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = -1;
  return rc == 0 ? 0 : fail(VSRMC_E_HIP, std::string("RCCL all-to-all-v: ") + (rc > 0 && a->GetErrorString ? a->GetErrorString(rc) : "stream error"));
}
int rccl_allgather(void* vctx, const void* send, void* recv, uint32_t bytes) {
  RcclCtx* x = (RcclCtx*)vctx;
  RcclApi* a = rccl_api();
  if (bytes > RcclCtx::STAGE) return fail(VSRMC_E_ARG, "all-gather record larger than the staging buffer");
  if (x->world == 1 && !x->always_call) { std::memcpy(recv, send, bytes); return 0; }   // VSRMC_COMM_ALWAYS_CALL=1: measure the collective's latency at world 1
  if (hipMemcpyAsync(x->d_stage, send, bytes, hipMemcpyHostToDevice, x->stream) != hipSuccess) return fail(VSRMC_E_HIP, "all-gather staging copy");
  const int rc = a->AllGather(x->d_stage, x->d_stage + RcclCtx::STAGE, bytes, NCCL_CHAR, x->comm, x->stream);
  if (rc != 0) return fail(VSRMC_E_HIP, std::string("RCCL all-gather: ") + (a->GetErrorString ? a->GetErrorString(rc) : "?"));
  if (hipMemcpyAsync(recv, x->d_stage + RcclCtx::STAGE, (size_t)bytes * x->world, hipMemcpyDeviceToHost, x->stream) != hipSuccess ||
      hipStreamSynchronize(x->stream) != hipSuccess)
    return fail(VSRMC_E_HIP, "all-gather result copy");
  return 0;
}

// deterministic plan (the same on every rank): move k states from src to dst so that every rank ends within `tol` of the mean
struct Move { int src, dst; u64 k; };
std::vector<Move> balance_plan(const std::vector<u64>& counts, double tol = 1.25, u64 min_per_rank = 64) {
  const int w = (int)counts.size();
  u64 total = 0, hi = 0, lo = ~(u64)0;
  for (u64 c : counts) { total += c; hi = std::max(hi, c); lo = std::min(lo, c); }
  std::vector<Move> plan;
  if (w < 2 || total < (u64)w * min_per_rank) return plan;
  const double mean = (double)total / w;
  if ((double)hi <= tol * mean && (double)lo >= mean / tol) return plan;
  std::vector<std::pair<int, u64>> surplus, deficit;
  for (int r = 0; r < w; r++) {
    const u64 target = total / w + ((u64)r < total % w ? 1 : 0);
    if (counts[r] > target) surplus.push_back({r, counts[r] - target});
    else if (counts[r] < target) deficit.push_back({r, target - counts[r]});
  }
  for (auto& s : surplus)
    for (auto& d : deficit) {
      if (s.second == 0) break;
      const u64 k = std::min(s.second, d.second);
      if (k) { plan.push_back(Move{s.first, d.first, k}); s.second -= k; d.second -= k; }
    }
  return plan;
}

}  // namespace

struct vsrmc_shard_loop {
  vsrmc_checker* c = nullptr;
  vsrmc_comm comm;
  int rank = 0, world = 1;
  u64 cand_cap = 0, rec_cap = 0, rec_words_cap = 0, replicate_below = 0;
  bool replicated = true;
  int level = 1;
  u64 distinct = 1, n_frontier = 1, moved = 0, bytes_sent = 0;
  bool has_violation = false;
  u64 viol_fp = 0;
  int viol_mask = 0, viol_level = 0;
  // levels beyond the record buffers (vsr_deep.hpp, sharded): `deep` levels above `level` are complete in the seen-sets
  int deep = 0;
  u64 probe_parent_fp = 0;       // a violation found by a probe pass: its parent (a state of level viol_level - 1, in some shard) ...
  bool viol_probed = false;      // ... and that it was found that way (the violator itself is in no seen-set)
  // device buffers
  u64* cand_send = nullptr;      // [world][cand_cap][2]
  u64* cand_recv = nullptr;      // up to world * cand_cap pairs, packed by source rank
  uint8_t* verdict_out = nullptr;  // one byte per received candidate
  uint8_t* verdict_in = nullptr;   // [world][cand_cap]
  u64 *mv_words = nullptr, *mv_off = nullptr, *mv_fp = nullptr, *rv_words = nullptr, *rv_off = nullptr, *rv_fp = nullptr;   // rebalancing streams (allocated on first use)
  // host staging for transports that move host memory
  char* h_send = nullptr;
  char* h_recv = nullptr;
  u64 h_cap = 0;
  // the overlapped exchange (round 6): a second set of buckets and a second stream — the all-to-all, the owners' claims, the verdicts and their application
  // of slice k run on `stream2` while k_expand of slice k + 1 runs on the checker's stream
  hipStream_t stream2 = nullptr;           // (the two sets of buckets are the two HALVES of the loop's buffers: a slice is sized for half a bucket, no extra memory)
  u64 overlap_levels = 0, overlap_slices = 0;
};

namespace {

int loop_stage_cap(vsrmc_shard_loop* l, u64 bytes) {
  if (bytes <= l->h_cap) return 0;
  if (l->h_send) (void)hipHostFree(l->h_send);
  if (l->h_recv) (void)hipHostFree(l->h_recv);
  l->h_send = l->h_recv = nullptr;
  l->h_cap = 0;
  if (hipHostMalloc((void**)&l->h_send, bytes) != hipSuccess || hipHostMalloc((void**)&l->h_recv, bytes) != hipSuccess)
    return fail(VSRMC_E_HIP, "hipHostMalloc of the exchange staging buffers failed");
  l->h_cap = bytes;
  return 0;
}

// all-to-all-v of device buffers through the loop's transport (element offsets / counts; nothing is sent to oneself)
int loop_alltoallv(vsrmc_shard_loop* l, const void* d_send, const u64* scnt, const u64* soff, void* d_recv, const u64* rcnt, const u64* roff, u32 eb,
                   hipStream_t stream = nullptr) {
  if (!stream) stream = l->c->stream;
  for (int p = 0; p < l->world; p++)
    if (p != l->rank) l->bytes_sent += scnt[p] * eb;
  if (!l->comm.host_buffers) return l->comm.alltoallv(l->comm.ctx, d_send, scnt, soff, d_recv, rcnt, roff, eb, (void*)stream);
  // host transport: pack the non-empty buckets contiguously on the host, exchange, unpack
  std::vector<u64> hs(l->world, 0), hr(l->world, 0);
  u64 st = 0, rt = 0;
  for (int p = 0; p < l->world; p++) {
    hs[p] = st; hr[p] = rt;
    if (p != l->rank) { st += scnt[p]; rt += rcnt[p]; }
  }
  int rc = loop_stage_cap(l, std::max<u64>(1, std::max(st, rt)) * eb);
  if (rc) return rc;
  for (int p = 0; p < l->world; p++)
    if (p != l->rank && scnt[p] && hipMemcpyAsync(l->h_send + hs[p] * eb, (const char*)d_send + soff[p] * eb, scnt[p] * eb, hipMemcpyDeviceToHost, stream) != hipSuccess)
      return fail(VSRMC_E_HIP, "staging copy (device to host)");
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(VSRMC_E_HIP, "stream synchronisation");
  std::vector<u64> sc(scnt, scnt + l->world), rcv(rcnt, rcnt + l->world);
  sc[l->rank] = rcv[l->rank] = 0;
  rc = l->comm.alltoallv(l->comm.ctx, l->h_send, sc.data(), hs.data(), l->h_recv, rcv.data(), hr.data(), eb, nullptr);
  if (rc) return rc > 0 ? fail(VSRMC_E_HIP, "the caller's all-to-all-v failed") : rc;
  for (int p = 0; p < l->world; p++)
    if (p != l->rank && rcnt[p] && hipMemcpyAsync((char*)d_recv + roff[p] * eb, l->h_recv + hr[p] * eb, rcnt[p] * eb, hipMemcpyHostToDevice, stream) != hipSuccess)
      return fail(VSRMC_E_HIP, "staging copy (host to device)");
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(VSRMC_E_HIP, "stream synchronisation");
  return 0;
}

int loop_allgather(vsrmc_shard_loop* l, const void* mine, void* all, u32 bytes) {
  const int rc = l->comm.allgather(l->comm.ctx, mine, all, bytes);
  return rc > 0 ? fail(VSRMC_E_HIP, "the caller's all-gather failed") : rc;
}

// ---- the overlapped exchange (round 6) ---------------------------------------------------------------------------------------------------------
// Rounds 2-5 ran a sharded level strictly in sequence: k_expand over the whole frontier, THEN the candidates to their owners, the owners' claims, the verdict
// bytes back, the losers withdrawn — the fabric idle while the kernel runs and the CUs idle while the fabric does.  Here the frontier is cut into `nb` slices
// (the same number on every rank: every slice has its collectives) and the exchange of slice k — count all-gather, all-to-all of (fp, key), k_claim_batch_fused,
// all-to-all of the verdict bytes, k_apply_verdict, on `stream2` and a second set of buckets — runs while k_expand of slice k + 1 runs on the checker's stream.
// The slices of a level append to one next frontier and add up in one control block (host_checker.hpp: expand_slice_launch); the seen-set is claimed by the
// lanes of k_expand and by k_claim_batch_fused concurrently, both with the same compare-and-swap / min-merge, so a state is inserted by exactly one of them
// and its key is the minimum over every candidate whichever slice carried it.  What a level leaves — new states, counts, keys — does not depend on nb.
int loop_overlap_buffers(vsrmc_shard_loop* l) {
  vsrmc_checker* c = l->c;
  if (!l->stream2 && hipStreamCreateWithFlags(&l->stream2, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); return fail(VSRMC_E_HIP, "overlapped exchange: second stream"); }
  // the two sets of buckets are the two halves of the loop's exchange buffers (and of the checker's "where was it written" array): a slice announces at most
  // 0.45 of half a bucket per owner, so the overlap costs no memory — what autosize_options set aside for the winner set and the scratch buffers of the deep
  // search stays theirs (a second set of full-size buckets, 15 GB per rank on the README configuration at world 2, took exactly that)
  if (c->cand_idx_cap < (u64)l->world * l->cand_cap) {
    if (c->cand_idx) (void)hipFree(c->cand_idx);
    c->cand_idx = nullptr;
    c->cand_idx_cap = 0;
    if (hipMalloc((void**)&c->cand_idx, (u64)l->world * l->cand_cap * 8) != hipSuccess) { (void)hipGetLastError(); return fail(VSRMC_E_HIP, "hipMalloc of the candidate index array failed"); }
    c->cand_idx_cap = (u64)l->world * l->cand_cap;
  }
  return 0;
}

// the exchange of one slice on stream2: counts gathered, candidates out, claims, verdicts back, losers withdrawn.  *rc: this rank's local error (it keeps
// taking part in the collectives: their sizes are fixed by the counts); return value: a failed collective (nobody can go on).
int loop_exchange_slice(vsrmc_shard_loop* l, int set, const u64* counts, int* rc) {
  vsrmc_checker* c = l->c;
  const int w = l->world, me = l->rank;
  const u64 hcap = l->cand_cap / 2;                               // set 0 / 1 = the first / second half of every buffer, [owner][hcap] each
  u64* cand_send = l->cand_send + (u64)set * (u64)w * hcap * 2;
  u64* cand_recv = l->cand_recv + (u64)set * (u64)w * hcap * 2;
  uint8_t* verdict_out = l->verdict_out + (u64)set * (u64)w * hcap;
  uint8_t* verdict_in = l->verdict_in + (u64)set * (u64)w * hcap;
  u64* cand_idx = c->cand_idx + (u64)set * (u64)w * hcap;
  struct CountRow { u64 cnt[8]; u64 err; } mine, zero;
  std::memset(&zero, 0, sizeof(zero));
  mine = zero;
  for (int p = 0; p < w; p++) mine.cnt[p] = (p == me || *rc) ? 0 : std::min<u64>(counts[p], hcap);
  mine.err = *rc ? (u64)(*rc < 0 ? -*rc : *rc) : 0;
  std::vector<CountRow> all((size_t)w, zero);
  const int crc = loop_allgather(l, &mine, all.data(), (u32)sizeof(CountRow));
  if (crc) return crc;
  u64 worst = 0;
  for (const CountRow& r : all) worst = std::max(worst, r.err);
  if (worst) return *rc ? *rc : fail(VSRMC_E_STATE, "level " + std::to_string(l->level + 1) + ", phase expand (a slice): error " + std::to_string((long long)worst) + " on another rank");
  u64 scnt[8], soff[8], rcnt[8], roff[8], n_recv = 0;
  for (int p = 0; p < w; p++) {
    scnt[p] = mine.cnt[p];
    soff[p] = (u64)p * hcap;
    rcnt[p] = p == me ? 0 : all[(size_t)p].cnt[me];
    roff[p] = n_recv;
    n_recv += rcnt[p];
  }
  if (n_recv > (u64)w * hcap) return fail(VSRMC_E_REP, "more candidates received than the exchange buffer holds");
  int xrc = loop_alltoallv(l, cand_send, scnt, soff, cand_recv, rcnt, roff, 16, l->stream2);
  if (!*rc) *rc = xrc;
  if (!*rc && n_recv) {
    hipLaunchKernelGGL(k_claim_batch_fused, dim3((unsigned)((n_recv + 255) / 256)), dim3(256), 0, l->stream2, c->table, c->tmask, cand_recv, n_recv, c->level + 1, verdict_out, c->ctl);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(l->stream2) != hipSuccess) *rc = fail(VSRMC_E_HIP, "overlapped exchange: claim kernel");
  }
  xrc = loop_alltoallv(l, verdict_out, rcnt, roff, verdict_in, scnt, soff, 1, l->stream2);
  if (!*rc) *rc = xrc;
  const int nxt = c->cur ^ 1;
  for (int o = 0; o < w && !*rc; o++) {
    if (o == me || scnt[o] == 0) continue;
    hipLaunchKernelGGL(k_apply_verdict, dim3((unsigned)((scnt[o] + 255) / 256)), dim3(256), 0, l->stream2, cand_send + 2 * (u64)o * hcap,
                       cand_idx + (u64)o * hcap, verdict_in + (u64)o * hcap, scnt[o], c->off[nxt], c->lvl_fp, c->ctl, (const WSet*)nullptr, 0);
    if (hipGetLastError() != hipSuccess) *rc = fail(VSRMC_E_HIP, "overlapped exchange: verdict kernel");
  }
  if (hipStreamSynchronize(l->stream2) != hipSuccess && !*rc) *rc = fail(VSRMC_E_HIP, "overlapped exchange: stream");
  return 0;
}

// steps 1-5 of a sharded level in nb >= 2 slices; leaves the checker as vsrmc_shard_materialize does (c->h, nx_n, nx_w)
int loop_level_head_overlapped(vsrmc_shard_loop* l, int nb, int* rc) {
  vsrmc_checker* c = l->c;
  *rc = loop_overlap_buffers(l);
  const u64 n = c->n_frontier;
  u64 counts[2][8];
  std::memset(counts, 0, sizeof(counts));
  for (int k = 0; k <= nb; k++) {
    // (1) slice k onto the checker's stream — its buckets are set k & 1, free since the exchange of slice k - 2 ended (it ended inside iteration k - 1)
    bool launched = false;
    if (k < nb && !*rc) {
      const u64 a = n * (u64)k / (u64)nb, b = n * (u64)(k + 1) / (u64)nb;
      const u64 hcap = l->cand_cap / 2;
      vsrmc_shard_io io;
      io.cand_send = l->cand_send + (u64)(k & 1) * (u64)l->world * hcap * 2;
      io.cand_cap = hcap;
      *rc = expand_slice_launch(c, &io, c->cand_idx + (u64)(k & 1) * (u64)l->world * hcap, a, b - a, k == 0);
      launched = !*rc;
    }
    // (2) meanwhile: the exchange of slice k - 1 (every rank takes part, whatever its own state)
    if (k >= 1) {
      const int crc = loop_exchange_slice(l, (k - 1) & 1, counts[(k - 1) & 1], rc);
      if (crc) { if (launched) (void)hipStreamSynchronize(c->stream); return crc; }
    }
    // (3) slice k has run
    if (launched) {
      const int wrc = expand_slice_wait(c, counts[k & 1]);
      if (!*rc) *rc = wrc;
    } else if (k < nb) {
      std::memset(counts[k & 1], 0, sizeof(counts[0]));
    }
  }
  l->overlap_levels++;
  l->overlap_slices += (u64)nb;
  if (!*rc) {
    if (hipMemcpy(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost) != hipSuccess) *rc = fail(VSRMC_E_HIP, "overlapped level: control block");
    else if (c->h.err) *rc = level_error(c, c->h, c->level + 1);
    else if (c->h.full) { c->h.err = ERR_FRONTIER_FULL; c->h.err_info = c->h.full_info << 16; *rc = level_error(c, c->h, c->level + 1); }
    else if (c->h.ties) { c->failed = 1; *rc = fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2): create the checker with vsrmc_options.exact_ties = 1"); }
    else { c->nx_n = c->h.n_new; c->nx_w = c->h.words_new; }
  }
  return 0;
}

}  // namespace

extern "C" {

int32_t vsrmc_comm_rccl_unique_id(uint8_t* id128) {
  if (!id128) return fail(VSRMC_E_ARG, "NULL argument");
  RcclApi* a = rccl_api();
  if (!a) return fail(VSRMC_E_HIP, "librccl.so could not be loaded");
  const int rc = a->GetUniqueId(id128);
  return rc == 0 ? 0 : fail(VSRMC_E_HIP, std::string("ncclGetUniqueId: ") + (a->GetErrorString ? a->GetErrorString(rc) : "?"));
}

int32_t vsrmc_comm_rccl_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, vsrmc_comm** out) {
  if (!id128 || !out || world < 1 || world > 8 || rank < 0 || rank >= world) return fail(VSRMC_E_ARG, "bad RCCL communicator arguments (1 <= world <= 8)");
  RcclApi* a = rccl_api();
  if (!a) return fail(VSRMC_E_HIP, "librccl.so could not be loaded");
  HIPCHK(hipSetDevice(device));
  RcclCtx* x = new RcclCtx();
  x->rank = rank; x->world = world; x->device = device;
  x->always_call = std::getenv("VSRMC_COMM_ALWAYS_CALL") != nullptr;
  RcclId128 id;
  std::memcpy(id.b, id128, 128);
  int rc = a->CommInitRank(&x->comm, world, id, rank);
  if (rc != 0) { delete x; return fail(VSRMC_E_HIP, std::string("ncclCommInitRank: ") + (a->GetErrorString ? a->GetErrorString(rc) : "?")); }
  if (hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking) != hipSuccess || hipMalloc((void**)&x->d_stage, (size_t)(world + 1) * RcclCtx::STAGE) != hipSuccess) {
    a->CommDestroy(x->comm);
    delete x;
    return fail(VSRMC_E_HIP, "RCCL communicator: stream / staging buffer");
  }
  vsrmc_comm* cm = new vsrmc_comm();
  cm->ctx = x; cm->rank = rank; cm->world = world; cm->host_buffers = 0;
  cm->alltoallv = rccl_alltoallv;
  cm->allgather = rccl_allgather;
  *out = cm;
  return 0;
}

void vsrmc_comm_rccl_destroy(vsrmc_comm* cm) {
  if (!cm) return;
  RcclCtx* x = (RcclCtx*)cm->ctx;
  if (x) {
    RcclApi* a = rccl_api();
    if (a && x->comm) a->CommDestroy(x->comm);
    if (x->d_stage) (void)hipFree(x->d_stage);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
  }
  delete cm;
}

int32_t vsrmc_shard_loop_create(vsrmc_checker* c, const vsrmc_comm* comm, uint64_t cand_cap, uint64_t rec_cap, uint64_t rec_words_cap,
                                uint64_t replicate_below, vsrmc_shard_loop** out) {
  if (!c || !comm || !out || !comm->alltoallv || !comm->allgather) return fail(VSRMC_E_ARG, "NULL argument");
  if (comm->world != c->opt.world || comm->rank != c->opt.rank) return fail(VSRMC_E_ARG, "the communicator and the checker disagree about rank / world");
  if (c->opt.exact_ties) return fail(VSRMC_E_STATE, "the native level loop runs single-pass levels (exact_ties = 0)");
  if (cand_cap < 1024) return fail(VSRMC_E_ARG, "cand_cap too small");
  if (replicate_below > 1 && (c->level != 1 || c->n_valid != c->n_frontier || c->n_frontier != 1))
    return fail(VSRMC_E_STATE, "a loop that starts in the replicated phase needs a checker in its initial state (level 1, Init on every rank)");
  HIPCHK(hipSetDevice(c->opt.device));
  vsrmc_shard_loop* l = new vsrmc_shard_loop();
  l->c = c; l->comm = *comm; l->rank = comm->rank; l->world = comm->world;
  l->cand_cap = cand_cap; l->rec_cap = rec_cap; l->rec_words_cap = rec_words_cap; l->replicate_below = replicate_below;
  const u64 w = (u64)l->world;
  hipError_t e = hipMalloc((void**)&l->cand_send, w * cand_cap * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&l->cand_recv, w * cand_cap * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&l->verdict_out, w * cand_cap);
  if (e == hipSuccess) e = hipMalloc((void**)&l->verdict_in, w * cand_cap);
  if (e != hipSuccess) { vsrmc_shard_loop_destroy(l); return fail(VSRMC_E_HIP, std::string("hipMalloc of the exchange buffers: ") + hipGetErrorString(e)); }
  l->level = c->level; l->distinct = c->distinct; l->n_frontier = c->n_frontier;
  if (replicate_below <= 1) {
    u64 kept = 0;
    const int rc = vsrmc_shard_partition(c, &kept);
    if (rc) { vsrmc_shard_loop_destroy(l); return rc; }
    l->replicated = false;
  }
  *out = l;
  return 0;
}

void vsrmc_shard_loop_destroy(vsrmc_shard_loop* l) {
  if (!l) return;
  for (void* p : {(void*)l->cand_send, (void*)l->cand_recv, (void*)l->verdict_out, (void*)l->verdict_in, (void*)l->mv_words, (void*)l->mv_off,
                  (void*)l->mv_fp, (void*)l->rv_words, (void*)l->rv_off, (void*)l->rv_fp})
    if (p) (void)hipFree(p);
  if (l->stream2) (void)hipStreamDestroy(l->stream2);
  if (l->h_send) (void)hipHostFree(l->h_send);
  if (l->h_recv) (void)hipHostFree(l->h_recv);
  delete l;
}

// One BFS level on every rank (collective).  global = the level's figures summed / maximised over the ranks (viol_fp = the smallest
// violating fingerprint of any rank, ~0 if none); local = this rank's own.  global->n_new == 0: the search is exhausted.
int32_t vsrmc_shard_loop_step(vsrmc_shard_loop* l, vsrmc_level_info* global, vsrmc_level_info* local) {
  if (!l || !global || !local) return fail(VSRMC_E_ARG, "NULL argument");
  vsrmc_checker* c = l->c;
  const int w = l->world, me = l->rank;
  std::memset(global, 0, sizeof(*global));
  std::memset(local, 0, sizeof(*local));
  if (l->replicated) {                                           // every rank explores the small early levels on its own: no collective
    int rc = vsrmc_shard_local_step(c, local);
    if (rc) return rc;
    vsrmc_shard_set_max_bag(c, local->max_bag);
    *global = *local;
    l->level += 1;
    l->n_frontier = local->n_new;
    l->distinct += local->n_new;
    global->distinct = l->distinct;
    if (local->viol_mask && !l->has_violation) {
      l->has_violation = true; l->viol_fp = local->viol_fp; l->viol_mask = local->viol_mask; l->viol_level = l->level;
    } else if (local->n_new >= l->replicate_below) {
      u64 kept = 0;
      rc = vsrmc_shard_partition(c, &kept);
      if (rc) return rc;
      l->replicated = false;
    }
    return 0;
  }
  // ---- 0. in slices with the exchange overlapped (round 6)?  Every rank must cut its frontier into the SAME number of slices: the largest any rank wants —
  // enough that a slice's announcements fill at most 0.45 of an owner's bucket, at least VSRMC_OVERLAP_MIN_SLICES (default 2) once a rank holds 2^20 states
  // (below that a level is a millisecond of kernel: nothing to hide an exchange behind), at most 8.  VSRMC_OVERLAP=0: the sequential level of rounds 2-5.
  vsrmc_shard_io io;
  io.cand_send = l->cand_send;
  io.cand_cap = l->cand_cap;
  int rc = 0, crc = 0;
  u64 worst = 0;
  bool head_done = false;
  static const bool overlap_on = !(std::getenv("VSRMC_OVERLAP") && std::atoi(std::getenv("VSRMC_OVERLAP")) == 0);
  if (w > 1 && overlap_on) {
    static const u64 min_slices = std::getenv("VSRMC_OVERLAP_MIN_SLICES") ? (u64)std::max(1, std::atoi(std::getenv("VSRMC_OVERLAP_MIN_SLICES"))) : 2;
    static const u64 min_states = std::getenv("VSRMC_OVERLAP_MIN_STATES") ? (u64)std::atoll(std::getenv("VSRMC_OVERLAP_MIN_STATES")) : ((u64)1 << 20);
    const u64 per_slice = std::max<u64>(4096, (u64)(0.45 * (double)(l->cand_cap / 2) * (double)w / (double)std::max<u64>(2, c->g_last)));   // (as deep_cand_bound, for HALF a bucket: 1 / w of a parent's successors go to each owner)
    u64 want = c->n_frontier >= min_states ? std::max<u64>(min_slices, (c->n_frontier + per_slice - 1) / per_slice) : 1;
    want = std::min<u64>(want, 8);                              // (every slice leaves a partly used index and word chunk per block behind: next_level_fits has margin for about that many)
    u64 all_want[8] = {0};
    crc = loop_allgather(l, &want, all_want, 8);
    if (crc) return crc;
    for (int p = 0; p < w; p++) want = std::max(want, all_want[p]);
    if (want >= 2) {
      crc = loop_level_head_overlapped(l, (int)want, &rc);
      if (crc) return crc;
      head_done = true;
    }
  }
  if (!head_done) {
  // ---- 1. expand; candidates owned by other ranks are bucketed per owner
  uint64_t counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  rc = vsrmc_shard_expand(c, &io, counts);
  struct CountRow { u64 cnt[8]; u64 err; } mine, zero;
  std::memset(&zero, 0, sizeof(zero));
  mine = zero;
  for (int p = 0; p < w; p++) mine.cnt[p] = (p == me || rc) ? 0 : std::min<u64>(counts[p], l->cand_cap);
  mine.err = rc ? (u64)(rc < 0 ? -rc : rc) : 0;
  std::vector<CountRow> all(w, zero);
  crc = loop_allgather(l, &mine, all.data(), (u32)sizeof(CountRow));
  if (crc) return crc;
  worst = 0;
  for (const CountRow& r : all) worst = std::max(worst, r.err);
  if (worst) return rc ? rc : fail(VSRMC_E_STATE, "level " + std::to_string(l->level + 1) + ", phase expand: error " + std::to_string((long long)worst) + " on another rank");
  // ---- 2. candidates to their owners
  u64 scnt[8], soff[8], rcnt[8], roff[8], n_recv = 0;
  for (int p = 0; p < w; p++) {
    scnt[p] = mine.cnt[p];
    soff[p] = (u64)p * l->cand_cap;
    rcnt[p] = p == me ? 0 : all[p].cnt[me];
    roff[p] = n_recv;
    n_recv += rcnt[p];
  }
  if (n_recv > (u64)w * l->cand_cap) return fail(VSRMC_E_REP, "more candidates received than the exchange buffer holds");
  rc = loop_alltoallv(l, l->cand_send, scnt, soff, l->cand_recv, rcnt, roff, 16);
  // ---- 3. claim them in this rank's shard; 4. verdict bytes back to the generators
  // (a failed claim still answers: the sizes of the verdict exchange are fixed by the candidate counts, so no rank is left waiting; the
  // error code rides on the next all-gather)
  if (!rc) rc = vsrmc_shard_claim(c, l->cand_recv, n_recv, l->verdict_out);
  {
    const int xrc = loop_alltoallv(l, l->verdict_out, rcnt, roff, l->verdict_in, scnt, soff, 1);
    if (!rc) rc = xrc;
  }
  // ---- 5. withdraw the announced successors that lost
  if (!rc) rc = vsrmc_shard_materialize(c, &io, l->verdict_in);
  }   // (!head_done)
  // ---- 6. compare the frontier sizes; move records where they are missing (rare)
  u64 valid = 0, range = 0;
  if (!rc) rc = vsrmc_shard_count(c, &valid, &range);
  struct BalRow { u64 valid, err; } bmine = {valid, rc ? (u64)(rc < 0 ? -rc : rc) : 0};
  std::vector<BalRow> ball(w);
  crc = loop_allgather(l, &bmine, ball.data(), (u32)sizeof(BalRow));
  if (crc) return crc;
  worst = 0;
  std::vector<u64> sizes(w);
  for (int p = 0; p < w; p++) { worst = std::max(worst, ball[p].err); sizes[p] = ball[p].valid; }
  if (worst) return rc ? rc : fail(VSRMC_E_STATE, "level " + std::to_string(l->level + 1) + ", phase materialize: error " + std::to_string((long long)worst) + " on another rank");
  const std::vector<Move> plan = balance_plan(sizes);
  if (!plan.empty()) {
    if (!l->mv_words) {
      hipError_t e = hipMalloc((void**)&l->mv_words, l->rec_words_cap * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&l->mv_off, l->rec_cap * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&l->mv_fp, l->rec_cap * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&l->rv_words, l->rec_words_cap * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&l->rv_off, l->rec_cap * 8);
      if (e == hipSuccess) e = hipMalloc((void**)&l->rv_fp, l->rec_cap * 8);
      if (e != hipSuccess) rc = fail(VSRMC_E_HIP, "hipMalloc of the rebalancing buffers failed");
    }
    // this rank's exports, packed one after the other: (records, words) per destination
    struct MvRow { u64 n[8], nw[8], err; } mrow;
    std::memset(&mrow, 0, sizeof(mrow));
    u64 on = 0, ow = 0, hi = range;
    u64 so_n[8] = {0}, so_w[8] = {0};
    for (const Move& mv : plan) {
      if (mv.src != me || rc) continue;
      u64 width = std::min<u64>(hi, (mv.k * range + std::max<u64>(1, valid) - 1) / std::max<u64>(1, valid));   // index window holding about k valid records
      // all of this rank's moves share one export buffer: a window never holds more records than indices, nor more words than indices x
      // the longest record — clamp it to what is left, so that a large imbalance is exported in part (the next level moves the rest)
      // instead of failing in k_export
      if (on >= l->rec_cap || ow >= l->rec_words_cap) break;
      width = std::min<u64>(width, std::min<u64>(l->rec_cap - on, (l->rec_words_cap - ow) / (u64)std::max(1, c->lds_stride)));
      if (width == 0) break;
      u64 got = 0, gotw = 0;
      rc = vsrmc_shard_export(c, hi - width, width, l->mv_words + ow, l->rec_words_cap - ow, l->mv_off + on, l->mv_fp + on, l->rec_cap - on, &got, &gotw);
      if (rc) break;
      hi -= width;
      so_n[mv.dst] = on; so_w[mv.dst] = ow;
      mrow.n[mv.dst] = got; mrow.nw[mv.dst] = gotw;
      on += got; ow += gotw;
    }
    mrow.err = rc ? (u64)(rc < 0 ? -rc : rc) : 0;
    std::vector<MvRow> mall(w);
    crc = loop_allgather(l, &mrow, mall.data(), (u32)sizeof(MvRow));
    if (crc) return crc;
    worst = 0;
    for (const MvRow& r : mall) worst = std::max(worst, r.err);
    if (worst) return rc ? rc : fail(VSRMC_E_STATE, "level " + std::to_string(l->level + 1) + ", phase rebalance: error " + std::to_string((long long)worst) + " on another rank");
    u64 rn[8], rnw[8], ro_n[8], ro_w[8], tn = 0, tw = 0;
    for (int p = 0; p < w; p++) {
      rn[p] = p == me ? 0 : mall[p].n[me];
      rnw[p] = p == me ? 0 : mall[p].nw[me];
      ro_n[p] = tn; ro_w[p] = tw;
      tn += rn[p]; tw += rnw[p];
    }
    // every rank holds `mall`: the receive totals of EVERY rank are checked by every rank, so that all of them skip the three
    // exchanges together (a rank that found out alone would leave its peers waiting in a collective addressed to it)
    for (int q = 0; q < w && !rc; q++) {
      u64 qn = 0, qw = 0;
      for (int p2 = 0; p2 < w; p2++)
        if (p2 != q) { qn += mall[p2].n[q]; qw += mall[p2].nw[q]; }
      if (qn > l->rec_cap || qw > l->rec_words_cap)
        rc = fail(VSRMC_E_REP, "rank " + std::to_string(q) + " would receive more records than its rebalancing buffers hold (rec_cap / rec_words_cap)");
    }
    if (!rc) rc = loop_alltoallv(l, l->mv_words, mrow.nw, so_w, l->rv_words, rnw, ro_w, 8);
    if (!rc) rc = loop_alltoallv(l, l->mv_off, mrow.n, so_n, l->rv_off, rn, ro_n, 8);
    if (!rc) rc = loop_alltoallv(l, l->mv_fp, mrow.n, so_n, l->rv_fp, rn, ro_n, 8);
    for (int p = 0; p < w && !rc; p++)
      if (p != me && rn[p]) {
        rc = vsrmc_shard_append(c, l->rv_words + ro_w[p], rnw[p], l->rv_off + ro_n[p], l->rv_fp + ro_n[p], rn[p]);
        l->moved += rn[p];
      }
  }
  // ---- 7. commit; one all-gather carries every figure of the level
  if (!rc) rc = vsrmc_shard_commit(c, local);
  struct InfoRow { u64 n_new, generated, deadlocks, pending, viol_fp, err, viol_mask, max_bag, probes, record_words, act[16]; } irow;
  for (int i = 0; i < 16; i++) irow.act[i] = local->act_generated[i];
  irow.record_words = local->record_words;
  irow.n_new = local->n_new; irow.generated = local->generated; irow.deadlocks = local->deadlocks; irow.pending = local->pending;
  irow.viol_fp = local->viol_mask ? local->viol_fp : ~(u64)0; irow.err = rc ? (u64)(rc < 0 ? -rc : rc) : 0;
  irow.viol_mask = (u64)local->viol_mask; irow.max_bag = local->max_bag; irow.probes = local->probes;
  std::vector<InfoRow> iall(w);
  crc = loop_allgather(l, &irow, iall.data(), (u32)sizeof(InfoRow));
  if (crc) return crc;
  worst = 0;
  u64 gviol = ~(u64)0, gbag = 0;
  for (const InfoRow& r : iall) {
    worst = std::max(worst, r.err);
    global->n_new += r.n_new; global->generated += r.generated; global->deadlocks += r.deadlocks; global->pending += r.pending; global->probes += r.probes;
    global->record_words += r.record_words;
    for (int i = 0; i < 16; i++) global->act_generated[i] += r.act[i];
    gviol = std::min(gviol, r.viol_fp);
    gbag = std::max(gbag, r.max_bag);
  }
  if (worst) return rc ? rc : fail(VSRMC_E_STATE, "level " + std::to_string(l->level + 1) + ", phase commit: error " + std::to_string((long long)worst) + " on another rank");
  vsrmc_shard_set_max_bag(c, gbag);
  l->level += 1;
  l->n_frontier = global->n_new;
  l->distinct += global->n_new;
  global->level = l->level;
  global->distinct = l->distinct;
  global->max_bag = gbag;
  global->viol_fp = gviol;
  global->frontier = local->frontier;
  global->expand_ms = local->expand_ms;
  global->materialize_ms = local->materialize_ms;
  if (gviol != ~(u64)0) {
    for (const InfoRow& r : iall)
      if (r.viol_fp == gviol) global->viol_mask |= (int32_t)r.viol_mask;
    if (!l->has_violation) { l->has_violation = true; l->viol_fp = gviol; l->viol_mask = global->viol_mask; l->viol_level = l->level; }
  }
  return 0;
}

// vsrmc_check for a sharded run (collective): stop_reason 0 exhausted, 1 invariant violated, 2 max_depth
int32_t vsrmc_shard_loop_advance(vsrmc_shard_loop* l, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what);
// The seen-sets before the next unit of progress (collective once the run is sharded): *state = 0 room on every rank, 2 = some rank's shard is more than
// 85 % full — every rank learns it in the same call, so that all of them stop together with "incomplete at depth N" instead of one of them running
// into ERR_TABLE_FULL inside a collective.  (A shard does not grow: the ranks of a node share no spare memory to re-hash into.)
int32_t vsrmc_shard_loop_room(vsrmc_shard_loop* l, int32_t* state) {
  if (!l || !state) return fail(VSRMC_E_ARG, "NULL argument");
  u64 full = (double)(l->c->deep ? l->c->deep_distinct : l->c->distinct) > 0.85 * (double)(l->c->tmask + 1) ? 1 : 0;
  std::string why;                                              // (what vsrmc_last_error() says after an answer of 2: which set, how full)
  if (full) why = "seen-set shard: " + std::to_string((unsigned long long)(l->c->deep ? l->c->deep_distinct : l->c->distinct)) + " states in " +
                  std::to_string((unsigned long long)(l->c->tmask + 1)) + " slots";
  // the winner set of the deep search (round-5 advice): it fills with the rank's share of every level beyond the base, long before the seen-set shard does —
  // grown between two passes while the device has the memory, else the run stops here, cleanly, instead of ERR_TABLE_FULL inside a pass
  if (l->c->d_wset && l->c->deep) {
    const u64 held = wset_count(l->c);
    // the next level's share: this rank's share of the newest level times the last growth factor (these models' levels grow by less from level to level)
    const size_t nd = l->c->deep_lv.size();
    const double grow = nd >= 2 && l->c->deep_lv[nd - 2].n_local ? std::min(2.0, (double)l->c->deep_lv[nd - 1].n_local / (double)l->c->deep_lv[nd - 2].n_local) : 2.0;
    const u64 next = nd ? (u64)(grow * (double)l->c->deep_lv.back().n_local) : 0;
    while ((double)(held + next) > 0.6 * (double)(l->c->h_wset.mask + 1) && wset_grow(l->c) == 0) {}
    // (linear probing, 65 536 steps at most: a set that cannot grow is used as long as the next level can fit at all — the README run at world 2 with both
    // ranks on ONE GPU ends at 0.91 of rank 1's 2^30 slots (the ranks' shares differ: 533 M against 479 M states before the last level), and the estimate
    // of the next level — the last growth factor again — overshoots, these models' levels grow by less every level.  A set that does run full is
    // ERR_TABLE_FULL inside the pass, an error and never a wrong count)
    if ((double)(held + next) > 0.98 * (double)(l->c->h_wset.mask + 1)) {
      full = 1;
      why = "winner set: " + std::to_string((unsigned long long)held) + " states held + " + std::to_string((unsigned long long)next) + " expected of the next level in " +
            std::to_string((unsigned long long)(l->c->h_wset.mask + 1)) + " slots, and the device has no memory for twice as many";
    }
  }
  if (!l->replicated) {
    u64 all[8] = {0};
    const int rc0 = l->comm.allgather(l->comm.ctx, &full, all, 8);
    if (rc0) return rc0 > 0 ? fail(VSRMC_E_HIP, "the caller's all-gather failed") : rc0;
    for (int p = 0; p < l->world; p++) full = std::max(full, all[p]);
  }
  *state = full ? 2 : 0;
  if (full) (void)fail(0, why.empty() ? std::string("another rank's seen-set shard or winner set is full") : "rank " + std::to_string(l->rank) + ": " + why);
  return 0;
}

int32_t vsrmc_shard_loop_run(vsrmc_shard_loop* l, int32_t max_depth, int32_t stop_on_violation, int32_t* stop_reason, vsrmc_level_info* last) {
  if (!l || !stop_reason || !last) return fail(VSRMC_E_ARG, "NULL argument");
  vsrmc_level_info b;
  for (;;) {
    if (max_depth > 0 && l->level + l->deep >= max_depth) { *stop_reason = 2; return 0; }
    // a seen-set that fills up: every rank asks, any rank's answer stops all (the all-gather inside advance needs every rank)
    int32_t room = 0;
    const int rc0 = vsrmc_shard_loop_room(l, &room);
    if (rc0) return rc0;
    if (room == 2) { *stop_reason = 4; return 0; }
    int32_t what = 0;
    const int rc = vsrmc_shard_loop_advance(l, last, &b, &what);
    if (rc) return rc;
    if (last->n_new == 0) { *stop_reason = 0; return 0; }
    if (what == 2 && b.level && b.viol_mask && !last->viol_mask) *last = b;
    if (l->has_violation && stop_on_violation) { *stop_reason = 1; return 0; }
  }
}

int32_t vsrmc_shard_loop_status(vsrmc_shard_loop* l, int32_t* level, uint64_t* distinct, uint64_t* n_frontier, int32_t* replicated, uint64_t* viol_fp,
                                int32_t* viol_mask, int32_t* viol_level, uint64_t* moved, uint64_t* bytes_sent) {
  if (!l) return fail(VSRMC_E_ARG, "NULL argument");
  if (level) *level = l->level;
  if (distinct) *distinct = l->distinct;
  if (n_frontier) *n_frontier = l->n_frontier;
  if (replicated) *replicated = l->replicated ? 1 : 0;
  if (viol_fp) *viol_fp = l->has_violation ? l->viol_fp : ~(u64)0;
  if (viol_mask) *viol_mask = l->has_violation ? l->viol_mask : 0;
  if (viol_level) *viol_level = l->viol_level;
  if (moved) *moved = l->moved;
  if (bytes_sent) *bytes_sent = l->bytes_sent;
  return 0;
}

// how many sharded levels ran in slices with the exchange overlapped (vsrmc_shard_loop_step), and how many slices in all
int32_t vsrmc_shard_loop_overlap_stats(vsrmc_shard_loop* l, uint64_t* levels, uint64_t* slices) {
  if (!l || !levels || !slices) return fail(VSRMC_E_ARG, "NULL argument");
  *levels = l->overlap_levels;
  *slices = l->overlap_slices;
  return 0;
}

// Walk the predecessor pointers — they live in the seen-set slots, i.e. on the owner of each state — from the level-`level` state with
// fingerprint `fp` back to Init (collective).  fps[0 .. level) = the fingerprints of the path, Init first.  Two small all-gathers per level;
// whole (matches, fingerprint, meta) rows are gathered: identical answers (the replicated early levels live in every table) are one
// answer, different ones are an ambiguous pointer and are reported.
int32_t vsrmc_shard_loop_trace_fps(vsrmc_shard_loop* l, int32_t level, uint64_t fp, uint64_t* fps) {
  if (!l || !fps || level < 1) return fail(VSRMC_E_ARG, "bad trace arguments");
  struct Row { u64 n, fp, meta, err; };
  auto agree = [&](u64 key, int lvl, int by_low, Row* out) -> int {
    int32_t found = 0;
    u64 f = 0, m = 0;
    int rc = vsrmc_checker_lookup(l->c, key, lvl, by_low, &found, &f, &m);
    Row mine = {rc ? 0 : (u64)found, f, m, rc ? (u64)(rc < 0 ? -rc : rc) : 0};
    std::vector<Row> all(l->world);
    const int crc = loop_allgather(l, &mine, all.data(), (u32)sizeof(Row));
    if (crc) return crc;
    for (const Row& r : all)                                    // a failed lookup on any rank is that error on every rank, not "not found"
      if (r.err) return rc ? rc : fail(VSRMC_E_STATE, "trace walk: the seen-set lookup failed on another rank (error " + std::to_string((long long)r.err) + ")");
    out->n = 0;
    for (const Row& r : all) {
      if (!r.n) continue;
      if (r.n > 1 || (out->n && r.fp != out->fp))
        return fail(VSRMC_E_STATE, "trace walk: ambiguous predecessor pointer — several states match the 45 fingerprint bits of one parent over the ranks");
      if (!out->n || r.meta < out->meta) { out->fp = r.fp; out->meta = r.meta; }
      out->n = 1;
    }
    return 0;
  };
  fps[level - 1] = fp;
  for (int lv = level; lv > 1; lv--) {
    Row hit, parent;
    int rc = agree(fp, lv, 0, &hit);
    if (rc) return rc;
    if (!hit.n || (int)meta_level(hit.meta) != lv) return fail(VSRMC_E_STATE, "trace walk: no state with this fingerprint at this level in any shard");
    rc = agree(meta_pfp(hit.meta), lv - 1, 1, &parent);
    if (rc) return rc;
    if (!parent.n) return fail(VSRMC_E_STATE, "trace walk: the parent of a state of the path is in no shard");
    fp = parent.fp;
    fps[lv - 2] = fp;
  }
  return 0;
}

}  // extern "C"

// ================================================================================================================================
// Levels beyond the record buffers on a sharded run (vsr_deep.hpp with the collective hooks).  A pass that INSERTS a level is the protocol of a
// sharded level with other sources / targets: k_expand over a slice announces the successors other ranks own through the sent-filter (virtual
// level: nothing written; inserted level: written speculatively into the scratch buffer), the owners claim (k_claim_batch_fused: first inserter
// wins), the verdict bytes come back, and the winners are counted (k_count_verdict) or kept (k_apply_verdict withdraws the rest) — and remembered
// in the rank's winner set (vsr_kernels.hpp: WSet) together with the states the rank's own lanes inserted.  A pass that REGENERATES a level is
// local: the rank rebuilds what its winner set holds, once per descent, and nothing crosses the fabric (rounds 3-4 asked the owners about every
// candidate of every regenerating pass).
// Loops run as long as ANY rank has a slice left; a rank that has run out takes part in the others' exchanges with empty buckets.
// ================================================================================================================================
namespace {

int loop_deep_any(void* ctx, u64* flag) {
  vsrmc_shard_loop* l = (vsrmc_shard_loop*)ctx;
  u64 all[8] = {0};
  const int rc = loop_allgather(l, flag, all, 8);
  if (rc) return rc;
  u64 m = 0;
  for (int p = 0; p < l->world; p++) m = std::max(m, all[p]);
  *flag = m;
  return 0;
}

int loop_deep_pass(void* ctx, const u64* sw, const u64* so, u64 n, u64 p_off, int level, int mode, u64 bag, const PassDst* dst) {
  vsrmc_shard_loop* l = (vsrmc_shard_loop*)ctx;
  vsrmc_checker* c = l->c;
  const int w = l->world, me = l->rank;
  vsrmc_shard_io io;
  io.cand_send = l->cand_send;
  io.cand_cap = l->cand_cap;
  int rc = wset_ensure(c);                                     // the rank's winner set (vsr_kernels.hpp: WSet), allocated at the first deep pass (world > 1)
  if (!rc && c->d_wset) c->wset_used = true;
  if (!rc) rc = expand_pass(c, sw, so, n, p_off, level, mode, bag, dst, &io);
  if (mode == MODE_REGEN) {
    // A regenerated level costs NO exchange (round 5): every rank rebuilds the states its own candidates inserted — it has kept their fingerprints
    // since the verdict of the pass that inserted them — and k_expand has written them already.  The ranks only tell each other that they are done
    // (nobody goes on alone).  Rounds 3-4: every candidate to its owner, a verdict byte back, k_materialize for the winners.
    u64 e0 = rc ? (u64)(rc < 0 ? -rc : rc) : 0, eall0[8] = {0};
    const int crc0 = loop_allgather(l, &e0, eall0, 8);
    if (crc0) return crc0;
    for (int p = 0; p < w; p++)
      if (eall0[p] && !rc) rc = fail(VSRMC_E_STATE, "deep pass (level " + std::to_string(level) + "), regeneration: error " + std::to_string((long long)eall0[p]) + " on another rank");
    return rc;
  }
  struct CountRow { u64 cnt[8]; u64 err; } mine;
  std::memset(&mine, 0, sizeof(mine));
  for (int p = 0; p < w; p++) mine.cnt[p] = (p == me || rc) ? 0 : std::min<u64>(c->h.cand_cnt[p], l->cand_cap);
  mine.err = rc ? (u64)(rc < 0 ? -rc : rc) : 0;
  std::vector<CountRow> all((size_t)w);
  int crc = loop_allgather(l, &mine, all.data(), (u32)sizeof(CountRow));
  if (crc) return crc;
  u64 worst = 0;
  for (const CountRow& r : all) worst = std::max(worst, r.err);
  if (worst) return rc ? rc : fail(VSRMC_E_STATE, "deep pass (level " + std::to_string(level) + "), phase expand: error " + std::to_string((long long)worst) + " on another rank");
  u64 scnt[8], soff[8], rcnt[8], roff[8], n_recv = 0;
  for (int p = 0; p < w; p++) {
    scnt[p] = mine.cnt[p];
    soff[p] = (u64)p * l->cand_cap;
    rcnt[p] = p == me ? 0 : all[(size_t)p].cnt[me];
    roff[p] = n_recv;
    n_recv += rcnt[p];
  }
  rc = loop_alltoallv(l, l->cand_send, scnt, soff, l->cand_recv, rcnt, roff, 16);
  // the owner's side: claim (virtual / inserted level) or grant the regeneration (the candidate that carries the slot's final key)
  if (!rc && n_recv) {
    const unsigned grid = (unsigned)((n_recv + 255) / 256);
    hipLaunchKernelGGL(k_claim_batch_fused, dim3(grid), dim3(256), 0, c->stream, c->table, c->tmask, l->cand_recv, n_recv, level, l->verdict_out, c->ctl);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(VSRMC_E_HIP, "deep pass: claim kernel");
  }
  {
    const int xrc = loop_alltoallv(l, l->verdict_out, rcnt, roff, l->verdict_in, scnt, soff, 1);
    if (!rc) rc = xrc;
  }
  // the generator's side
  for (int o = 0; o < w && !rc; o++) {
    if (o == me || scnt[o] == 0) continue;
    const u64 k = scnt[o];
    const u64* ent = l->cand_send + 2 * (u64)o * l->cand_cap;
    const u64* cidx = c->cand_idx + (u64)o * l->cand_cap;
    const uint8_t* ver = l->verdict_in + (u64)o * l->cand_cap;
    // the winners are counted (virtual level) or kept (inserted level) — and REMEMBERED in the rank's winner set: they are what it regenerates later
    if (mode == MODE_INSERT)
      hipLaunchKernelGGL(k_count_verdict, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, ent, cidx, ver, k, c->pending, c->opt.pending_entries, c->ctl, (const WSet*)c->d_wset, level);
    else
      hipLaunchKernelGGL(k_apply_verdict, dim3((unsigned)((k + 255) / 256)), dim3(256), 0, c->stream, ent, cidx, ver, k, dst->off, dst->fp, c->ctl, (const WSet*)c->d_wset, level);
    if (hipGetLastError() != hipSuccess) rc = fail(VSRMC_E_HIP, "deep pass: verdict kernel");
  }
  if (!rc && (hipMemcpyAsync(&c->h, c->ctl, sizeof(c->h), hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess))
    rc = fail(VSRMC_E_HIP, "deep pass: control block");
  if (!rc && c->h.err) rc = level_error(c, c->h, level);
  if (!rc && c->h.ties) rc = fail(VSRMC_E_STATE, "two successors of one level share a VIEW fingerprint but differ in the aux variables (SURVEY F2)");
  // nobody goes on alone: the error codes of the claim / apply half
  u64 e1 = rc ? (u64)(rc < 0 ? -rc : rc) : 0, eall[8] = {0};
  crc = loop_allgather(l, &e1, eall, 8);
  if (crc) return crc;
  for (int p = 0; p < w; p++)
    if (eall[p] && !rc) rc = fail(VSRMC_E_STATE, "deep pass (level " + std::to_string(level) + "), phase claim: error " + std::to_string((long long)eall[p]) + " on another rank");
  return rc;
}

// this rank's figures of the pass -> the level's (sums; largest bag; smallest violating fingerprint; masks or-ed; checksums xor / sum)
int loop_deep_reduce(void* ctx, DeepRun* R) {
  vsrmc_shard_loop* l = (vsrmc_shard_loop*)ctx;
  struct Row { u64 n_new, generated, deadlocks, probes, words, fx, fs, bag, viol, mask, frontier, act[16], p_gen, p_dead, p_probes, p_mask, p_lim, p_act[16]; };
  static_assert(sizeof(Row) <= 512, "one all-gather record");
  Row mine;
  std::memset(&mine, 0, sizeof(mine));
  const vsrmc_level_info& a = R->ins;
  const vsrmc_level_info& b = R->prb;
  mine.n_new = a.n_new; mine.generated = a.generated; mine.deadlocks = a.deadlocks; mine.probes = a.probes; mine.words = a.record_words;
  mine.fx = a.fp_xor; mine.fs = a.fp_sum; mine.bag = a.max_bag; mine.viol = a.viol_fp; mine.mask = (u64)(u32)a.viol_mask; mine.frontier = a.frontier;
  for (int i = 0; i < 16; i++) { mine.act[i] = a.act_generated[i]; mine.p_act[i] = b.act_generated[i]; }
  mine.p_gen = b.generated; mine.p_dead = b.deadlocks; mine.p_probes = b.probes; mine.p_mask = (u64)(u32)b.viol_mask; mine.p_lim = b.limit_rechecked;
  std::vector<Row> all((size_t)l->world);
  const int rc = loop_allgather(l, &mine, all.data(), (u32)sizeof(Row));
  if (rc) return rc;
  Row t;
  std::memset(&t, 0, sizeof(t));
  t.viol = ~(u64)0;
  for (const Row& r : all) {
    t.n_new += r.n_new; t.generated += r.generated; t.deadlocks += r.deadlocks; t.probes += r.probes; t.words += r.words;
    t.fx ^= r.fx; t.fs += r.fs; t.bag = std::max(t.bag, r.bag); t.frontier += r.frontier;
    if (r.viol < t.viol) t.viol = r.viol;
    t.p_gen += r.p_gen; t.p_dead += r.p_dead; t.p_probes += r.p_probes; t.p_mask |= r.p_mask; t.p_lim += r.p_lim;
    for (int i = 0; i < 16; i++) { t.act[i] += r.act[i]; t.p_act[i] += r.p_act[i]; }
  }
  for (const Row& r : all)
    if (t.viol != ~(u64)0 && r.viol == t.viol) t.mask |= r.mask;
  vsrmc_level_info& A = R->ins;
  vsrmc_level_info& B = R->prb;
  A.n_new = t.n_new; A.generated = t.generated; A.deadlocks = t.deadlocks; A.probes = t.probes; A.record_words = t.words;
  A.fp_xor = t.fx; A.fp_sum = t.fs; A.max_bag = t.bag; A.viol_fp = t.viol; A.viol_mask = (int32_t)t.mask; A.frontier = t.frontier;
  B.generated = t.p_gen; B.deadlocks = t.p_dead; B.probes = t.p_probes; B.viol_mask = (int32_t)t.p_mask; B.limit_rechecked = t.p_lim;
  for (int i = 0; i < 16; i++) { A.act_generated[i] = t.act[i]; B.act_generated[i] = t.p_act[i]; }
  R->viol_ins = t.viol; R->mask_ins = (u32)t.mask; R->mask_prb = (u32)t.p_mask;
  return 0;
}

// Every rank holds the violating successors ITS probe passes could not find in its own shard.  A rank can only vouch for fingerprints it
// owns, so every (fingerprint, smallest key) pair is shown to all ranks, 30 per round; the owner drops what it has seen at a level below
// the probed one; the smallest surviving fingerprint is the violation, the smallest key among its copies names its parent.
int loop_deep_resolve(void* ctx, DeepRun* R) {
  vsrmc_shard_loop* l = (vsrmc_shard_loop*)ctx;
  vsrmc_checker* c = l->c;
  const int w = l->world, plevel = R->probed_level();
  std::vector<std::pair<u64, u64>> pairs;
  for (size_t i = 0; i + 1 < R->bad.size(); i += 2) pairs.push_back(std::make_pair(R->bad[i], R->bad[i + 1]));
  std::sort(pairs.begin(), pairs.end());
  std::vector<std::pair<u64, u64>> uniq;                       // one entry per fingerprint: its smallest key
  for (const auto& pr : pairs)
    if (uniq.empty() || uniq.back().first != pr.first) uniq.push_back(pr);
  struct Round { u64 n; u64 fp[30], key[30]; };
  static_assert(sizeof(Round) <= 512, "one all-gather record");
  u64 best_fp = ~(u64)0, best_key = ~(u64)0, seen_total = 0;
  for (size_t at = 0;; at += 30) {
    u64 more = at < uniq.size() ? 1 : 0;
    int rc = loop_deep_any(l, &more);
    if (rc) return rc;
    if (!more) break;
    Round mine;
    std::memset(&mine, 0, sizeof(mine));
    for (size_t i = at; i < uniq.size() && i < at + 30; i++) { mine.fp[mine.n] = uniq[i].first; mine.key[mine.n] = uniq[i].second; mine.n++; }
    std::vector<Round> all((size_t)w);
    rc = loop_allgather(l, &mine, all.data(), (u32)sizeof(Round));
    if (rc) return rc;
    std::vector<u64> fps, keys;
    for (const Round& r : all)
      for (u64 i = 0; i < r.n; i++)
        if (owner_of(r.fp[i], w) == l->rank) { fps.push_back(r.fp[i]); keys.push_back(r.key[i]); }
    std::vector<uint8_t> seen(fps.size(), 0);
    rc = vsrmc_checker_seen_batch(c, fps.data(), (u64)fps.size(), plevel, seen.data());
    u64 e1 = rc ? 1 : 0;
    int crc = loop_deep_any(l, &e1);
    if (crc) return crc;
    if (e1) return rc ? rc : fail(VSRMC_E_STATE, "deep pass: the seen-set lookup of the probe's candidates failed on another rank");
    for (size_t i = 0; i < fps.size(); i++) {
      if (seen[i]) continue;
      seen_total++;
      if (fps[i] < best_fp || (fps[i] == best_fp && keys[i] < best_key)) { best_fp = fps[i]; best_key = keys[i]; }
    }
  }
  struct Best { u64 fp, key, cnt; } bm = {best_fp, best_key, seen_total};
  std::vector<Best> ball((size_t)w);
  int rc = loop_allgather(l, &bm, ball.data(), (u32)sizeof(Best));
  if (rc) return rc;
  Best g = {~(u64)0, ~(u64)0, 0};
  for (const Best& b : ball) {
    g.cnt += b.cnt;
    if (b.fp < g.fp || (b.fp == g.fp && b.key < g.key)) { g.fp = b.fp; g.key = b.key; }
  }
  R->prb.pending = g.cnt;
  if (g.fp == ~(u64)0) return 0;
  R->prb.viol_fp = g.fp;
  R->prb.viol_mask = (int32_t)R->mask_prb;
  l->probe_parent_fp = 0;
  // its parent: the state of the level below whose fingerprint ends in the key's 45 bits, in whichever shard (collective lookup)
  struct Row { u64 n, fp, meta, err; };
  int32_t found = 0;
  u64 f = 0, m = 0;
  rc = vsrmc_checker_lookup(c, meta_pfp(g.key), plevel - 1, 1, &found, &f, &m);
  Row mine = {rc ? 0 : (u64)found, f, m, rc ? (u64)1 : 0};
  std::vector<Row> rows((size_t)w);
  const int crc = loop_allgather(l, &mine, rows.data(), (u32)sizeof(Row));
  if (crc) return crc;
  u64 hits = 0, pfp = 0;
  for (const Row& r : rows) {
    if (r.err) return rc ? rc : fail(VSRMC_E_STATE, "deep pass: the parent lookup failed on another rank");
    if (!r.n) continue;
    if (r.n > 1 || (hits && r.fp != pfp)) return fail(VSRMC_E_STATE, "ambiguous predecessor pointer: several states of the parent's level share the 45 fingerprint bits the violating successor keeps of its parent");
    hits = 1; pfp = r.fp;
  }
  if (!hits) return fail(VSRMC_E_STATE, "deep pass: the parent of the violating successor is in no shard");
  l->probe_parent_fp = pfp;
  return 0;
}

DeepIo loop_deep_io(vsrmc_shard_loop* l) {
  DeepIo io;
  io.ctx = l; io.world = l->world; io.cand_cap = l->cand_cap;
  io.pass = loop_deep_pass; io.any = loop_deep_any; io.reduce = loop_deep_reduce; io.resolve = loop_deep_resolve;
  return io;
}

}  // namespace

extern "C" {

// vsrmc_checker_deepen for a sharded run (collective): one more level beyond the ranks' record buffers.  inserted / probed = the level's
// figures over all ranks.  A violation found by the probe: vsrmc_shard_loop_probe_trace_fps walks from its parent.
int32_t vsrmc_shard_loop_deepen(vsrmc_shard_loop* l, vsrmc_level_info* inserted, vsrmc_level_info* probed) {
  if (!l || !inserted || !probed) return fail(VSRMC_E_ARG, "NULL argument");
  vsrmc_checker* c = l->c;
  int rc = deep_check_ready(c, 2, true);
  // (every rank reaches the same verdict here: level, deep and failed move in lock-step; a rank that failed alone has reported it in a collective)
  if (rc) return rc;
  if (l->replicated) return fail(VSRMC_E_STATE, "the replicated phase of a sharded run stores its levels (they are small by definition)");
  HIPCHK(hipSetDevice(c->opt.device));
  std::memset(inserted, 0, sizeof(*inserted));
  std::memset(probed, 0, sizeof(*probed));
  inserted->viol_fp = inserted->viol_index = probed->viol_fp = probed->viol_index = ~(u64)0;
  const DeepIo io = loop_deep_io(l);
  const u64 distinct_before = l->distinct;
  if (c->deep == 0) rc = deep_first_pass(c, inserted, &io);
  else rc = deep_pass(c, c->level + c->deep, true, inserted, probed, &io);
  if (rc) return rc;
  l->deep = c->deep;
  l->distinct = distinct_before + inserted->n_new;
  inserted->distinct = l->distinct;
  if (probed->level) probed->distinct = l->distinct;
  if (!l->has_violation) {
    if (inserted->viol_mask) {
      l->has_violation = true; l->viol_fp = inserted->viol_fp; l->viol_mask = inserted->viol_mask; l->viol_level = inserted->level; l->viol_probed = false;
    } else if (probed->level && probed->viol_mask) {
      l->has_violation = true; l->viol_fp = probed->viol_fp; l->viol_mask = probed->viol_mask; l->viol_level = probed->level; l->viol_probed = true;
    }
  }
  return 0;
}

// One unit of progress of the automatic level scheme on a sharded run (collective): an ordinary sharded level while EVERY rank predicts
// that its part of the next one fits its idle record buffer (*what = 1: a = the level over all ranks), else vsrmc_shard_loop_deepen
// (*what = 2: a = the inserted level, b = the probed one or b->level == 0).
int32_t vsrmc_shard_loop_advance(vsrmc_shard_loop* l, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what) {
  if (!l || !a || !b || !what) return fail(VSRMC_E_ARG, "NULL argument");
  std::memset(b, 0, sizeof(*b));
  b->viol_fp = b->viol_index = ~(u64)0;
  u64 deep_wanted = 0;
  if (!l->replicated) {
    // a rank's share of the next level: what its own states generate and win — about its share of this one times the level's growth
    deep_wanted = (l->c->deep || !next_level_fits(l->c)) ? 1 : 0;
    const int rc = loop_deep_any(l, &deep_wanted);
    if (rc) return rc;
  }
  if (!deep_wanted) {
    vsrmc_level_info local;
    *what = 1;
    return vsrmc_shard_loop_step(l, a, &local);
  }
  *what = 2;
  return vsrmc_shard_loop_deepen(l, a, b);
}

// ---- checkpoint / recover of a sharded run between two units of progress (≙ TLC's checkpoints, per worker; round 6: also of a search that has gone
// beyond its record buffers) -------------------------------------------------------------------------------------------------------------------
// Every rank writes <prefix>.rank<r>of<w> (vsrmc_checker_save: its seen-set shard, its part of the newest stored level, the descriptors of the levels
// that live in the seen-sets only, its winner set) and <prefix>.rank<r>of<w>.loop (the level loop's own state: the run's totals, the violation — the
// same on every rank).  Two phases, so that a failure on one rank cannot leave files of different depths under one prefix: (1) level-tagged names,
// error codes gathered — on any failure the tagged files are removed and the previous checkpoint stays whole; (2) renames, gathered again.
namespace {
struct LoopChk {
  char magic[8];                 // "VSRMCLP1"
  int32_t world, rank, level, deep, replicated, has_violation, viol_mask, viol_level, viol_probed, pad_;
  u64 distinct, n_frontier, moved, bytes_sent, viol_fp, probe_parent_fp, replicate_below;
};
}  // namespace

int32_t vsrmc_shard_loop_save(vsrmc_shard_loop* l, const char* prefix) {
  if (!l || !prefix) return fail(VSRMC_E_ARG, "NULL argument");
  const std::string fin = std::string(prefix) + ".rank" + std::to_string(l->rank) + "of" + std::to_string(l->world);
  const std::string tag = fin + ".D" + std::to_string(l->level + l->deep);
  int rc = vsrmc_checker_save(l->c, tag.c_str());
  if (!rc) {
    LoopChk k;
    std::memset(&k, 0, sizeof(k));
    std::memcpy(k.magic, "VSRMCLP1", 8);
    k.world = l->world; k.rank = l->rank; k.level = l->level; k.deep = l->deep; k.replicated = l->replicated ? 1 : 0;
    k.has_violation = l->has_violation ? 1 : 0; k.viol_mask = l->viol_mask; k.viol_level = l->viol_level; k.viol_probed = l->viol_probed ? 1 : 0;
    k.distinct = l->distinct; k.n_frontier = l->n_frontier; k.moved = l->moved; k.bytes_sent = l->bytes_sent; k.viol_fp = l->viol_fp;
    k.probe_parent_fp = l->probe_parent_fp; k.replicate_below = l->replicate_below;
    FILE* f = std::fopen((tag + ".loop").c_str(), "wb");
    const bool ok = f && std::fwrite(&k, sizeof(k), 1, f) == 1;
    if (f && std::fclose(f) != 0) rc = fail(VSRMC_E_CFG, "writing " + tag + ".loop failed");
    if (!ok && !rc) rc = fail(VSRMC_E_CFG, "writing " + tag + ".loop failed");
  }
  u64 e = rc ? 1 : 0;
  int crc = loop_deep_any(l, &e);                                // (also before the run is sharded: the replicated phase's ranks checkpoint together too)
  if (crc || e) {
    std::remove(tag.c_str());
    std::remove((tag + ".loop").c_str());
    return crc ? crc : (rc ? rc : fail(VSRMC_E_STATE, "checkpoint: another rank could not write its shard; the previous checkpoint stays whole"));
  }
  e = (std::rename(tag.c_str(), fin.c_str()) != 0 || std::rename((tag + ".loop").c_str(), (fin + ".loop").c_str()) != 0) ? 1 : 0;
  crc = loop_deep_any(l, &e);
  if (crc) return crc;
  if (e) return fail(VSRMC_E_CFG, std::string("checkpoint: a rank could not move its shard into place under ") + prefix);
  return 0;
}

// The loop of a run vsrmc_shard_loop_save wrote, over a checker vsrmc_checker_load has recovered from THIS rank's file of the same prefix (collective:
// every rank must have loaded the same depth).
int32_t vsrmc_shard_loop_restore(vsrmc_checker* c, const vsrmc_comm* comm, uint64_t cand_cap, uint64_t rec_cap, uint64_t rec_words_cap, const char* prefix,
                                 vsrmc_shard_loop** out) {
  if (!c || !comm || !out || !prefix || !comm->alltoallv || !comm->allgather) return fail(VSRMC_E_ARG, "NULL argument");
  if (comm->world != c->opt.world || comm->rank != c->opt.rank) return fail(VSRMC_E_ARG, "the communicator and the checker disagree about rank / world");
  if (c->opt.exact_ties) return fail(VSRMC_E_STATE, "the native level loop runs single-pass levels (exact_ties = 0)");
  if (cand_cap < 1024) return fail(VSRMC_E_ARG, "cand_cap too small");
  const std::string path = std::string(prefix) + ".rank" + std::to_string(comm->rank) + "of" + std::to_string(comm->world) + ".loop";
  LoopChk k;
  std::memset(&k, 0, sizeof(k));
  FILE* f = std::fopen(path.c_str(), "rb");
  bool ok = f && std::fread(&k, sizeof(k), 1, f) == 1 && std::memcmp(k.magic, "VSRMCLP1", 8) == 0;
  if (f) std::fclose(f);
  ok = ok && k.world == comm->world && k.rank == comm->rank && k.level == c->level && k.deep == c->deep;
  HIPCHK(hipSetDevice(c->opt.device));
  vsrmc_shard_loop* l = new vsrmc_shard_loop();
  l->c = c; l->comm = *comm; l->rank = comm->rank; l->world = comm->world;
  l->cand_cap = cand_cap; l->rec_cap = rec_cap; l->rec_words_cap = rec_words_cap;
  // every rank says what it loaded: one checkpoint, or nobody goes on
  struct Row { u64 ok, level, deep, distinct; } mine = {ok ? 1ull : 0ull, (u64)k.level, (u64)k.deep, k.distinct};
  std::vector<Row> all((size_t)l->world);
  const int crc = loop_allgather(l, &mine, all.data(), (u32)sizeof(Row));
  bool same = crc == 0;
  for (const Row& r : all) same = same && r.ok && r.level == mine.level && r.deep == mine.deep && r.distinct == mine.distinct;
  if (!same) {
    delete l;
    return crc ? crc : fail(VSRMC_E_CFG, std::string(prefix) + ": the ranks' files are not one checkpoint of this run (missing, another world size, or different depths)");
  }
  const u64 w = (u64)l->world;
  hipError_t e = hipMalloc((void**)&l->cand_send, w * cand_cap * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&l->cand_recv, w * cand_cap * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&l->verdict_out, w * cand_cap);
  if (e == hipSuccess) e = hipMalloc((void**)&l->verdict_in, w * cand_cap);
  if (e != hipSuccess) { vsrmc_shard_loop_destroy(l); return fail(VSRMC_E_HIP, std::string("hipMalloc of the exchange buffers: ") + hipGetErrorString(e)); }
  l->level = k.level; l->deep = k.deep; l->replicated = k.replicated != 0; l->replicate_below = k.replicate_below;
  l->distinct = k.distinct; l->n_frontier = k.n_frontier; l->moved = k.moved; l->bytes_sent = k.bytes_sent;
  l->has_violation = k.has_violation != 0; l->viol_fp = k.viol_fp; l->viol_mask = k.viol_mask; l->viol_level = k.viol_level;
  l->viol_probed = k.viol_probed != 0; l->probe_parent_fp = k.probe_parent_fp;
  *out = l;
  return 0;
}

// the fingerprints of the counter-example of a violation a probe pass found (collective): Init .. the violator's parent, then the violator
int32_t vsrmc_shard_loop_probe_trace_fps(vsrmc_shard_loop* l, uint64_t* fps, int32_t cap, int32_t* n) {
  if (!l || !fps || !n) return fail(VSRMC_E_ARG, "NULL argument");
  if (!l->has_violation || !l->viol_probed || !l->probe_parent_fp) return fail(VSRMC_E_STATE, "no violation recorded by a probe pass");
  if (cap < l->viol_level) return fail(VSRMC_E_ARG, "buffer too small");
  const int rc = vsrmc_shard_loop_trace_fps(l, l->viol_level - 1, l->probe_parent_fp, fps);
  if (rc) return rc;
  fps[l->viol_level - 1] = l->viol_fp;
  *n = l->viol_level;
  return 0;
}

}  // extern "C"


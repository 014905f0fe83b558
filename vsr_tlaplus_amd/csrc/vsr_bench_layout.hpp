// vsr_bench_layout.hpp — a measurement, not a product path: the staging of k_expand over the checker's newest stored level, once as it is (variable-length
// RECORDS behind a ref array: SURVEY §8a row a1 "packed state record") and once with the same level laid out as fixed-stride COLUMNS (the SoA layout
// BASELINE.json's north_star names: one column per 8-byte word, a wave reads 64 consecutive states' word k in one 512-byte access).  Both kernels fill the
// same LDS tile (64 records at the level's odd word stride) and consume it the same way (every word read once more from LDS, xor-ed into a sink), from the
// same persistent-block / atomic-tile-cursor skeleton as k_expand: what differs is how the bytes of a tile come out of the HBM.  Five rounds argued which is
// better; tools/bench_layout.py prints the answer.  (included by vsrmc.hip; C ABI: vsrmc_checker_bench_staging)
#pragma once

namespace vsr {

// layout 0: records — exactly k_expand's staging (refs through LDS, 16 lanes per record, 16 loads per thread in flight)
// layout 1: columns — word k of state i at cols[k * n_pad + i]; lens[i] = its length; thread t takes state (t & 63), words (t >> 6) + 4 j
template <int LAYOUT, int OCC = 4>
__global__ void __launch_bounds__(256, OCC)
k_stage_bench(const u64* __restrict__ words, const u64* __restrict__ off, const u64* __restrict__ cols, const uint8_t* __restrict__ lens, u64 n_pad,
              u64 n, int stride, unsigned long long* cursor, unsigned long long* sink, int batch = 1 /* tiles per draw from the cursor */) {
  extern __shared__ u64 s_rec[];
  __shared__ u64 s_ref[64];
  __shared__ u64 s_tile;
  const int tid = threadIdx.x;
  const u64 ntiles = (n + 63) / 64;
  u64 acc = 0;
  u64 my_next = 0;
  int my_left = 0;
  auto draw = [&]() { if (my_left > 0) { my_left--; my_next++; } else { my_next = atomicAdd(cursor, 1ull) * (u64)batch; my_left = batch - 1; } };
  if (tid == 0) draw();
  for (;;) {
    if (tid == 0) {
      s_tile = my_next;
      if (my_next < ntiles) draw();
    }
    __syncthreads();
    const u64 t = s_tile;
    if (t >= ntiles) break;
    const u64 p_base = t * 64;
    const int np = (int)((n - p_base) < 64 ? (n - p_base) : 64);
    if constexpr (LAYOUT == 0) {
      if (tid < np) s_ref[tid] = off[p_base + tid];
      __syncthreads();
      u64 v[4][4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int p = (tid >> 4) + 16 * q;
        const u64 ref = p < np ? s_ref[p] : 0;
        const u64 o = ref >> 8;
        const int len = (int)(ref & 255) < stride ? (int)(ref & 255) : stride;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = (tid & 15) + 16 * j;
          v[q][j] = k < len ? words[o + k] : 0;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int p = (tid >> 4) + 16 * q;
        const int len = p < np ? ((int)(s_ref[p] & 255) < stride ? (int)(s_ref[p] & 255) : stride) : 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int k = (tid & 15) + 16 * j;
          if (k < len) s_rec[p * stride + k] = v[q][j];
        }
      }
    } else {
      const int p = tid & 63, g = tid >> 6;
      const int len = p < np ? (int)lens[p_base + p] : 0;
      u64 v[16];
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int k = g + 4 * j;
        v[j] = k < len ? cols[(u64)k * n_pad + p_base + p] : 0;
      }
#pragma unroll
      for (int j = 0; j < 16; j++) {
        const int k = g + 4 * j;
        if (k < len) s_rec[p * stride + k] = v[j];
      }
      if (g == 0) s_ref[p] = (u64)len;
    }
    __syncthreads();
    // the consumer: every staged word once more, as the enumeration and the apply phase read them (thread t owns record t & 63)
    {
      const int p = tid & 63, g = tid >> 6;
      const int len = p < np ? (int)(s_ref[p] & 255) : 0;
      for (int k = g; k < len && k < stride; k += 4) acc ^= s_rec[p * stride + k] + (u64)k;
    }
    __syncthreads();
  }
  if (acc == 0x5EEDull) atomicXor(sink, acc);                    // (never true in practice: keeps the loads alive without a store per thread)
  if (tid == 0 && blockIdx.x == 0) atomicAdd(sink, 1ull);
}

// The same tile loop, software-pipelined (round 6: is the staging bound by the CHAIN of dependent trips per tile — cursor, refs, words — or by how many
// bytes the resident blocks keep in flight?).  PIPE 1: the refs of the NEXT tile are fetched while this tile's words are in flight (one dependent trip per
// tile instead of two); PIPE 2: the WORDS of the next tile too — 16 registers per thread in flight while this tile is consumed (cursor three tiles ahead,
// refs two, words one).  OCC = resident blocks per CU.
template <int PIPE, int OCC>
__global__ void __launch_bounds__(256, OCC)
k_stage_pipe(const u64* __restrict__ words, const u64* __restrict__ off, u64 n, int stride, unsigned long long* cursor, unsigned long long* sink, int batch = 1) {
  extern __shared__ u64 s_rec[];
  __shared__ u64 s_ref[2][64];
  __shared__ u64 s_tile[4];                                     // ring: s_tile[i & 3] = index of the block's i-th tile
  const int tid = threadIdx.x;
  const u64 ntiles = (n + 63) / 64;
  const u64 NONE = ~(u64)0;
  u64 acc = 0;
  auto load_refs = [&](u64 t) -> u64 { return (tid < 64 && t < ntiles && t * 64 + tid < n) ? off[t * 64 + tid] : 0; };
  u64 v[4][4];
  auto load_words = [&](const u64* refs) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int p = (tid >> 4) + 16 * q;
      const u64 ref = refs[p];
      const u64 o = ref >> 8;
      const int len = (int)(ref & 255) < stride ? (int)(ref & 255) : stride;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = (tid & 15) + 16 * j;
        v[q][j] = k < len ? words[o + k] : 0;
      }
    }
  };
  auto store_words = [&](const u64* refs) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int p = (tid >> 4) + 16 * q;
      const int len = (int)(refs[p] & 255) < stride ? (int)(refs[p] & 255) : stride;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int k = (tid & 15) + 16 * j;
        if (k < len) s_rec[p * stride + k] = v[q][j];
      }
    }
  };
  // prologue: three tile indices (a block that draws an index past the end stops drawing)
  u64 d_next = 0;
  int d_left = 0;
  auto draw = [&]() -> u64 { if (d_left > 0) { d_left--; return ++d_next; } d_next = atomicAdd(cursor, 1ull) * (u64)batch; d_left = batch - 1; return d_next; };
  if (tid == 0) {
    u64 t0 = draw();
    u64 t1 = t0 < ntiles ? draw() : NONE;
    u64 t2 = t1 < ntiles ? draw() : NONE;
    s_tile[0] = t0; s_tile[1] = t1; s_tile[2] = t2; s_tile[3] = NONE;
  }
  __syncthreads();
  if (s_tile[0] >= ntiles) return;
  u64 my_ref = load_refs(s_tile[0]);                            // refs of tile i (PIPE 1) / of tile i + 1 (PIPE 2, after the prologue below)
  if constexpr (PIPE == 2) {
    if (tid < 64) s_ref[0][tid] = my_ref;
    __syncthreads();
    load_words(s_ref[0]);
    my_ref = load_refs(s_tile[1]);
  }
  for (u64 i = 0;; i++) {
    const int cur = (int)(i & 1);
    const u64 t_next = s_tile[(i + 1) & 3];
    if constexpr (PIPE == 1) {
      if (tid < 64) s_ref[cur][tid] = my_ref;                   // refs of tile i: fetched an iteration ago
      __syncthreads();
      load_words(s_ref[cur]);
      my_ref = load_refs(t_next);                               // refs of tile i + 1: in flight with the words of tile i
      if (tid == 0) s_tile[(i + 3) & 3] = s_tile[(i + 2) & 3] < ntiles ? draw() : NONE;
      store_words(s_ref[cur]);
    } else {
      store_words(s_ref[cur]);                                  // words of tile i: fetched an iteration ago
      if (tid < 64) s_ref[cur ^ 1][tid] = my_ref;               // refs of tile i + 1: fetched an iteration ago
    }
    __syncthreads();
    if constexpr (PIPE == 2) {
      if (t_next < ntiles) load_words(s_ref[cur ^ 1]);          // words of tile i + 1: in flight while tile i is consumed
      my_ref = load_refs(s_tile[(i + 2) & 3]);                  // refs of tile i + 2
      if (tid == 0) s_tile[(i + 3) & 3] = s_tile[(i + 2) & 3] < ntiles ? draw() : NONE;
    }
    {
      const int p = tid & 63, g = tid >> 6;
      const int len = (int)(s_ref[cur][p] & 255);
      for (int k = g; k < len && k < stride; k += 4) acc ^= s_rec[p * stride + k] + (u64)k;
    }
    __syncthreads();
    if (t_next >= ntiles) break;
  }
  if (acc == 0x5EEDull) atomicXor(sink, acc);
  if (tid == 0 && blockIdx.x == 0) atomicAdd(sink, 1ull);
}

// the transposition (untimed): record i, word k -> cols[k * n_pad + i]; holes (ref 0) get length 0
__global__ void k_to_columns(const u64* __restrict__ words, const u64* __restrict__ off, u64 n, u64 n_pad, int stride, u64* cols, uint8_t* lens) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 ref = off[i];
  const int len = (int)(ref & 255) < stride ? (int)(ref & 255) : stride;
  lens[i] = (uint8_t)len;
  const u64* r = words + (ref >> 8);
  for (int k = 0; k < len; k++) cols[(u64)k * n_pad + i] = r[k];
}

}  // namespace vsr

extern "C" {

// layout: 0 = the records as they are, 1 = fixed-stride columns (a transposed copy is made first, untimed; needs stride x 8 B x states of free memory).
// *ms_per_pass: HIP-event time of one staging pass over the newest stored level (average of `reps`); *bytes_per_pass: the bytes a pass has to read
// (layout 0: refs + record words; layout 1: lengths + record words: the padding of a column is never fetched).  Layouts 2-6: k_stage_pipe, see include/vsrmc.h.
int32_t vsrmc_checker_bench_staging(vsrmc_checker* c, int32_t layout, int32_t reps, double* ms_per_pass, uint64_t* bytes_per_pass) {
  if (!c || !ms_per_pass || !bytes_per_pass || reps < 1 || layout < 0 || layout > 10) return fail(VSRMC_E_ARG, "bad arguments");
  if (c->n_frontier == 0 || c->deep) return fail(VSRMC_E_STATE, "the staging benchmark reads the newest STORED level");
  HIPCHK(hipSetDevice(c->opt.device));
  const Model& M = c->model.M;
  const int stride = (int)std::min<u64>((u64)c->lds_stride, (u64)((M.fixed + (int)std::min<u64>(c->bag_known ? c->cur_max_bag : (u64)M.max_bag, 255)) | 1));
  if (stride > 64) return fail(VSRMC_E_ARG, "the staging benchmark covers records of at most 64 words (R <= 3)");
  const u64 n = c->n_frontier, n_pad = (n + 63) & ~(u64)63;
  u64* cols = nullptr;
  uint8_t* lens = nullptr;
  unsigned long long* d_ctr = nullptr;
  struct Free { void** p[3]; ~Free() { for (void** q : p) if (q && *q) (void)hipFree(*q); } } guard{{(void**)&cols, (void**)&lens, (void**)&d_ctr}};
  HIPCHK(hipMalloc((void**)&d_ctr, 16));
  if (layout == 1) {
    HIPCHK(hipMalloc((void**)&cols, (size_t)stride * n_pad * 8));
    HIPCHK(hipMalloc((void**)&lens, n_pad));
    HIPCHK(hipMemsetAsync(lens, 0, n_pad, c->stream));
    hipLaunchKernelGGL(vsr::k_to_columns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, (const u64*)c->words[c->cur], (const u64*)c->off[c->cur], n, n_pad, stride, cols, lens);
    HIPCHK(hipGetLastError());
  }
  const size_t lds = (size_t)64 * stride * 8;
  const int occ = (layout >= 4 && layout <= 6) || layout == 9 ? 8 : 4;                          // (layouts 4-6: eight resident blocks per CU — what a kernel of 64 registers could have)
  const unsigned grid = (unsigned)std::min<u64>((n + 63) / 64, (u64)c->num_cus * occ);
  const u64* const W = (const u64*)c->words[c->cur];
  const u64* const O = (const u64*)c->off[c->cur];
  float total = 0;
  for (int r = 0; r < reps + 1; r++) {                           // (one warm-up pass)
    HIPCHK(hipMemsetAsync(d_ctr, 0, 16, c->stream));
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (layout == 0)
      hipLaunchKernelGGL(vsr::k_stage_bench<0>, dim3(grid), dim3(256), lds, c->stream, (const u64*)c->words[c->cur], (const u64*)c->off[c->cur], (const u64*)nullptr, (const uint8_t*)nullptr, n_pad, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 2) hipLaunchKernelGGL((vsr::k_stage_pipe<1, 4>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 3) hipLaunchKernelGGL((vsr::k_stage_pipe<2, 4>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 4) hipLaunchKernelGGL((vsr::k_stage_bench<0, 8>), dim3(grid), dim3(256), lds, c->stream, W, O, (const u64*)nullptr, (const uint8_t*)nullptr, n_pad, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 5) hipLaunchKernelGGL((vsr::k_stage_pipe<1, 8>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 6) hipLaunchKernelGGL((vsr::k_stage_pipe<2, 8>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 1);
    else if (layout == 7) hipLaunchKernelGGL((vsr::k_stage_bench<0, 4>), dim3(grid), dim3(256), lds, c->stream, W, O, (const u64*)nullptr, (const uint8_t*)nullptr, n_pad, n, stride, d_ctr, d_ctr + 1, 4);
    else if (layout == 8) hipLaunchKernelGGL((vsr::k_stage_pipe<2, 4>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 4);
    else if (layout == 9) hipLaunchKernelGGL((vsr::k_stage_pipe<2, 8>), dim3(grid), dim3(256), lds, c->stream, W, O, n, stride, d_ctr, d_ctr + 1, 4);
    else if (layout == 10) hipLaunchKernelGGL((vsr::k_stage_bench<0, 4>), dim3(grid), dim3(256), lds, c->stream, W, O, (const u64*)nullptr, (const uint8_t*)nullptr, n_pad, n, stride, d_ctr, d_ctr + 1, 16);
    else
      hipLaunchKernelGGL(vsr::k_stage_bench<1>, dim3(grid), dim3(256), lds, c->stream, (const u64*)nullptr, (const u64*)nullptr, (const u64*)cols, (const uint8_t*)lens, n_pad, n, stride, d_ctr, d_ctr + 1, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, c->ev[0], c->ev[1]));
    if (r) total += ms;
  }
  *ms_per_pass = (double)total / reps;
  *bytes_per_pass = (layout != 1 ? n * 8 : n) + c->cur_rec_w * 8;
  return 0;
}

}  // extern "C"

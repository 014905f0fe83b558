// bench_claim_partition.hip — the structural experiment the round-5 review asked for (VERDICT r05 "next round" item 1), as a MEASUREMENT beside the product:
// is a REGION-PARTITIONED claim of the seen-set cheaper than the direct one k_expand makes?
//
//   direct       every candidate (fingerprint, key) probes the 2^31-slot table where its fingerprint sends it: a random 16-byte access in 32 GB, a 128-byte
//                line per probe through the fabric (what table_claim_fused does inside k_expand, here alone in a kernel at full occupancy)
//   partitioned  the same candidates are first bucketed by the top bits of their slot index into 256 regions of 134 MB (2^31 slots; 268 MB at 2^32) (histogram + prefix + scatter: 16 bytes
//                written and read again per candidate, coalesced), then claimed region by region, so that the probes of a moment fall into one window
//                of the table that fits the 256-MB Infinity Cache
//
// Same table layout (16-byte slots: fingerprint | meta), same operations per candidate (16-byte load of the home slot, compare-and-swap on empty, atomicMin on
// the meta word, linear probing) as csrc/vsr_kernels.hpp: probe_insert / table_claim_fused.  Synthetic keys as SURVEY §8(d) prescribes for seen-set
// micro-benchmarks: splitmix64, 85 % of the candidates duplicates of states already in the table (1 - 1/g for g = 6.5), table load 0.15 (config 2's final
// load) or 0.42 (README's).  Build + run on the GPU box: tools/bench_claim_partition.sh.  Prints one JSON line; rocprofv3 --pmc passes of the same binary
// give the fabric traffic per kernel (profiles/r06_claim_partition_*.md).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint64_t u64;
typedef uint32_t u32;
struct Slot { u64 fp, meta; };
typedef u64 u64x2 __attribute__((ext_vector_type(2)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__host__ __device__ inline u64 splitmix64(u64 x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ inline u64 key_of_state(u64 i) { u64 k = splitmix64(i * 2 + 0x5EED); return k ? k : 1; }   // the fingerprint of "state i"

__global__ void k_init(Slot* t, u64 n) {
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (u64)gridDim.x * blockDim.x) { t[i].fp = 0; t[i].meta = ~(u64)0; }
}

// find-or-insert + min-merge of the key, as the product does it (home slot first: one 16-byte load answers a duplicate)
__device__ inline void claim(Slot* table, u64 mask, u64 fp, u64 key, u32* n_new /* per lane: no hot global counter, as in the product */) {
  u64 i = fp & mask;
  for (u32 step = 0; step < 4096; step++, i = (i + 1) & mask) {
    const u64x2 s = *(const u64x2*)&table[i];
    u64 cur = s.x;
    if (cur == 0) {
      cur = atomicCAS((unsigned long long*)&table[i].fp, 0ull, (unsigned long long)fp);
      if (cur == 0) { atomicMin((unsigned long long*)&table[i].meta, (unsigned long long)key); (*n_new)++; return; }
    }
    if (cur == fp) {
      if (s.y > key || cur != s.x) atomicMin((unsigned long long*)&table[i].meta, (unsigned long long)key);
      return;
    }
  }
}

// candidate c: with probability dup / 256 a state the table already holds (one of n_old), else a state of its own (new; several candidates may name it)
__device__ inline void candidate(u64 c, u64 n_old, u32 dup, u64 seed, u64* fp, u64* key) {
  const u64 r = splitmix64(c ^ seed);
  const bool old = (u32)(r & 255) < dup;
  const u64 state = old ? (r >> 8) % n_old : n_old + (r >> 8) % (n_old / 2 + 1);
  *fp = key_of_state(state);
  *key = (r >> 20) | ((u64)1 << 55);                                 // a meta word of some level: what atomicMin merges
}

__device__ inline void flush_count(u32 mine, unsigned long long* n_new) {   // one atomic per wave
  for (int o = 32; o > 0; o >>= 1) mine += __shfl_down(mine, o);
  if ((threadIdx.x & 63) == 0 && mine) atomicAdd(n_new, (unsigned long long)mine);
}
__global__ void k_preload(Slot* table, u64 mask, u64 n_old, unsigned long long* n_new) {
  u32 mine = 0;
  for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < n_old; i += (u64)gridDim.x * blockDim.x) claim(table, mask, key_of_state(i), (u64)1 << 54, &mine);
  flush_count(mine, n_new);
}
__global__ void k_make(u64* cand, u64 m, u64 n_old, u32 dup, u64 seed) {
  for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += (u64)gridDim.x * blockDim.x) candidate(c, n_old, dup, seed, &cand[2 * c], &cand[2 * c + 1]);
}
// ORDER = 0: direct — candidates in the order k_expand would produce them (arbitrary); 1: the same kernel over the bucketed array (a name of its own for rocprofv3)
template <int ORDER>
__global__ void k_claim(Slot* table, u64 mask, const u64* __restrict__ cand, u64 m, unsigned long long* n_new) {
  u32 mine = 0;
  for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += (u64)gridDim.x * blockDim.x) claim(table, mask, cand[2 * c], cand[2 * c + 1], &mine);
  flush_count(mine, n_new);
}
// partitioned, pass 1: how many candidates fall into each region (block-local histogram in LDS, one global atomic per region and block)
template <int R>
__global__ void k_hist(const u64* __restrict__ cand, u64 m, u64 mask, int shift, unsigned long long* counts) {
  __shared__ u32 h[R];
  for (int i = threadIdx.x; i < R; i += blockDim.x) h[i] = 0;
  __syncthreads();
  for (u64 c = (u64)blockIdx.x * blockDim.x + threadIdx.x; c < m; c += (u64)gridDim.x * blockDim.x) atomicAdd(&h[(cand[2 * c] & mask) >> shift], 1u);
  __syncthreads();
  for (int i = threadIdx.x; i < R; i += blockDim.x) if (h[i]) atomicAdd(&counts[i], (unsigned long long)h[i]);
}
// pass 2: scatter into the regions' buckets — a block takes a contiguous chunk of candidates, counts it per region in LDS, reserves its share of every
// bucket with one global atomic per region, then writes (the writes of one region are contiguous within the block's reservation)
template <int R>
__global__ void k_scatter(const u64* __restrict__ cand, u64 m, u64 mask, int shift, unsigned long long* cursors, u64* out, u64 chunk) {
  __shared__ u32 h[R];
  __shared__ unsigned long long base[R];
  for (u64 c0 = (u64)blockIdx.x * chunk; c0 < m; c0 += (u64)gridDim.x * chunk) {
    const u64 c1 = c0 + chunk < m ? c0 + chunk : m;
    for (int i = threadIdx.x; i < R; i += blockDim.x) h[i] = 0;
    __syncthreads();
    for (u64 c = c0 + threadIdx.x; c < c1; c += blockDim.x) atomicAdd(&h[(cand[2 * c] & mask) >> shift], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < R; i += blockDim.x) { base[i] = h[i] ? atomicAdd(&cursors[i], (unsigned long long)h[i]) : 0; h[i] = 0; }
    __syncthreads();
    for (u64 c = c0 + threadIdx.x; c < c1; c += blockDim.x) {
      const u64 fp = cand[2 * c], key = cand[2 * c + 1];
      const int r = (int)((fp & mask) >> shift);
      const u64 o = base[r] + atomicAdd(&h[r], 1u);
      out[2 * o] = fp;
      out[2 * o + 1] = key;
    }
    __syncthreads();
  }
}

int main(int argc, char** argv) {
  const int table_log2 = argc > 1 ? std::atoi(argv[1]) : 31;
  const double load = argc > 2 ? std::atof(argv[2]) : 0.15;
  const int m_log2 = argc > 3 ? std::atoi(argv[3]) : 28;
  const u32 dup = 218;                                               // 218 / 256 = 85 % duplicates of old states
  constexpr int R = 256;                                             // 2^31 slots x 16 B / 256 = 134 MB per region (2^32: 268 MB): inside the 256-MB Infinity Cache
  const u64 slots = (u64)1 << table_log2, mask = slots - 1, n_old = (u64)(load * (double)slots), m = (u64)1 << m_log2;
  const int shift = table_log2 - 8;
  Slot* table;
  u64 *cand, *bucketed;
  unsigned long long *d_cnt;
  CHK(hipMalloc((void**)&table, slots * sizeof(Slot)));
  CHK(hipMalloc((void**)&cand, m * 16));
  CHK(hipMalloc((void**)&bucketed, m * 16));
  CHK(hipMalloc((void**)&d_cnt, (2 * R + 2) * 8));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  auto timed = [&](auto&& launch) { CHK(hipEventRecord(e0)); launch(); CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1)); float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); CHK(hipGetLastError()); return (double)ms; };
  auto fresh_table = [&]() {
    hipLaunchKernelGGL(k_init, dim3(8192), dim3(256), 0, 0, table, slots);
    CHK(hipMemset(d_cnt, 0, (2 * R + 2) * 8));
    hipLaunchKernelGGL(k_preload, dim3(8192), dim3(256), 0, 0, table, mask, n_old, d_cnt + 2 * R);
    CHK(hipDeviceSynchronize());
  };
  double ms_direct[3], ms_hist[3], ms_scatter[3], ms_claim_part[3];
  unsigned long long new_direct = 0, new_part = 0;
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_make, dim3(8192), dim3(256), 0, 0, cand, m, n_old, dup, (u64)(0xC0FFEE + rep));
    fresh_table();
    CHK(hipMemset(d_cnt + 2 * R + 1, 0, 8));
    ms_direct[rep] = timed([&]() { hipLaunchKernelGGL(k_claim<0>, dim3(256 * 8), dim3(256), 0, 0, table, mask, (const u64*)cand, m, d_cnt + 2 * R + 1); });
    CHK(hipMemcpy(&new_direct, d_cnt + 2 * R + 1, 8, hipMemcpyDeviceToHost));
    fresh_table();
    ms_hist[rep] = timed([&]() { hipLaunchKernelGGL(k_hist<R>, dim3(256 * 8), dim3(256), 0, 0, (const u64*)cand, m, mask, shift, d_cnt); });
    std::vector<unsigned long long> cnt(R), cur(R);
    CHK(hipMemcpy(cnt.data(), d_cnt, R * 8, hipMemcpyDeviceToHost));
    unsigned long long acc = 0;
    for (int i = 0; i < R; i++) { cur[i] = acc; acc += cnt[i]; }
    CHK(hipMemcpy(d_cnt + R, cur.data(), R * 8, hipMemcpyHostToDevice));
    ms_scatter[rep] = timed([&]() { hipLaunchKernelGGL(k_scatter<R>, dim3(256 * 4), dim3(256), 0, 0, (const u64*)cand, m, mask, shift, d_cnt + R, bucketed, (u64)16384); });
    CHK(hipMemset(d_cnt + 2 * R + 1, 0, 8));
    // one launch over the bucketed array, in order: the blocks work through region 0, then region 1, ... (grid-stride: all of them in the same window at a time)
    ms_claim_part[rep] = timed([&]() { hipLaunchKernelGGL(k_claim<1>, dim3(256 * 8), dim3(256), 0, 0, table, mask, (const u64*)bucketed, m, d_cnt + 2 * R + 1); });
    CHK(hipMemcpy(&new_part, d_cnt + 2 * R + 1, 8, hipMemcpyDeviceToHost));
  }
  auto best = [](double* a) { double b = a[0]; for (int i = 1; i < 3; i++) b = a[i] < b ? a[i] : b; return b; };
  const double d = best(ms_direct), h = best(ms_hist), s = best(ms_scatter), p = best(ms_claim_part);
  std::printf("{\"table_slots_log2\": %d, \"table_GB\": %.1f, \"load\": %.2f, \"candidates\": %llu, \"duplicates_of_old_states\": 0.85, \"regions\": %d, \"region_MB\": %.0f, "
              "\"direct_claim_ms\": %.3f, \"partitioned\": {\"histogram_ms\": %.3f, \"scatter_ms\": %.3f, \"claim_ms\": %.3f, \"total_ms\": %.3f}, "
              "\"partitioned_over_direct\": %.3f, \"claim_only_over_direct\": %.3f, \"new_states_direct\": %llu, \"new_states_partitioned\": %llu, "
              "\"direct_candidates_per_s\": %.3e, \"note\": \"best of 3; the product's k_expand makes these probes inside a kernel that also stages, enumerates, hashes and writes\"}\n",
              table_log2, (double)slots * 16 / 1e9, load, (unsigned long long)m, R, (double)slots * 16 / R / 1e6, d, h, s, p, h + s + p, (h + s + p) / d, p / d,
              new_direct, new_part, (double)m / (d / 1e3));
  return 0;
}

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

#define VSRMC_FP_VERSION 2          // fingerprint function of this build; checkpoints of another version are refused

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

// ---- the sections of the host side, one file per concern (one translation unit: they share the helpers above) ------------------
#include "host_model.hpp"        // models, cfg reader, state printing / parsing
#include "host_fpset.hpp"        // FPSet / StateQueue handles
#include "host_batch.hpp"        // expand / fingerprint batches, trace import, simulation
#include "host_checker.hpp"      // the checker: buffers, level phases, passes, trace walks
extern "C" {
#include "vsr_deep.hpp"          // levels beyond the record buffers (virtual / regenerated / streamed / probed), vsrmc_checker_deepen
}  // extern "C"
#include "host_search.hpp"       // probe level, automatic level scheme, vsrmc_check
#include "host_shard.hpp"        // phases of a sharded level (C ABI)
#include "host_checkpoint.hpp"   // checkpoint / recover, accessors, traces, destroy
#include "vsr_shard_loop.hpp"    // the sharded level loop in C++ over RCCL / host callbacks
#include "host_tlcfp.hpp"        // TLC's FP64 as a mode
#include "vsr_bench_layout.hpp"  // measurement: k_expand's staging over records vs over fixed-stride columns (tools/bench_layout.py)

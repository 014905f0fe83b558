// vsr_model.hpp — the lowered VSR model: packed record layout, field accessors, value permutation and the
// component hash that the fingerprint is built from.  Shared by every kernel and by the host driver.
//
// Reference behaviour being lowered (read-only at /root/reference):
//   vsr-revisited/paper/VSR.tla:119-138  the 20 state variables
//   vsr-revisited/paper/VSR.tla:149-151  `view` (state identity) and `symmValues` (symmetry set)
//   vsr-revisited/paper/VSR.cfg:4-8,29,31 constants, VIEW view, SYMMETRY symmValues
// TLC side replaced: tlc2.tool.TLCState (state record) and TLCState.fingerPrint().
//
// Record ("device layout"), in 64-bit words:
//   [0]                      header: nmsg(8) | aux_svc(3)<<8 | acked[v](2)<<(11+2v)      (aux: NOT in the view)
//   [1 .. 1+R*wpr)           R replica blocks of wpr words: A word + packed x-slots (log, R DVC slots)
//   [h0 .. h0+np)            H[i] = hash of the view of this record under value-permutation i (np = n! or 1)
//   [fixed .. fixed+nmsg)    the message bag: one word per (message record, delivery count); unordered
// The "wire layout" used across the C ABI (and by the CPU oracle's codec) is the same record without the H words.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VSR_HD __host__ __device__ __forceinline__
#else
#define VSR_HD inline
#endif

// EXPERIMENT KNOB (tools/ab_build.sh NAME "-DVSR_PAD_WORDS=6"): dead words between the H words and the bag of every device-layout record.
// They are staged, copied to every successor and never looked at: the measured cost of S + 8 * VSR_PAD_WORDS bytes per state brackets what a
// record SMALLER by as much would buy (the slope dT/dS of k_expand).  0 in the product.
#ifndef VSR_PAD_WORDS
#define VSR_PAD_WORDS 0
#endif

namespace vsr {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint16_t u16;

enum { ST_NORMAL = 0, ST_VIEWCHANGE = 1, ST_RECOVERING = 2 };                                     // VSR.tla:99-101
enum { T_SVC = 1, T_PREPARE = 2, T_PREPAREOK = 3, T_DVC = 4, T_SV = 5, T_GETSTATE = 6, T_NEWSTATE = 7 };  // :104-115

// Action ids in `Next` order (VSR.tla:896-918); 0 = Init.
enum {
  A_Init = 0, A_TimerSendSVC, A_ReceiveHigherSVC, A_ReceiveMatchingSVC, A_SendDVC, A_ReceiveHigherDVC,
  A_ReceiveMatchingDVC, A_SendSV, A_ReceiveSV, A_ReceiveClientRequest, A_ReceivePrepareMsg, A_ReceivePrepareOkMsg,
  A_ExecuteOp, A_SendGetState, A_ReceiveGetState, A_ReceiveNewState, A_COUNT
};

// Error codes raised by the device path (first one wins; the run aborts like a TLC evaluation error would).
enum {
  ERR_NONE = 0,
  ERR_EVAL_421 = 1,        // VSR.tla:421 `m.commit`: nonexistent record field (ClientCount >= 2)
  ERR_EVAL_DOMAIN = 2,     // function applied outside its domain (log index, aux_client_acked)
  ERR_EVAL_CHOOSE = 3,     // CHOOSE over an empty set
  ERR_REP_RANGE = 10,      // a field left the range the packed record can hold
  ERR_REP_I1 = 11,         // rep_svc_recv / rep_dvc_recv holds a record the mask/slot form cannot express
  ERR_REP_I2 = 12,         // two different DoViewChange records from one source
  ERR_REP_COUNT = 13,      // delivery count > 3
  ERR_REP_BAG = 14,        // bag larger than the configured capacity
  ERR_TABLE_FULL = 20,     // seen-set out of slots
  ERR_FRONTIER_FULL = 21,  // frontier / pending buffers out of space
  ERR_LEVELS = 22,         // more BFS levels than the meta word can hold
  ERR_INTERNAL = 30        // the two statements of the guards (guard_slot / gen) disagree
};

struct Model {
  int model_id;          // 0 = VSR.tla, 1 = analysis/03-state-transfer/VR_STATE_TRANSFER.tla (vrst_actions.hpp), 2 = analysis/04-application-state/VR_APP_STATE.tla (vras_actions.hpp)
  int R, C, n, L;        // ReplicaCount, ClientCount, Cardinality(Values), StartViewOnTimerLimit
  int wpr;               // words per replica block
  int h0;                // index of H[0]
  int np;                // number of value permutations hashed
  int fixed;             // index of the first bag word
  int assume_commit;     // policy for VSR.tla:421 (0 = strict: evaluation error)
  int inv_mask;          // bit0 AcknowledgedWriteNotLost, bit1 AcknowledgedWritesExistOnMajority
  int max_bag;           // bag capacity enforced by the kernels
  int m0;                // first message-bound ordinal = 4R + R*C*n
  u32 primtab;           // Primary(v) for v = 0..7, 3 bits each (a table: `%` by a run-time R costs ~20 instructions)
  u32 pitab[6];          // permutation i: pi[v] in bits 2v..2v+1
  u64 test_bad_fp;       // TEST HOOK (VSRMC_TEST_FORCE_BAD=<hex fingerprint>:<mask>, read when a model is built): the state with this fingerprint
  u32 test_bad_mask;     // fails the invariants of this mask — whatever they say — in the sharded and the two-kernel paths.  0 = off.  Lets a test
  u32 test_pad_;         // put a violator of ANY mask on a rank that does not own it (the shipped cfgs reach no violation of masks 4 / 8 / 16).
  u64 fp_seed;           // xor-ed into every salt of the view hash: 0 = the function the fixtures were made with; any other value is an
                         // independent member of the same family (second-hash audit: counts must not depend on it, vsrmc_model_set_fp_seed)
};

static const u64 SALT_MSG = 0x9E3779B97F4A7C15ULL;
static const u64 KEYMASK = ~((u64)3 << 21);              // a bag word without its delivery count
// bytes of a word that hold log entries (0x01 per byte)
static const u64 LOGB_REP1 = 0x0101010000010101ULL;      // replica word 1: x0 = own log (bytes 0-2), x1 DVC slot (log bytes 5-7)
static const u64 LOGB_REPK = 0x0101010001010100ULL;      // replica word k>=2: two DVC slots (log bytes 1-3, 5-7)
static const u64 LOGB_MSG = 0x0001010100000000ULL;       // bag word: entry / log in bits 32-55

VSR_HD u64 fmix64(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}

// Position salt of word k of replica r's block: fmix64(0xA0761D6478BD642F + 8 r + k), folded at compile time and picked with a
// select chain over r (a per-lane table index would force the table into scratch memory; computing it costs one fmix64 per use).
constexpr u64 fmix64_c(u64 x) {
  x ^= x >> 33;
  x *= 0xff51afd7ed558ccdULL;
  x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL;
  x ^= x >> 33;
  return x;
}
template <int K>
VSR_HD u64 salt_word(int r) {
  constexpr u64 B = 0xA0761D6478BD642FULL + (u64)K;
  constexpr u64 s1 = fmix64_c(B + 8), s2 = fmix64_c(B + 16), s3 = fmix64_c(B + 24), s4 = fmix64_c(B + 32), s5 = fmix64_c(B + 40);
  return r == 1 ? s1 : r == 2 ? s2 : r == 3 ? s3 : r == 4 ? s4 : s5;
}

#ifndef VSR_SALT_PIN
#define VSR_SALT_PIN 1
#endif
// salt(position) ^ fp_seed.  The salt is a literal chosen by r; fp_seed is a kernel argument.  Written as one expression the optimiser turned the choice into a
// switch over r whose arms are `literal ^ fp_seed` — loop-invariant, so hoisted out of the tile loop, nine 64-bit values (README configuration) held in VGPRs,
// spilled, and RELOADED FROM SCRATCH in the middle of every successor's hash chain (three dependent scratch loads per successor; the review's "34 spilled
// VGPRs").  The empty asm pins the chosen literal in a VGPR pair before the seed is xor-ed in: nothing invariant is left to hoist.
template <int K>
VSR_HD u64 salt_seeded(u64 fp_seed, int r) {
  u64 sr = salt_word<K>(r);
#if defined(__HIP_DEVICE_COMPILE__) && VSR_SALT_PIN
  asm volatile("" : "+v"(sr));
#endif
  return sr ^ fp_seed;
}

// ---- header -------------------------------------------------------------------------------------------------
VSR_HD int hdr_nmsg(u64 h) { return (int)(h & 0xFF); }
VSR_HD int hdr_aux_svc(u64 h) { return (int)((h >> 8) & 7); }
VSR_HD int hdr_acked(u64 h, int v) { return (int)((h >> (11 + 2 * v)) & 3); }   // 0 absent, 1 FALSE, 2 TRUE
VSR_HD u64 hdr_set_nmsg(u64 h, int n) { return (h & ~(u64)0xFF) | (u64)n; }
VSR_HD u64 hdr_set_acked(u64 h, int v, int a) { return (h & ~((u64)3 << (11 + 2 * v))) | ((u64)a << (11 + 2 * v)); }

// ---- replica A word -----------------------------------------------------------------------------------------
VSR_HD int a_status(u64 A) { return (int)(A & 3); }
VSR_HD int a_view(u64 A) { return (int)((A >> 2) & 7); }
VSR_HD int a_op(u64 A) { return (int)((A >> 5) & 3); }
VSR_HD int a_commit(u64 A) { return (int)((A >> 7) & 3); }
VSR_HD int a_lnv(u64 A) { return (int)((A >> 9) & 7); }
VSR_HD int a_sent_dvc(u64 A) { return (int)((A >> 12) & 1); }
VSR_HD int a_sent_sv(u64 A) { return (int)((A >> 13) & 1); }
VSR_HD int a_svcmask(u64 A) { return (int)((A >> 14) & 31); }
VSR_HD int a_peer(u64 A, int p) { return (int)((A >> (19 + 2 * (p - 1))) & 3); }
VSR_HD int a_ctrow(u64 A, int c) { return (int)((A >> (29 + 5 * (c - 1))) & 31); }
VSR_HD u64 a_set(u64 A, int shift, int width, int val) {
  u64 m = (((u64)1 << width) - 1) << shift;
  return (A & ~m) | ((u64)val << shift);
}
VSR_HD u64 a_set_status(u64 A, int v) { return a_set(A, 0, 2, v); }
VSR_HD u64 a_set_view(u64 A, int v) { return a_set(A, 2, 3, v); }
VSR_HD u64 a_set_op(u64 A, int v) { return a_set(A, 5, 2, v); }
VSR_HD u64 a_set_commit(u64 A, int v) { return a_set(A, 7, 2, v); }
VSR_HD u64 a_set_lnv(u64 A, int v) { return a_set(A, 9, 3, v); }
VSR_HD u64 a_set_sent_dvc(u64 A, int v) { return a_set(A, 12, 1, v); }
VSR_HD u64 a_set_sent_sv(u64 A, int v) { return a_set(A, 13, 1, v); }
VSR_HD u64 a_set_svcmask(u64 A, int v) { return a_set(A, 14, 5, v); }
VSR_HD u64 a_set_peer(u64 A, int p, int v) { return a_set(A, 19 + 2 * (p - 1), 2, v); }
VSR_HD u64 a_set_ctrow(u64 A, int c, int row) { return a_set(A, 29 + 5 * (c - 1), 5, row); }
VSR_HD int ct_req(int row) { return row & 3; }
VSR_HD int ct_op(int row) { return (row >> 2) & 3; }
VSR_HD int ct_exec(int row) { return (row >> 4) & 1; }
VSR_HD int ct_make(int req, int op, int exec) { return req | (op << 2) | (exec << 4); }

// ---- log (3 entry bytes; entry with op number i in byte i-1) and entry byte ------------------------------------
VSR_HD int entry_make(int view, int val, int client, int req) { return view | (val << 3) | ((client - 1) << 5) | (req << 6); }
VSR_HD int entry_val(int b) { return (b >> 3) & 3; }
VSR_HD int entry_client(int b) { return ((b >> 5) & 1) + 1; }
VSR_HD int entry_req(int b) { return (b >> 6) & 3; }
VSR_HD int log_byte(u32 lg, int opn) { return (int)((lg >> (8 * (opn - 1))) & 0xFF); }
VSR_HD int log_len(u32 lg) { return ((lg & 0xFF) ? 1 : 0) + ((lg & 0xFF00) ? 1 : 0) + ((lg & 0xFF0000) ? 1 : 0); }
VSR_HD u32 log_prefix(u32 lg, int t) { return t >= 3 ? lg : (lg & ((1u << (8 * t)) - 1)); }   // entries 1..t

// ---- DVC slot (one per source inside a replica block) ----------------------------------------------------------
VSR_HD u32 dvc_make(int lnv, int op, int commit, u32 lg) { return 1u | ((u32)lnv << 1) | ((u32)op << 4) | ((u32)commit << 6) | (lg << 8); }
VSR_HD int dvc_lnv(u32 x) { return (x >> 1) & 7; }
VSR_HD int dvc_op(u32 x) { return (x >> 4) & 3; }
VSR_HD int dvc_commit(u32 x) { return (x >> 6) & 3; }
VSR_HD u32 dvc_log(u32 x) { return x >> 8; }

// ---- bag word -------------------------------------------------------------------------------------------------
VSR_HD int m_type(u64 w) { return (int)(w & 7); }
VSR_HD int m_view(u64 w) { return (int)((w >> 3) & 7); }
VSR_HD int m_dest(u64 w) { return (int)((w >> 6) & 7); }
VSR_HD int m_source(u64 w) { return (int)((w >> 9) & 7); }
VSR_HD int m_op(u64 w) { return (int)((w >> 12) & 3); }
VSR_HD int m_commit(u64 w) { return (int)((w >> 14) & 3); }
VSR_HD int m_lnv(u64 w) { return (int)((w >> 16) & 7); }
VSR_HD int m_first_op(u64 w) { return (int)((w >> 19) & 3); }
VSR_HD int m_count(u64 w) { return (int)((w >> 21) & 3); }
VSR_HD u32 m_lg(u64 w) { return (u32)(w >> 32); }
VSR_HD u64 m_make(int type, int view, int dest, int source, int op, int commit, int lnv, int first_op, u32 lg) {
  return (u64)type | ((u64)view << 3) | ((u64)dest << 6) | ((u64)source << 9) | ((u64)op << 12) | ((u64)commit << 14) |
         ((u64)lnv << 16) | ((u64)first_op << 19) | ((u64)lg << 32);
}
VSR_HD u64 m_set_dest(u64 w, int d) { return (w & ~((u64)7 << 6)) | ((u64)d << 6); }
VSR_HD u64 m_set_count(u64 w, int c) { return (w & KEYMASK) | ((u64)c << 21); }

VSR_HD int primary_of(const Model& M, int view) { return (int)((M.primtab >> (3 * view)) & 7); }   // VSR.tla:287-288, tabulated

// ---- value permutation (VSR.tla:151) of the log-entry bytes of one word ----------------------------------------
// m01 has 0x01 in every byte that is a log entry; an entry byte is in use iff its view field (bits 0-2) != 0.
VSR_HD u64 permute_word(u64 w, u64 m01, u32 pt) {
  if (pt == 0x24u) return w;                                    // identity (pi[v] = v): permutation 0 of every model
  u64 nz = (w | (w >> 1) | (w >> 2)) & m01;
  if (!nz) return w;
  u64 a = (w >> 3) & nz, b = (w >> 4) & nz;
  u64 is1 = a & ~b, is2 = b & ~a, is0 = nz & ~(a | b);
  u64 na = (is0 & (0 - (u64)(pt & 1))) | (is1 & (0 - (u64)((pt >> 2) & 1))) | (is2 & (0 - (u64)((pt >> 4) & 1)));
  u64 nb = (is0 & (0 - (u64)((pt >> 1) & 1))) | (is1 & (0 - (u64)((pt >> 3) & 1))) | (is2 & (0 - (u64)((pt >> 5) & 1)));
  return (w & ~((nz << 3) | (nz << 4))) | (na << 3) | (nb << 4);
}

// hash of one bag word under permutation pt
VSR_HD u64 hash_msg(const Model& M, u64 w, u32 pt) { return fmix64(permute_word(w, LOGB_MSG, pt) ^ (SALT_MSG ^ M.fp_seed)); }

// does a word hold any value (an in-use log entry byte)?  If not, its hash term is the same under every permutation.
VSR_HD bool word_has_values(u64 w, u64 m01) { return ((w | (w >> 1) | (w >> 2)) & m01) != 0; }

// Zobrist-style view hash: H_pi = sum over the words of the replica blocks of fmix64(pi(word) ^ salt(position))
//                                + sum over the bag of fmix64(pi(word) ^ SALT_MSG)
// hash term of word K of replica r's block under permutation pt
template <int K>
VSR_HD u64 hash_rep_word(const Model& M, u64 w, int r, u32 pt) {
  return fmix64(permute_word(w, K == 0 ? (u64)0 : K == 1 ? LOGB_REP1 : LOGB_REPK, pt) ^ salt_seeded<K>(M.fp_seed, r));
}
template <typename PTR>
VSR_HD u64 hash_rep_block(const Model& M, PTR b, int r, u32 pt) {
  u64 h = hash_rep_word<0>(M, b[0], r, pt) + hash_rep_word<1>(M, b[1], r, pt);
  if (M.wpr > 2) h += hash_rep_word<2>(M, b[2], r, pt);           // written out: b may be a register array
  if (M.wpr > 3) h += hash_rep_word<3>(M, b[3], r, pt);
  return h;
}

// canonical aux key under permutation pt: aux_svc | acked'[pi[v]] = acked[v]
VSR_HD u32 auxkey_of(const Model& M, u64 hdr, u32 pt) {
  u32 ak = (u32)hdr_aux_svc(hdr);
  for (int v = 0; v < M.n; v++) ak |= (u32)hdr_acked(hdr, v) << (3 + 2 * ((pt >> (2 * v)) & 3));
  return ak;
}

// Full (non-incremental) view hashes of a device-layout record; writes H[0..np).  Used for Init and by tests.
template <typename PTR>
VSR_HD void hash_full(const Model& M, PTR rec, u64* H) {
  int nmsg = hdr_nmsg(rec[0]);
  for (int i = 0; i < M.np; i++) {
    u32 pt = M.pitab[i];
    u64 sum = 0;
    for (int r = 1; r <= M.R; r++) sum += hash_rep_block(M, rec + 1 + (r - 1) * M.wpr, r, pt);
    for (int j = 0; j < nmsg; j++) sum += hash_msg(M, rec[M.fixed + j], pt);
    H[i] = sum;
  }
}

// canonical (fingerprint, auxkey) = lexicographic min over the permutations of (H[i], auxkey_i)
VSR_HD void canonical_fp(const Model& M, u64 hdr, const u64* H, u64* fp, u32* auxkey) {
  u64 bf = H[0];
  u32 ba = auxkey_of(M, hdr, M.pitab[0]);
#pragma unroll
  for (int i = 1; i < 6; i++) {
    if (i >= M.np) break;
    u64 h = H[i];
    if (h > bf) continue;
    u32 ak = auxkey_of(M, hdr, M.pitab[i]);
    if (h < bf || ak < ba) { bf = h; ba = ak; }
  }
  if (bf == 0) bf = 1;      // 0 is the empty-slot sentinel of the seen-set
  *fp = bf;
  *auxkey = ba;
}

// ---- meta word of a seen-set slot: smaller = wins the slot --------------------------------------------------------------
//   level(9) << 55 | canonical auxkey(9) << 46 | low 45 bits of the PARENT's fingerprint << 1 | taken(1)
// Every field is a property of the states themselves — no buffer index, no rank, no ordinal (an ordinal names a position in the
// parent record's bag, and the order of a bag depends on which candidate materialised the parent) — so the meta word a level's
// candidates min-merge into a slot is the same in every run, on any number of GPUs, in any scheme: (i) shallower levels always
// win (F2: the first discoverer keeps its aux values), (ii) same-level duplicates resolve to the smallest canonical auxkey, then
// to the parent with the smallest fingerprint bits, (iii) the slot IS the predecessor pointer of its state (TLCTrace): a trace
// is the chain of fingerprints walked through the table (k_trace_walk) and re-executed forwards by searching, at every step, the
// successor with the next fingerprint (k_replay_fps); there is no separate log.  `taken` is set by the one candidate that
// materialises the state in the schemes that pick the winner after the level (exact levels, MODE_REGEN): a compare-and-swap
// key -> key | 1 makes that exactly-once when several candidates carry the same 64 bits (two instances of one parent with the
// same successor do).
static const u64 META_EMPTY = ~(u64)0;
static const u64 META_TAKEN = 1;
static const u64 PFP_MASK = ((u64)1 << 45) - 1;
VSR_HD u64 meta_make(int level, u32 auxkey, u64 parent_fp) {
  return ((u64)level << 55) | ((u64)auxkey << 46) | ((parent_fp & PFP_MASK) << 1);
}
VSR_HD int meta_level(u64 m) { return (int)(m >> 55); }
VSR_HD int meta_auxkey(u64 m) { return (int)((m >> 46) & 511); }
VSR_HD u64 meta_pfp(u64 m) { return (m >> 1) & PFP_MASK; }
// sharded single-pass levels: what the generator keeps beside a successor it announced to a remote owner — the state index it wrote
// the record to speculatively, and the mask of invariants the successor violates (5 bits today: VR_APP_STATE's NoAppStateDivergence
// is bit 4; round 2 kept 2 bits and lost the masks 4, 8 and 16 of the analysis models)
constexpr u64 cand_pack(u64 idx, u32 bad) { return idx | ((u64)bad << 56); }
constexpr u64 cand_index(u64 e) { return e & (((u64)1 << 56) - 1); }
constexpr u32 cand_bad(u64 e) { return (u32)(e >> 56); }
static_assert(cand_bad(cand_pack(((u64)1 << 56) - 1, 31)) == 31 && cand_index(cand_pack(((u64)1 << 56) - 1, 31)) == ((u64)1 << 56) - 1 &&
              cand_bad(cand_pack(12345, 16)) == 16 && cand_bad(cand_pack(12345, 4)) == 4 && cand_bad(cand_pack(12345, 6)) == 6,
              "every invariant bit of every model survives the candidate entry");
// what the generator keeps beside a candidate of the schemes that materialise after the level: parent index | ordinal << 40
VSR_HD u64 origin_make(u64 pidx, int ord) { return pidx | ((u64)ord << 40); }
VSR_HD u64 origin_pidx(u64 o) { return o & (((u64)1 << 40) - 1); }
VSR_HD int origin_ord(u64 o) { return (int)(o >> 40); }

}  // namespace vsr

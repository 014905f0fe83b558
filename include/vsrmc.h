/* vsrmc.h — C ABI of libvsrmc.so, the MI355X-native explicit-state checker for VSR.tla.
 *
 * Drop-in boundary for the BFS hot path of TLC when it checks
 *   /root/reference/vsr-revisited/paper/VSR.tla  under  /root/reference/vsr-revisited/paper/VSR.cfg.
 * The reference repository holds no code of its own: the interfaces replaced are those of the TLC
 * classes the north star names — tlc2.tool.fp.FPSet, tlc2.tool.queue.StateQueue, tlc2.tool.Worker /
 * tlc2.tool.ModelChecker, tlc2.tool.impl.Tool, tlc2.tool.TLCTrace (recalled from TLC's public API) —
 * parameterised by VSR.cfg:3-37 (CONSTANTS / INIT / NEXT / VIEW / SYMMETRY / INVARIANT).
 *
 * Conventions: every entry point returns 0 on success and a negative VSRMC_E_* code on failure, with a text in
 * vsrmc_last_error() (thread-local).  Plain pointers and sizes only; the caller owns every buffer it passes, the
 * library copies in/out; handles are library-owned until *_destroy.  One host thread per handle.
 * "Wire layout" of a state record = 64-bit words [header][R replica blocks][bag words], documented in DESIGN.md
 * ("Packed record"); `off` arrays hold n+1 word offsets.
 */
#ifndef VSRMC_H
#define VSRMC_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSRMC_E_ARG (-1)         /* bad argument / buffer too small */
#define VSRMC_E_CFG (-2)         /* .cfg / .tla not accepted by the lowering (names the line) */
#define VSRMC_E_HIP (-3)         /* HIP runtime failure or no GPU */
#define VSRMC_E_EVAL (-4)        /* TLC-style evaluation error inside the spec (e.g. VSR.tla:421) */
#define VSRMC_E_REP (-5)         /* a state left the range of the packed record / a capacity was exceeded */
#define VSRMC_E_STATE (-6)       /* call not valid in the handle's current state */

typedef struct vsrmc_model vsrmc_model;
typedef struct vsrmc_fpset vsrmc_fpset;
typedef struct vsrmc_checker vsrmc_checker;

const char* vsrmc_last_error(void);
int32_t vsrmc_version(void);
/* number of visible HIP devices (0 when there is no GPU; never an error) */
int32_t vsrmc_device_count(void);

/* ---- model = (VSR.tla, VSR.cfg) lowered to a record layout + action table ------------------------------------
 * vsrmc_model_load ≙ tlc2.TLC reading `-config VSR.cfg VSR.tla` (ModelConfig + SpecProcessor); grammar accepted:
 * VSR.cfg:3-37.  tla_path may be NULL (then only the cfg is read); if given, its SHA-256 must be the VSR.tla this
 * build lowers, any other module is refused. */
typedef struct vsrmc_layout {
  int32_t replica_count, client_count, value_count, start_view_on_timer_limit;
  int32_t symmetry, invariant_mask, assume_commit_number, check_deadlock;
  int32_t words_per_replica;     /* wire layout */
  int32_t fixed_words;           /* wire layout: index of the first bag word */
  int32_t permutations;          /* |Permutations(Values)| hashed per state */
  int32_t max_bag;               /* largest message bag a record may hold */
  int32_t max_record_words;      /* wire layout upper bound */
  int32_t module;                /* 0 = VSR.tla, 1 = VR_STATE_TRANSFER.tla, 2 = VR_APP_STATE.tla */
  int32_t reserved[2];
} vsrmc_layout;

int32_t vsrmc_model_load(const char* tla_path, const char* cfg_path, vsrmc_model** out);
int32_t vsrmc_model_from_constants(int32_t replica_count, int32_t client_count, int32_t value_count,
                                   int32_t start_view_on_timer_limit, int32_t restart_empty_limit, int32_t symmetry,
                                   int32_t invariant_mask, int32_t assume_commit_number, vsrmc_model** out);
/* The second model this build lowers: analysis/03-state-transfer/VR_STATE_TRANSFER.tla under the constants of
 * VR_STATE_TRANSFER.cfg:4-7.  invariant_mask: 1 AcknowledgedWriteNotLost, 2 AcknowledgedWritesExistOnMajority, 4 NoLogDivergence,
 * 8 CommitNumberNeverHigherThanOpNumber (the shipped cfg checks 2 + 4 + 8).  vsrmc_model_load recognises either module by the
 * SHA-256 of the .tla (or, without a .tla, by the cfg's constants); every other entry point takes either model. */
int32_t vsrmc_model2_from_constants(int32_t replica_count, int32_t value_count, int32_t start_view_on_timer_limit,
                                    int32_t no_progress_change_limit, int32_t symmetry, int32_t invariant_mask, vsrmc_model** out);
/* The third model ("then 04-application-state"): analysis/04-application-state/VR_APP_STATE.tla under the constants
 * of VR_APP_STATE.cfg:4-7 (ReplicaCount <= 3).  invariant_mask as above plus 16 NoAppStateDivergence (the shipped cfg checks
 * 2 + 4 + 8 + 16 = 30).  Without a .tla, vsrmc_model_load takes an analysis cfg that lists NoAppStateDivergence for this model. */
int32_t vsrmc_model3_from_constants(int32_t replica_count, int32_t value_count, int32_t start_view_on_timer_limit,
                                    int32_t no_progress_change_limit, int32_t symmetry, int32_t invariant_mask, vsrmc_model** out);
int32_t vsrmc_model_info(const vsrmc_model* m, vsrmc_layout* out);
/* Second-hash audit (TLC: a run repeated with another -fp N polynomial).  The seen-set knows a state by 64 bits of a hash; a false
 * merge of two states would drop one of them from every count, silently, in this checker AND in an oracle keyed by the same function.
 * `seed` is xor-ed into every salt of the view hash (csrc/vsr_model.hpp): seed 0 is the function the committed fixtures were made
 * with, every other seed an independent member of the same family — fingerprints, checksums and the counter-example's tie-breaks
 * change, distinct-state counts, generated / deadlock / per-action counts must not.  Set it before a checker is created from the
 * model (a checker copies the model); a checkpoint written under one seed is refused under another. */
int32_t vsrmc_model_set_fp_seed(vsrmc_model* m, uint64_t seed);
uint64_t vsrmc_model_fp_seed(const vsrmc_model* m);
/* Init (VSR.tla:323-348) in wire layout */
int32_t vsrmc_model_init_state(const vsrmc_model* m, uint64_t* rec, int32_t cap_words, int32_t* n_words);
/* TLC-syntax text of one state (the format of state_transfer_violation_trace.txt); returns needed size in *n */
int32_t vsrmc_model_format_state(const vsrmc_model* m, const uint64_t* rec, char* buf, int64_t cap, int64_t* n);
const char* vsrmc_action_name(int32_t action_id);
void vsrmc_model_destroy(vsrmc_model* m);

/* ---- FPSet ≙ tlc2.tool.fp.FPSet (init / putBlock / containsBlock / size / close) ---------------------------------
 * Open-addressing table of 2^log2_slots 16-byte slots in HBM.  put: was_present[i] = 1 iff fp[i] was already in the
 * set (FPSet.put's return value); equal fingerprints inside one batch: exactly one reports 0. */
int32_t vsrmc_fpset_create(int32_t device, int32_t log2_slots, vsrmc_fpset** out);
int32_t vsrmc_fpset_put_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* was_present);
int32_t vsrmc_fpset_contains_batch(vsrmc_fpset* s, const uint64_t* fps, uint64_t n, uint8_t* present);
/* same, buffers already in HBM (device pointers), enqueued on `hip_stream` (a hipStream_t, may be NULL) */
int32_t vsrmc_fpset_put_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_was_present,
                                     void* hip_stream);
int32_t vsrmc_fpset_contains_batch_device(vsrmc_fpset* s, const uint64_t* d_fps, uint64_t n, uint8_t* d_present,
                                          void* hip_stream);
int32_t vsrmc_fpset_size(vsrmc_fpset* s, uint64_t* size);
void vsrmc_fpset_destroy(vsrmc_fpset* s);

/* ---- Tool.getNextStates over all actions, for a batch of states (≙ tlc2.tool.impl.Tool.getNextStates) ------------
 * in: n records (wire layout).  out: successors in (parent, ordinal) order, wire layout, plus 8 words of meta each:
 * [parent index, ordinal, action id, fingerprint, auxkey, violated-invariant mask, error code, word offset]. */
int32_t vsrmc_expand_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                           uint64_t* out_words, uint64_t out_words_cap, uint64_t* out_meta, uint64_t out_cap,
                           uint64_t* n_out, uint64_t* words_out);
/* ---- TLC state / trace import (≙ reading a TLC trace file or -dump output) ------------------------------
 * `text` holds states in TLC's value syntax: a trace expression  << [ _TEAction |-> [...], var |-> value, ... ], ... >>
 * (the format of the reference's state_transfer_violation_trace.txt), one state record  [ var |-> value, ... ]  (what
 * vsrmc_model_format_state prints), or TLC's console form ("State k: <Action ...>" + "/\ var = value" conjuncts).
 * out: wire records (bag words sorted), n+1 offsets, per state the action id named in the text (-1 if none).
 * words == NULL: only *n_states is set.  Variables the text leaves out keep their Init value. */
int32_t vsrmc_model_parse_states(const vsrmc_model* m, const char* text, uint64_t* words, uint64_t cap_words, uint64_t* off,
                                 int32_t* actions, uint64_t cap_states, uint64_t* n_states);
/* Is this sequence of states a behaviour of the model?  State 0 must be Init and every later state one of the successors the
 * GPU generates for its predecessor.  ords[i] / actions[i+1]: ordinal and action id of the step into state i+1 (ords can be fed
 * to vsrmc_model_replay); *first_bad: index of the first state that does not follow, -1 if none; *inv_mask_last: invariants
 * the last state violates. */
int32_t vsrmc_model_check_trace(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n_states,
                                uint32_t* ords, int32_t* actions, int64_t* first_bad, int32_t* inv_mask_last);
/* canonical VIEW fingerprints (TLCState.fingerPrint with VIEW + SYMMETRY) of n records */
int32_t vsrmc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off,
                                uint64_t n, uint64_t* fps, uint32_t* auxkeys);

/* ---- TLC's own fingerprint: tlc2.util.FP64 over Value.fingerPrint of the `view` value (VSR.tla:140-150) ------------
 * A characterisation mode (VSR.tla only): the seen-set never uses it.  A TLC-STYLE FP64, not a parity-checked one: everything
 * TLC-specific is recalled, not pinned — see vsr_tlcfp.hpp.  One recalled detail is known to be doubtful: a model value is serialised
 * here as MODELVALUE + its index in cfg creation order, whereas ModelValue.fingerPrint may extend with the value's UniqueString token,
 * which is assigned in interning order over ALL strings of the parsed spec (module names, operators, fields ...), not over the model
 * values alone — if so, neither the FP64 values nor the min-permutation choice under SYMMETRY can be bit-equal to TLC's until a real
 * run's token table is supplied (tools/tlc_handoff.sh is where that would be found out).  With SYMMETRY the state fingerprinted is the permuted state TLC picks (TLCStateMut.fingerPrint: the smallest under
 * Value.compareTo, variable by variable in declaration order).
 * vsrmc_tlc_fingerprint_batch: n wire records -> FP64s, computed on the GPU.
 * vsrmc_tlc_view_bytes: the byte stream the fingerprint of ONE wire record is taken over, under value permutation `permutation`
 *   (0 = identity, < permutations); *n_bytes = its length, the first `cap` bytes stored.  Host code (a diagnostic).
 * vsrmc_tlc_min_permutation: the number of the permutation whose permuted state TLC fingerprints (host code).
 * vsrmc_fp64_new / vsrmc_fp64_extend: FP64.New() / FP64.Extend(fp, byte[]) (≙ tlc2.util.FP64).
 * vsrmc_checker_tlc_level_fps: FP64 of every state of the newest level, from the frontier in HBM, sorted; out may be NULL (time only);
 *   *kernel_ms = duration of the kernel. */
int32_t vsrmc_tlc_fingerprint_batch(const vsrmc_model* m, int32_t device, const uint64_t* words, const uint64_t* off, uint64_t n,
                                    uint64_t* fps);
int32_t vsrmc_tlc_view_bytes(const vsrmc_model* m, const uint64_t* record, int32_t permutation, uint8_t* out, uint64_t cap,
                             uint64_t* n_bytes);
int32_t vsrmc_tlc_min_permutation(const vsrmc_model* m, const uint64_t* record, int32_t* permutation);
uint64_t vsrmc_fp64_new(void);
uint64_t vsrmc_fp64_extend(uint64_t fp, const uint8_t* bytes, uint64_t n);

/* ---- the checker ≙ tlc2.tool.ModelChecker + Worker.run + StateQueue + TLCTrace ---------------------------------- */
typedef struct vsrmc_options {
  int32_t device;                /* HIP device ordinal */
  int32_t table_log2;            /* seen-set slots = 2^table_log2 (16 B each) */
  uint64_t frontier_words;       /* capacity of each of the two frontier buffers, in 8-byte words */
  uint64_t frontier_states;      /* capacity of each frontier in states */
  uint64_t pending_entries;      /* capacity of the per-level pending list (24 B each; only the exact scheme fills it) */
  int32_t keep_trace;            /* ignored (kept for ABI stability): predecessor pointers live in the seen-set slots, always */
  int32_t rank, world;           /* shard of the seen-set owned by this process (world = 1: everything) */
  uint64_t trace_entries;        /* ignored (there is no separate trace log any more) */
  int32_t exact_ties;            /* 0 (default): single-pass levels — the lane that inserts a fingerprint writes the successor;
                                    a same-level VIEW collision with different aux variables (never observed) stops the run
                                    with VSRMC_E_STATE.  1: two-kernel levels (k_expand + k_materialize) that arbitrate
                                    such ties exactly like the oracle (smallest canonical auxkey wins).  Sharded runs
                                    follow the same switch. */
  int32_t filter_log2;           /* sharded single-pass runs: entries (8 B) of this rank's sent-filter = 2^filter_log2;
                                    0 = table_log2 */
  int32_t host_frontier;         /* bit 0 / bit 1: the first (levels 1, 3, 5, ...) / second (levels 2, 4, ...) record buffer is
                                    pinned HOST memory that the kernels read and write over PCIe (3 = both); refs, fingerprints,
                                    trace log and seen-set stay in HBM.  For frontiers beyond 288 GB (≙ TLC's DiskStateQueue) */
  int32_t reserved0;
  uint64_t frontier_words_b;     /* capacity of the SECOND record buffer (levels 2, 4, 6, ...); 0 = frontier_words.  Level sizes
                                    grow geometrically, so the last two levels differ by that factor: size the buffers apart */
} vsrmc_options;

typedef struct vsrmc_level_info {
  int32_t level;                 /* level just completed (Init = 1) */
  int32_t error_code;            /* 0 or a device ERR_* code */
  uint64_t frontier;             /* states expanded in this step */
  uint64_t generated;            /* successors generated (TLC "states generated") */
  uint64_t n_new;                /* new distinct states = size of the next frontier */
  uint64_t distinct;             /* total distinct states so far */
  uint64_t total_generated;
  uint64_t deadlocks;            /* expanded states without successor */
  uint64_t pending;              /* candidates that raced for a slot claimed in this level */
  uint64_t probes;               /* seen-set slots inspected */
  uint64_t words_new;            /* words reserved in the next frontier buffer (device layout, chunk slack included) */
  uint64_t record_words;         /* words of the records of the next frontier themselves */
  uint64_t max_bag;
  uint64_t viol_fp;              /* smallest fingerprint of a violating new state, ~0 if none */
  uint64_t viol_index;           /* its index in the next frontier */
  int32_t viol_mask;
  int32_t reserved0;             /* vsrmc_checker_status: levels beyond `level` that exist in the seen-set only; 0 elsewhere */
  double seconds;                /* host wall time of the step */
  double expand_ms;              /* HIP-event time of k_expand (on the checker's stream) */
  double materialize_ms;         /* HIP-event time of k_materialize */
  uint64_t act_generated[16];    /* generated successors per action id */
  uint64_t phase_cycles[8];      /* k_expand shader clocks summed over blocks: stage, enumerate, sort, apply, tail */
  uint64_t fp_xor, fp_sum;       /* vsrmc_checker_probe2 / _probe3, the levels that are never stored: xor / sum (mod 2^64) of the level's
                                  * fingerprints (a stored level: vsrmc_checker_level_checksum) */
  uint64_t limit_rechecked;      /* a probed level: instances outside the invariants' footprint (not applied by a probe pass) that sat in a tile with a record
                                  * at a representation limit — bag within R - 1 entries of its capacity, a delivery count of 3.  Not 0: those passes were
                                  * run AGAIN with every action applied, so a limit hit by such a successor is reported like anywhere else (round 5: it went
                                  * unreported); 0 on every BASELINE configuration */
} vsrmc_level_info;

void vsrmc_options_default(vsrmc_options* o);
int32_t vsrmc_checker_create(const vsrmc_model* m, const vsrmc_options* o, vsrmc_checker** out);
/* the options the checker runs with: the caller's, with every size that was left 0 replaced by what was derived from the free memory */
int32_t vsrmc_checker_options(const vsrmc_checker* c, vsrmc_options* out);
/* back to the initial state: clears the seen-set and the trace log, keeps every allocation (≙ a fresh TLC run) */
int32_t vsrmc_checker_reset(vsrmc_checker* c);
/* expand the newest level by one BFS step; info->n_new == 0 means the search is exhausted */
int32_t vsrmc_checker_step(vsrmc_checker* c, vsrmc_level_info* info);
/* sorted fingerprints of the newest level */
int32_t vsrmc_checker_level_fps(vsrmc_checker* c, uint64_t* out, uint64_t cap, uint64_t* n);
int32_t vsrmc_checker_tlc_level_fps(vsrmc_checker* c, uint64_t* out, uint64_t cap, uint64_t* n, double* kernel_ms);
/* xor, sum (mod 2^64) and number of the fingerprints of the newest level, computed on the device (order-independent checksums
 * of a level's fingerprint SET: what the whole-workload fixtures of the CPU oracle hold per level) */
int32_t vsrmc_checker_level_checksum(vsrmc_checker* c, uint64_t* fp_xor, uint64_t* fp_sum, uint64_t* n_states);
/* the newest level in wire layout (≙ StateQueue.sDequeue(int) without removal) */
int32_t vsrmc_checker_frontier(vsrmc_checker* c, uint64_t* words, uint64_t cap_words, uint64_t* off, uint64_t cap_states,
                               uint64_t* n);
/* the states of the newest level in which at least one instance of an action of `action_mask` (bit a = action id a, the order of
 * `Next`, VSR.tla:896-918) is enabled — at most max_states of them, wire layout; *n_matching = how many there are in all
 * (≙ TLC's per-action coverage, as a filter) */
/* MEASUREMENT (not a product path; tools/bench_layout.py): one staging pass of k_expand's tile loop over the newest stored level, layout 0 =
 * the records as they are (refs + variable-length records), 1 = the same level as fixed-stride columns (SURVEY §8a row a1's SoA; a transposed copy is made
 * first, untimed).  Both fill the same LDS tile and read it back once.  *ms_per_pass = HIP-event time of one pass (mean of `reps`), *bytes_per_pass = what it reads.
 * Layouts 2-6 (round 6) are layout 0 with the tile loop software-pipelined / more resident blocks: 2 = refs one tile ahead, 3 = refs two and words one tile
 * ahead, 4 = layout 0 at eight blocks per CU, 5 / 6 = 2 / 3 at eight blocks per CU, 7 / 10 = layout 0 drawing 4 / 16 tiles from the cursor at a time, 8 / 9 =
 * 3 / 6 drawing 4 (csrc/vsr_bench_layout.hpp: k_stage_pipe). */
int32_t vsrmc_checker_bench_staging(vsrmc_checker* c, int32_t layout, int32_t reps, double* ms_per_pass, uint64_t* bytes_per_pass);
int32_t vsrmc_checker_select(vsrmc_checker* c, uint32_t action_mask, uint64_t max_states, uint64_t* words, uint64_t cap_words,
                             uint64_t* off, uint64_t* n_states, uint64_t* n_matching);
/* ≙ TLCTrace.getTrace: the path Init .. state `index` of the NEWEST level (level must be the current one); records in wire
 * layout, one action id per state (0 = Initial predicate).  The path is a function of the state alone: every state's slot in the
 * seen-set names its parent — of all states of the previous level that produce it the one with the smallest (canonical auxkey,
 * fingerprint) — and the step between two states of the path is the enabled instance with the smallest ordinal that leads from
 * one to the other; so the same counter-example comes back in every run, on any number of GPUs, in either level scheme. */
int32_t vsrmc_checker_trace(vsrmc_checker* c, int32_t level, uint64_t index, uint64_t* words, uint64_t cap_words,
                            uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states);
/* the same for any state of any completed level, addressed by its fingerprint */
int32_t vsrmc_checker_trace_fp(vsrmc_checker* c, int32_t level, uint64_t fp, uint64_t* words, uint64_t cap_words,
                               uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states);
/* forward half of TLCTrace.getTrace for a path given by the fingerprints of its states (fps[0] = Init's): at every step the
 * successor with the next fingerprint is taken — what a walk through the seen-set (vsrmc_checker_lookup) yields */
int32_t vsrmc_model_replay_fps(const vsrmc_model* m, int32_t device, const uint64_t* fps, int32_t n_fps, uint64_t* words,
                               uint64_t cap_words, uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states);
/* the same for a path of ordinals (simulation walks): re-execute `nsteps` ordinals from Init */
int32_t vsrmc_model_replay(const vsrmc_model* m, int32_t device, const uint32_t* ords, int32_t nsteps, uint64_t* words,
                           uint64_t cap_words, uint64_t* off, int32_t* actions, uint64_t cap_states, uint64_t* n_states);
/* one step of a trace walk through this checker's (shard of the) seen-set — what a walk that crosses ranks is made of.
 * by_low_bits = 0: the slot of fingerprint `key`; 1: the slot of the level-`level` state whose fingerprint ends in the 45 bits
 * `key` — *found = the number of such states in this shard: more than one (over all shards) is an ambiguous predecessor pointer,
 * which the walks report instead of following the first match.  meta = level(9) << 55 | auxkey(9) << 46 | parent fingerprint
 * bits(45) << 1 | taken(1). */
int32_t vsrmc_checker_lookup(vsrmc_checker* c, uint64_t key, int32_t level, int32_t by_low_bits, int32_t* found, uint64_t* fp,
                             uint64_t* meta);
/* index of fingerprint `fp` in the newest level (~0 if absent) */
int32_t vsrmc_checker_find_fp(vsrmc_checker* c, uint64_t fp, uint64_t* index);
void vsrmc_checker_destroy(vsrmc_checker* c);

/* Probe level: expand the newest level without storing its successors — invariants are evaluated on every successor that is
 * not a state of an earlier level, nothing is inserted or written, so the level costs no frontier memory; the search cannot
 * continue afterwards.  Finds a violation one level beyond what memory can hold.  Also valid right after a step that failed
 * with "frontier full".  Order of evaluation (the result is that of the sentence above): every enabled instance is enumerated and
 * counted; only the actions that write a variable the invariants read are applied (VSR.tla: rep_log, aux_client_acked — a successor
 * of any other action has the verdict of its parent, which passed); of their successors the invariants are evaluated first, and
 * only a successor that fails one is fingerprinted and looked up in the seen-set.  "Which passed" is the premise: when a level this
 * checker committed held a violating state (a caller that steps on after a reported violation), the probe passes apply EVERY action.
 * Representation errors (ERR_REP_*) of an action outside the footprint are not raised by a footprint pass — its instances are counted,
 * not applied.  `probes` counts those lookups only.  info: level (the probed one), generated, viol_fp / viol_mask, viol_index = index of the violator's
 * PARENT in the newest level, pending = violating successors seen.  vsrmc_checker_probe_trace: Init .. violator.
 * On a sharded checker (world > 1) the call probes this rank's part of the newest level against this rank's part of the seen-set and
 * resolves nothing: a violating successor owned by another rank may be a state that rank has seen, so the candidates
 * (vsrmc_checker_probe_candidates) must be shown to their owners (vsrmc_checker_seen_batch there) — sharded.py: ShardedChecker.probe. */
int32_t vsrmc_checker_probe(vsrmc_checker* c, vsrmc_level_info* info);
/* Two levels beyond the last materialised one: level L+1 becomes a VIRTUAL level (fingerprints claimed, invariants checked, exact
 * count, no records), then the newest level is expanded a second time in slices — the successors that won their slot are written
 * to the idle next buffer and at once expanded in probe mode (level L+2), then dropped.  Costs one extra expansion of the newest
 * level and no memory beyond a slice.  virt: level L+1 (n_new exact); probe: level L+2.  A violation in either is reported there
 * (viol_index = index in the newest level of the violator's parent resp. grandparent); vsrmc_checker_probe_trace gives the
 * counter-example.  The search cannot continue afterwards. */
int32_t vsrmc_checker_probe2(vsrmc_checker* c, vsrmc_level_info* virt, vsrmc_level_info* probe);
/* the same one level deeper: level L+1 virtual, level L+2 streamed through scratch buffers (inserted into the seen-set, exact
 * count, never kept), level L+3 probed.  Level L is expanded twice, L+1 and L+2 once; scratch buffers of a quarter of the record
 * buffers' size are allocated for the call.  virt2->pending = (slices of level L) << 32 | sub-slices of level L+1.  A violation
 * in level L+1 or L+2 is reported in that level's info (a violating level is still completed; nothing deeper is reported);
 * vsrmc_checker_probe_trace reconstructs the counter-example in every case. */
int32_t vsrmc_checker_probe3(vsrmc_checker* c, vsrmc_level_info* virt1, vsrmc_level_info* virt2, vsrmc_level_info* probe);
/* The general form of the two calls above (≙ TLC going on with DiskStateQueue / DiskFPSet when the frontier outgrows memory; here
 * nothing leaves the HBM): ONE MORE LEVEL beyond the record buffers per call.  The first call after the last vsrmc_checker_step makes
 * level L+1 a virtual level (inserted->level = L+1, probed->level = 0).  Every further call descends from the newest materialised
 * level L — regenerating levels L+1 .. L+j-1 slice inside slice from the seen-set's predecessor keys, each state exactly once —
 * inserts level L+j (exact count, per-action counts, checksums, invariants; its records pass through a scratch buffer) and probes level
 * L+j+1 (probed->level).  The search rolls on past the memory horizon at about 2.2 x the expansions of a stored BFS until the
 * seen-set is full, a level comes back empty (inserted->n_new == 0: exhausted) or an invariant fails (viol_mask of whichever info;
 * vsrmc_checker_probe_trace reconstructs the counter-example).  inserted->pending = k_expand launches of the pass, ->materialize_ms =
 * kernel time spent regenerating, ->words_new = (slices of level L) << 32 | slices of the levels below it.  vsrmc_checker_step is
 * refused from the first call on (the seen-set holds levels that have no frontier) until the search has been RE-BASED: when the levels shrink again
 * (the analysis models after their peak) and the newest seen-set-only level and the one predicted after it fit the record buffers, vsrmc_checker_advance
 * spends one descent on regenerating that level INTO the idle record buffer (*what = 3: a = the level, unchanged figures) and the rest of the run is
 * ordinary stored levels.  vsrmc_checker_save works between any two calls; vsrmc_checker_reset starts over. */
int32_t vsrmc_checker_deepen(vsrmc_checker* c, vsrmc_level_info* inserted, vsrmc_level_info* probed);
/* One unit of progress of the AUTOMATIC level scheme — no level numbers and no sizes from the caller: an ordinary BFS level while
 * the next one is predicted to fit the idle record buffer (*what = 1, a = its info), otherwise vsrmc_checker_deepen (*what = 2,
 * a = the inserted level, b = the probed one or b->level == 0; *what = 3: no new level, the deep search was re-based onto level a->level,
 * see vsrmc_checker_deepen).  The prediction: a level grows by at most the factor the last one
 * grew by (the factor falls from level to level in these models, tests/golden/oracle_levels_*.json) and by at most one bag entry
 * per record.  vsrmc_check is the loop over this call; vsrmc_options with table_log2 = 0 / frontier_words = 0 / frontier_states = 0 /
 * pending_entries = 0 are sized from the free memory of the device. */
int32_t vsrmc_checker_advance(vsrmc_checker* c, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what);
/* No fatal mispredictions.  (i) A level vsrmc_checker_advance stored although it did not fit ("frontier full": the prediction above was wrong, or the
 * caller's own vsrmc_checker_step ran out of buffer) is not lost: k_expand goes on claiming, counting and checking after the buffers are exhausted, so
 * the seen-set holds the whole level — the NEXT vsrmc_checker_advance (or the same one) keeps it as a seen-set-only level (*what = 2, its figures and
 * checksums exact) and the search rolls on.  (ii) The seen-set: *state = 0 there is room for the next level, 1 = it has just been re-hashed into a
 * table of twice the slots (device memory permitting; vsrmc_checker_options shows the new table_log2), 2 = it is more than 85 % full and cannot grow:
 * the search is incomplete at the depth reached.  vsrmc_check asks before every unit of progress; loops over vsrmc_checker_advance should too. */
int32_t vsrmc_checker_room(vsrmc_checker* c, int32_t* state);
/* the (fingerprint, key) pairs of the violating successors the last vsrmc_checker_probe saw and did not find in this checker's
 * seen-set (duplicates included; *n = their number; pairs == NULL asks for the number only).  Sharded runs show them to their owners. */
int32_t vsrmc_checker_probe_candidates(vsrmc_checker* c, uint64_t* pairs, uint64_t cap_pairs, uint64_t* n);
/* The distinct violating STATES of the level the last vsrmc_checker_probe / _deepen / _advance probed (unsharded checkers): the fingerprints,
 * ascending, of the violating successors that are states of no earlier level — viol_fp of the probe's info is fps[0].  Lets a caller ask whether
 * a KNOWN violating state (the last state of the reference's state_transfer_violation_trace.txt:556-577) is among the ones the search met, not
 * only which one it reports.  fps == NULL asks for the number only. */
int32_t vsrmc_checker_probe_violators(vsrmc_checker* c, uint64_t* fps, uint64_t cap, uint64_t* n);
/* seen[i] = 1 if fps[i] is in this checker's seen-set as a state of a level below `level` (host arrays) */
int32_t vsrmc_checker_seen_batch(vsrmc_checker* c, const uint64_t* fps, uint64_t n, int32_t level, uint8_t* seen);
int32_t vsrmc_checker_probe_trace(vsrmc_checker* c, uint64_t* words, uint64_t cap_words, uint64_t* off, int32_t* actions,
                                  uint64_t cap_states, uint64_t* n_states);
/* ≙ TLCTrace.getTrace for a violating state of the CALLER's choice: the counter-example that ends in `fp`, one of vsrmc_checker_probe_violators (unsharded
 * checkers).  vsrmc_checker_probe_trace walks to the violator with the smallest fingerprint; the reference's own counter-example
 * (state_transfer_violation_trace.txt:555-577) ends in another violating state of the same level — this call prints the path the search took to THAT state,
 * so that the two action sequences can be laid side by side.  Same layout as vsrmc_checker_trace. */
int32_t vsrmc_checker_trace_to_violator(vsrmc_checker* c, uint64_t fp, uint64_t* words, uint64_t cap_words, uint64_t* off, int32_t* actions,
                                        uint64_t cap_states, uint64_t* n_states);

/* ≙ TLC's checkpoints (ModelChecker.checkpoint → FPSet.beginChkpt/commitChkpt, StateQueue and TLCTrace checkpoints; `-recover`):
 * one file holds the search between two levels — the occupied seen-set slots, the newest stored frontier, what the automatic level scheme
 * has learnt and, once the search has gone beyond the record buffers (vsrmc_checker_deepen), the descriptors of the levels that exist in
 * the seen-set only: a deep search is saved between two passes and recovered to make every later pass as the uninterrupted run would.  It is written
 * to <path>.tmp and renamed.  vsrmc_checker_load creates a checker with `o` (capacities and table size may differ from the
 * run that saved; the model constants may not) and continues where the checkpoint stopped.  Unsharded checkers only. */
int32_t vsrmc_checker_save(vsrmc_checker* c, const char* path);
/* where the search stands (after create / reset / load / step / advance): level = the newest STORED level, n_new = its states, reserved0 = the
 * number of levels beyond it that are complete in the seen-set only (depth = level + reserved0), frontier = states of the deepest complete
 * level, distinct, total_generated (both over every complete level) */
int32_t vsrmc_checker_status(vsrmc_checker* c, vsrmc_level_info* info);
int32_t vsrmc_checker_load(const vsrmc_model* m, const vsrmc_options* o, const char* path, vsrmc_checker** out);

/* ≙ tlc2.tool.ModelChecker.runTLC / Worker.run until the queue is empty, an invariant fails, a bound is hit or the spec raises an
 * evaluation error: vsrmc_checker_step in a loop.  stop_reason: 0 exhausted, 1 invariant violated (last->viol_*), 2 max_depth,
 * 3 max_seconds; errors come back as the return code, with `last` describing the last completed level. */
int32_t vsrmc_check(vsrmc_checker* c, int32_t max_depth, double max_seconds, int32_t* stop_reason, vsrmc_level_info* last);

/* ---- StateQueue ≙ tlc2.tool.queue.StateQueue (sEnqueue(TLCState[]) / sDequeue(int) / size): a FIFO of state records kept in
 * HBM; batches cross the boundary in wire layout (words + n+1 offsets).  The checker keeps its own two frontier buffers; this
 * handle is the stand-alone queue for callers that drive the loop themselves (e.g. with vsrmc_expand_batch). */
typedef struct vsrmc_queue vsrmc_queue;
int32_t vsrmc_queue_create(int32_t device, uint64_t capacity_words, uint64_t capacity_states, vsrmc_queue** out);
int32_t vsrmc_queue_enqueue_batch(vsrmc_queue* q, const uint64_t* words, const uint64_t* off, uint64_t n);
int32_t vsrmc_queue_dequeue_batch(vsrmc_queue* q, uint64_t max_states, uint64_t* words, uint64_t cap_words, uint64_t* off,
                                  uint64_t* n);
int32_t vsrmc_queue_size(vsrmc_queue* q, uint64_t* n_states);
void vsrmc_queue_destroy(vsrmc_queue* q);

/* ---- simulation mode ≙ `tlc2.TLC -simulate` (the reference README:22 recommends it for the state-transfer defect) -------------
 * n_walkers random walks run concurrently (one GPU lane each): start at Init, up to max_depth steps chosen uniformly among
 * the enabled (action, binding) instances, invariants checked after every step, restart at the depth limit or in a terminal
 * state.  Stops at the first violation or after max_seconds.  The violating walk comes back as ordinals; pass them to
 * vsrmc_model_replay for the states and action names. */
typedef struct vsrmc_sim_result {
  int32_t found;                 /* 1 = invariant violated, 2 = evaluation / representation error on a walk, 0 = time-out */
  int32_t viol_mask;             /* violated invariants (found = 1) or device error code (found = 2) */
  int32_t viol_steps;            /* number of steps of the reported walk (trace = viol_steps + 1 states) */
  int32_t reserved;
  uint64_t steps, walks;         /* totals over all walkers */
  double seconds;
  uint32_t ords[512];
} vsrmc_sim_result;
int32_t vsrmc_simulate(const vsrmc_model* m, int32_t device, uint32_t n_walkers, int32_t max_depth, uint64_t seed,
                       double max_seconds, vsrmc_sim_result* out);

/* ---- sharded seen-set (≙ tlc2.tool.fp.MultiFPSet across GPUs): the phases of one BFS level -----------------------
 * world ranks, one per GPU; owner(fp) = ((fp >> 40) & 0xFFFFFF) % world.  The caller (vsr_tlaplus_amd/sharded.py over
 * torch.distributed / RCCL) owns the exchange buffers and moves them between ranks; every pointer is a device pointer.
 *   1. vsrmc_shard_expand       expand the local frontier; local-owner candidates are claimed at once, the others are
 *                               bucketed per owner as (fp, key) pairs in io->cand_send; returns the bucket sizes
 *   2. [all-to-all of the buckets]
 *   3. vsrmc_shard_claim        claim the received candidates in the local shard, then answer each with 1 = "won its slot"
 *   4. [all-to-all of the verdict bytes, back to the generators]
 *   5. vsrmc_shard_materialize  every winner (local owner or remote verdict) is rebuilt into THIS rank's next frontier:
 *                               records stay with their generator, only candidates and verdicts cross ranks
 *   6. vsrmc_shard_count        valid states / index range of the local next frontier; the caller compares the ranks and,
 *                               when they are out of balance, moves records in bulk:
 *      vsrmc_shard_export       copy the valid records of an index window into send streams and invalidate them here
 *      [all-to-all of the streams]
 *      vsrmc_shard_append       per received stream: copy into the next frontier, publish refs / fps
 *   7. vsrmc_shard_commit       swap the frontiers; local statistics (the caller all-reduces them)
 * With exact_ties = 0 (default) the levels are single-pass: in step 1 a successor owned by another rank that passes this
 * rank's sent-filter (exact repeats are dropped there) is written to the local next frontier SPECULATIVELY and announced;
 * in step 3 the candidate that inserts the fingerprint wins; step 5 only withdraws the announced successors that lost.
 * With exact_ties = 1 step 5 rebuilds the winners (k_materialize) and same-level ties follow the oracle's rule.
 *
 * Replicated phase: a sharded checker starts with Init on EVERY rank.  While the frontier is small the ranks explore whole
 * levels on their own, without any exchange (vsrmc_shard_local_step = vsrmc_checker_step on this rank's private copy;
 * every rank sees the same state sets, trace keys carry the local rank).  vsrmc_shard_partition ends that phase: each
 * rank keeps the states of the current frontier it owns and the sharded protocol above takes over.  (The early
 * fingerprints then sit in every rank's table, owners included, which is all the protocol needs.) */
int32_t vsrmc_shard_local_step(vsrmc_checker* c, vsrmc_level_info* info);
int32_t vsrmc_shard_partition(vsrmc_checker* c, uint64_t* n_kept);
/* after vsrmc_shard_commit / _local_step: the largest bag (vsrmc_level_info.max_bag) of the new level over ALL ranks, so that the next
 * expansion can size its LDS record slots for the level (optional; without it the format's worst case is used) */
int32_t vsrmc_shard_set_max_bag(vsrmc_checker* c, uint64_t max_bag);
typedef struct vsrmc_shard_io {
  uint64_t* cand_send;           /* [world][cand_cap][2]  (fp, key) */
  uint64_t cand_cap;             /* entries per owner */
} vsrmc_shard_io;
int32_t vsrmc_shard_expand(vsrmc_checker* c, const vsrmc_shard_io* io, uint64_t* cand_counts);
int32_t vsrmc_shard_claim(vsrmc_checker* c, const uint64_t* d_cand_recv, uint64_t n, uint8_t* d_verdict);
int32_t vsrmc_shard_materialize(vsrmc_checker* c, const vsrmc_shard_io* io, const uint8_t* d_verdict_in /* [world][cand_cap] */);
int32_t vsrmc_shard_count(vsrmc_checker* c, uint64_t* n_valid, uint64_t* n_range);
/* streams: d_words (device layout records, contiguous), d_off (ref = word offset in d_words << 8 | length), d_fp */
int32_t vsrmc_shard_export(vsrmc_checker* c, uint64_t first, uint64_t n, uint64_t* d_words, uint64_t words_cap, uint64_t* d_off,
                           uint64_t* d_fp, uint64_t cap, uint64_t* n_out, uint64_t* words_out);
int32_t vsrmc_shard_append(vsrmc_checker* c, const uint64_t* d_words, uint64_t n_words, const uint64_t* d_off,
                           const uint64_t* d_fp, uint64_t n);
int32_t vsrmc_shard_commit(vsrmc_checker* c, vsrmc_level_info* info);

/* ---- the level loop of a sharded run, natively (csrc/vsr_shard_loop.hpp) ≙ TLC's distributed mode (TLCServer / TLCWorker / FPSet servers)
 * The loop sequences the phases above on one rank and talks to the other ranks through a vsrmc_comm: two functions and a flag.
 *   alltoallv  element p of every array = peer p; `send + send_offs[p] * elem_bytes` holds send_counts[p] elements for p, what p sends
 *              arrives at `recv + recv_offs[p] * elem_bytes` (recv_counts[p] elements, known to the caller); nothing is ever addressed to
 *              oneself.  host_buffers = 0: the pointers are DEVICE memory and `stream` is the hipStream_t to order the transfer on (the
 *              function returns when it has completed); host_buffers = 1: HOST memory, stream = NULL (the loop stages the buckets).
 *   allgather  `bytes` of host memory from every rank, rank order, blocking.
 * Return 0, or non-zero on failure.  vsrmc_comm_rccl_* is the RCCL transport (grouped ncclSend / ncclRecv over xGMI; librccl is
 * dlopen'ed on first use): rank 0 makes the 128-byte id, the caller's bootstrap (torch.distributed, MPI, a file) hands it to every rank. */
typedef struct vsrmc_comm {
  void* ctx;
  int32_t rank, world;
  int32_t host_buffers;
  int32_t reserved0;
  int32_t (*alltoallv)(void* ctx, const void* send, const uint64_t* send_counts, const uint64_t* send_offs, void* recv,
                       const uint64_t* recv_counts, const uint64_t* recv_offs, uint32_t elem_bytes, void* stream);
  int32_t (*allgather)(void* ctx, const void* send, void* recv, uint32_t bytes);
} vsrmc_comm;
int32_t vsrmc_comm_rccl_unique_id(uint8_t* id128);
int32_t vsrmc_comm_rccl_create(const uint8_t* id128, int32_t rank, int32_t world, int32_t device, vsrmc_comm** out);
void vsrmc_comm_rccl_destroy(vsrmc_comm* comm);

typedef struct vsrmc_shard_loop vsrmc_shard_loop;
/* c: a sharded checker (vsrmc_options.world / rank, exact_ties = 0) in its initial state; cand_cap: (fp, key) candidates per peer and
 * level; rec_cap / rec_words_cap: records / words one rebalancing round may move; replicate_below: levels with fewer new states are
 * explored by every rank on its own (0 / 1: sharded from Init on).  The comm struct is copied; its ctx must outlive the loop. */
int32_t vsrmc_shard_loop_create(vsrmc_checker* c, const vsrmc_comm* comm, uint64_t cand_cap, uint64_t rec_cap, uint64_t rec_words_cap,
                                uint64_t replicate_below, vsrmc_shard_loop** out);
void vsrmc_shard_loop_destroy(vsrmc_shard_loop* l);
/* one BFS level on every rank (collective): global = the level's figures over all ranks (viol_fp = smallest violating fingerprint of
 * any rank, ~0 if none; n_new == 0: exhausted), local = this rank's own */
int32_t vsrmc_shard_loop_step(vsrmc_shard_loop* l, vsrmc_level_info* global, vsrmc_level_info* local);
/* ≙ vsrmc_check: stop_reason 0 exhausted, 1 invariant violated, 2 max_depth */
int32_t vsrmc_shard_loop_run(vsrmc_shard_loop* l, int32_t max_depth, int32_t stop_on_violation, int32_t* stop_reason, vsrmc_level_info* last);
/* the seen-set shards before the next vsrmc_shard_loop_advance (collective once the run is sharded): *state = 0 room on every rank, 2 = some rank's shard
 * is more than 85 % full — every rank learns it in the same call and all of them stop together ("incomplete at depth N"); loops over
 * vsrmc_shard_loop_advance ask before every call, vsrmc_shard_loop_run does.  Round 6: the winner set of a sharded deep search counts too — it is re-hashed
 * into twice the slots here when the next level is expected to take it past 60 % (while the device has the memory) and answers 2 when that level cannot fit;
 * after an answer of 2 vsrmc_last_error() names the set and its fill on this rank (the call itself returns 0) */
int32_t vsrmc_shard_loop_room(vsrmc_shard_loop* l, int32_t* state);
/* Since round 6 a sharded level of 2^20 states per rank or more runs in SLICES with the exchange overlapped: the all-to-all of slice k's (fp, key)
 * candidates, the owners' claims, the verdict bytes and the withdrawal of the losers run on a second stream and a second set of buckets while k_expand of
 * slice k + 1 runs (VSRMC_OVERLAP=0: the sequential level of rounds 2-5; VSRMC_OVERLAP_MIN_STATES / _MIN_SLICES: when and how finely).  What a level
 * leaves does not depend on the slicing.  *levels / *slices: how many levels ran that way so far, in how many slices. */
int32_t vsrmc_shard_loop_overlap_stats(vsrmc_shard_loop* l, uint64_t* levels, uint64_t* slices);
/* ≙ TLC's checkpoints for a sharded run (collective; between two vsrmc_shard_loop_advance calls — also once the search has gone beyond the ranks' record
 * buffers: the seen-set-only levels' descriptors and each rank's winner set travel in its shard file).  Every rank writes <prefix>.rank<r>of<w> (its checker:
 * vsrmc_checker_save) and <prefix>.rank<r>of<w>.loop (the loop's own state), in two phases: a failure on any rank leaves the previous checkpoint whole.
 * Recover: vsrmc_checker_load of the rank's own shard file, then vsrmc_shard_loop_restore over it (every rank must have loaded the same depth). */
int32_t vsrmc_shard_loop_save(vsrmc_shard_loop* l, const char* prefix);
int32_t vsrmc_shard_loop_restore(vsrmc_checker* c, const vsrmc_comm* comm, uint64_t cand_cap, uint64_t rec_cap, uint64_t rec_words_cap, const char* prefix,
                                 vsrmc_shard_loop** out);
/* vsrmc_checker_deepen / vsrmc_checker_advance for a sharded run (collective; figures over all ranks): levels beyond the ranks' record
 * buffers live in the ranks' seen-sets only and are regenerated from the newest stored level; every pass of the descent is the
 * protocol of a sharded level with other sources and targets (csrc/vsr_shard_loop.hpp).  advance: an ordinary sharded level while EVERY
 * rank predicts that its part of the next one fits, else deepen.  vsrmc_shard_loop_run is the loop over advance (stop_reason 4: a
 * rank's seen-set is 85 % full). */
int32_t vsrmc_shard_loop_deepen(vsrmc_shard_loop* l, vsrmc_level_info* inserted, vsrmc_level_info* probed);
int32_t vsrmc_shard_loop_advance(vsrmc_shard_loop* l, vsrmc_level_info* a, vsrmc_level_info* b, int32_t* what);
/* the fingerprints of the counter-example of a violation a probe pass found (collective), Init first; *n = its length */
int32_t vsrmc_shard_loop_probe_trace_fps(vsrmc_shard_loop* l, uint64_t* fps, int32_t cap, int32_t* n);
int32_t vsrmc_shard_loop_status(vsrmc_shard_loop* l, int32_t* level, uint64_t* distinct, uint64_t* n_frontier, int32_t* replicated,
                                uint64_t* viol_fp, int32_t* viol_mask, int32_t* viol_level, uint64_t* moved, uint64_t* bytes_sent);
/* TLCTrace.getTrace across ranks (collective): fps[0 .. level) = fingerprints of the path Init -> the level-`level` state `fp`;
 * vsrmc_model_replay_fps re-executes it on one GPU */
int32_t vsrmc_shard_loop_trace_fps(vsrmc_shard_loop* l, int32_t level, uint64_t fp, uint64_t* fps);

#ifdef __cplusplus
}
#endif
#endif

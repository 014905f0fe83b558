"""Host-side mirror of the TLC interfaces this path replaces, over the C ABI (include/vsrmc.h):

    Model         tlc2.tool.impl.Tool / ModelConfig   — (VSR.tla, VSR.cfg) lowered; getNextStates(batch)
    FPSet         tlc2.tool.fp.FPSet                  — put / contains / putBlock / containsBlock / size / close
    ModelChecker  tlc2.tool.ModelChecker + Worker.run + StateQueue + TLCTrace — level-synchronous BFS on the GPU

Everything below is plumbing: all model semantics, hashing and set operations run in the HIP kernels.
"""
import ctypes as C
import os

import numpy as np

from . import capi
from .capi import VsrmcError, check

ACTION_NAMES = ["Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC",
                "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest",
                "ReceivePrepareMsg", "ReceivePrepareOkMsg", "ExecuteOp", "SendGetState", "ReceiveGetState",
                "ReceiveNewState"]          # Next order, VSR.tla:896-913

INV_ACK_NOT_LOST = 1           # AcknowledgedWriteNotLost           VSR.tla:945-950
INV_ACK_ON_MAJORITY = 2        # AcknowledgedWritesExistOnMajority  VSR.tla:937-943


def _p(a):
    return C.c_void_p(a.ctypes.data)


class Model:
    """The lowered (VSR.tla, VSR.cfg) pair."""

    def __init__(self, handle):
        self._h = handle
        lay = capi.Layout()
        check(capi.load().vsrmc_model_info(self._h, C.byref(lay)))
        self.layout = lay

    @classmethod
    def load(cls, cfg_path, tla_path=None):
        """≙ `tlc2.TLC -config VSR.cfg VSR.tla`: reads the cfg grammar of VSR.cfg:1-39."""
        h = C.c_void_p()
        check(capi.load().vsrmc_model_load(tla_path.encode() if tla_path else None, cfg_path.encode(), C.byref(h)))
        return cls(h)

    @classmethod
    def from_constants(cls, R=3, C_=1, n=2, L=2, restart_limit=0, symmetry=True, invariant_mask=1,
                       assume_commit_number=False):
        h = C.c_void_p()
        check(capi.load().vsrmc_model_from_constants(R, C_, n, L, restart_limit, int(symmetry), invariant_mask,
                                                     int(assume_commit_number), C.byref(h)))
        return cls(h)

    @classmethod
    def second_model(cls, R=3, n=2, L=2, no_progress_limit=0, symmetry=False, invariant_mask=14):
        """analysis/03-state-transfer/VR_STATE_TRANSFER.tla under the constants of its cfg (the shipped one: 3, {v1,v2}, 2, 0;
        INVARIANT AcknowledgedWritesExistOnMajority, NoLogDivergence, CommitNumberNeverHigherThanOpNumber = mask 14)."""
        h = C.c_void_p()
        check(capi.load().vsrmc_model2_from_constants(R, n, L, no_progress_limit, int(symmetry), invariant_mask, C.byref(h)))
        return cls(h)

    @classmethod
    def third_model(cls, R=3, n=2, L=2, no_progress_limit=0, symmetry=False, invariant_mask=30):
        """analysis/04-application-state/VR_APP_STATE.tla under the constants of its cfg (the shipped one: 3, {a,b}, 2, 0; INVARIANT
        AcknowledgedWritesExistOnMajority, NoLogDivergence, NoAppStateDivergence, CommitNumberNeverHigherThanOpNumber = mask 30)."""
        h = C.c_void_p()
        check(capi.load().vsrmc_model3_from_constants(R, n, L, no_progress_limit, int(symmetry), invariant_mask, C.byref(h)))
        return cls(h)

    def set_fp_seed(self, seed):
        """Second-hash audit (TLC: a rerun under another -fp N): `seed` is xor-ed into every salt of the view hash.  Seed 0 is the
        function of the committed fixtures; under any other seed fingerprints and checksums change, counts must not.  Checkers
        created afterwards see it (a checker copies the model).  Returns self."""
        check(capi.load().vsrmc_model_set_fp_seed(self._h, C.c_uint64(int(seed) & (2 ** 64 - 1))))
        return self

    @property
    def fp_seed(self):
        return int(capi.load().vsrmc_model_fp_seed(self._h))

    def init_state(self):
        out = np.zeros(256, dtype=np.uint64)
        n = C.c_int32()
        check(capi.load().vsrmc_model_init_state(self._h, _p(out), 256, C.byref(n)))
        return out[: n.value].copy()

    def format_state(self, rec):
        rec = np.ascontiguousarray(rec, dtype=np.uint64)
        n = C.c_int64()
        check(capi.load().vsrmc_model_format_state(self._h, _p(rec), None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        check(capi.load().vsrmc_model_format_state(self._h, _p(rec), buf, n.value, C.byref(n)))
        return buf.value.decode()

    def get_next_states(self, words, off, device=0, cap_succ=None, cap_words=None):
        """Tool.getNextStates over all actions for a batch of states.
        -> list of dict(parent, ordinal, action, fp, auxkey, inv, err, words)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        cap_succ = cap_succ or max(64, 48 * n)
        cap_words = cap_words or cap_succ * int(self.layout.max_record_words)
        ow = np.zeros(cap_words, dtype=np.uint64)
        om = np.zeros(8 * cap_succ, dtype=np.uint64)
        n_out, w_out = C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_expand_batch(self._h, device, _p(words), _p(off), n, _p(ow), cap_words, _p(om), cap_succ,
                                             C.byref(n_out), C.byref(w_out)))
        out = []
        offs = [int(om[8 * k + 7]) for k in range(n_out.value)] + [w_out.value]
        for k in range(n_out.value):
            m = om[8 * k: 8 * k + 8]
            out.append(dict(parent=int(m[0]), ordinal=int(m[1]), action=int(m[2]), fp=int(m[3]), auxkey=int(m[4]),
                            inv=int(m[5]), err=int(m[6]), words=ow[offs[k]: offs[k + 1]].copy()))
        return out

    def fingerprints(self, words, off, device=0):
        """TLCState.fingerPrint (VIEW + SYMMETRY) of a batch of states -> (fps, auxkeys)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        fps = np.zeros(n, dtype=np.uint64)
        aks = np.zeros(n, dtype=np.uint32)
        check(capi.load().vsrmc_fingerprint_batch(self._h, device, _p(words), _p(off), n, _p(fps), _p(aks)))
        return fps, aks

    def tlc_fingerprints(self, words, off, device=0):
        """TLC's own fingerprint (tlc2.util.FP64 over Value.fingerPrint of `view`; [TLC-RECALLED], csrc/vsr_tlcfp.hpp) of a batch of states,
        computed on the GPU.  With SYMMETRY: of the permuted state TLC picks (tlc_min_permutation)."""
        words = np.ascontiguousarray(words, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        n = len(off) - 1
        fps = np.zeros(n, dtype=np.uint64)
        check(capi.load().vsrmc_tlc_fingerprint_batch(self._h, device, _p(words), _p(off), n, _p(fps)))
        return fps

    def tlc_min_permutation(self, record):
        """Number of the value permutation whose permuted state TLC fingerprints under SYMMETRY (host code)."""
        record = np.ascontiguousarray(record, dtype=np.uint64)
        perm = C.c_int32()
        check(capi.load().vsrmc_tlc_min_permutation(self._h, _p(record), C.byref(perm)))
        return perm.value

    def tlc_view_bytes(self, record, permutation=0):
        """The byte stream TLC's fingerprint of one wire record is taken over (host code, a diagnostic)."""
        record = np.ascontiguousarray(record, dtype=np.uint64)
        n = C.c_uint64()
        check(capi.load().vsrmc_tlc_view_bytes(self._h, _p(record), permutation, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.uint8)
        check(capi.load().vsrmc_tlc_view_bytes(self._h, _p(record), permutation, _p(out), len(out), C.byref(n)))
        return out.tobytes()

    def simulate(self, n_walkers=1 << 16, max_depth=100, seed=1, max_seconds=10.0, device=0):
        """≙ `tlc2.TLC -simulate -depth max_depth`: random walks on the GPU until an invariant fails or time runs out.
        -> dict(found, viol_mask, steps, walks, seconds, trace=[(action name, wire record)] or None)."""
        r = capi.SimResult()
        check(capi.load().vsrmc_simulate(self._h, device, n_walkers, max_depth, seed, max_seconds, C.byref(r)))
        out = dict(found=r.found, viol_mask=r.viol_mask, steps=r.steps, walks=r.walks, seconds=r.seconds, trace=None)
        if r.found == 1:
            ords = [int(r.ords[k]) for k in range(r.viol_steps)]
            out["ordinals"] = ords
            out["trace"] = self.replay(ords, device)
        return out

    def replay(self, ords, device=0):
        """Re-execute a path of ordinals from Init on the GPU -> [(action name, wire record)]."""
        n = len(ords)
        cap_w = (n + 2) * int(self.layout.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(n + 3, dtype=np.uint64)
        acts = np.zeros(n + 3, dtype=np.int32)
        o = np.array(list(ords) + [0], dtype=np.uint32)
        ns = C.c_uint64()
        check(capi.load().vsrmc_model_replay(self._h, device, _p(o), n, _p(words), cap_w, _p(off), _p(acts), len(off), C.byref(ns)))
        return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(ns.value)]

    def parse_states(self, text):
        """TLC's value syntax (trace expression, one state record, or console form) -> [(action name or None, wire record)];
        the inverse of format_state (≙ reading a TLC trace file)."""
        raw = text.encode() if isinstance(text, str) else text
        ns = C.c_uint64()
        check(capi.load().vsrmc_model_parse_states(self._h, raw, None, 0, None, None, 0, C.byref(ns)))
        n = ns.value
        cap_w = max(1, n) * int(self.layout.fixed_words + 255)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(n + 1, dtype=np.uint64)
        acts = np.zeros(max(1, n), dtype=np.int32)
        check(capi.load().vsrmc_model_parse_states(self._h, raw, _p(words), cap_w, _p(off), _p(acts), n + 1, C.byref(ns)))
        return [(ACTION_NAMES[acts[t]] if acts[t] >= 0 else None, words[int(off[t]): int(off[t + 1])].copy()) for t in range(n)]

    def check_trace(self, records, device=0):
        """Is this list of wire records a behaviour of the model (Init, then successor after successor, as generated on the
        GPU)?  -> dict(ok, first_bad, ords, actions, inv_mask_last)."""
        n = len(records)
        off = np.zeros(n + 1, dtype=np.uint64)
        for t, r in enumerate(records):
            off[t + 1] = off[t] + np.uint64(len(r))
        words = np.concatenate([np.asarray(r, dtype=np.uint64) for r in records]) if n else np.zeros(1, dtype=np.uint64)
        ords = np.zeros(max(1, n), dtype=np.uint32)
        acts = np.full(max(1, n), -1, dtype=np.int32)
        bad, inv = C.c_int64(), C.c_int32()
        check(capi.load().vsrmc_model_check_trace(self._h, device, _p(words), _p(off), n, _p(ords), _p(acts), C.byref(bad), C.byref(inv)))
        good = n if bad.value < 0 else bad.value
        return dict(ok=bad.value < 0, first_bad=bad.value, ords=[int(x) for x in ords[: max(0, good - 1)]],
                    actions=[ACTION_NAMES[a] for a in acts[:good]], inv_mask_last=inv.value)

    def close(self):
        if self._h:
            capi.load().vsrmc_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class FPSet:
    """≙ tlc2.tool.fp.FPSet: an open-addressing set of 64-bit fingerprints in HBM."""

    def __init__(self, log2_slots=20, device=0):
        self._h = C.c_void_p()
        check(capi.load().vsrmc_fpset_create(device, log2_slots, C.byref(self._h)))

    def put_block(self, fps):
        """-> uint8 array: 1 where the fingerprint was already present (FPSet.put's return value)."""
        fps = np.ascontiguousarray(fps, dtype=np.uint64)
        out = np.zeros(len(fps), dtype=np.uint8)
        check(capi.load().vsrmc_fpset_put_batch(self._h, _p(fps), len(fps), _p(out)))
        return out

    def contains_block(self, fps):
        fps = np.ascontiguousarray(fps, dtype=np.uint64)
        out = np.zeros(len(fps), dtype=np.uint8)
        check(capi.load().vsrmc_fpset_contains_batch(self._h, _p(fps), len(fps), _p(out)))
        return out

    def put(self, fp):
        return bool(self.put_block(np.array([fp], dtype=np.uint64))[0])

    def contains(self, fp):
        return bool(self.contains_block(np.array([fp], dtype=np.uint64))[0])

    def put_block_device(self, d_fps_ptr, n, d_out_ptr, stream=None):
        check(capi.load().vsrmc_fpset_put_batch_device(self._h, C.c_void_p(d_fps_ptr), n, C.c_void_p(d_out_ptr),
                                                       C.c_void_p(stream or 0)))

    def contains_block_device(self, d_fps_ptr, n, d_out_ptr, stream=None):
        check(capi.load().vsrmc_fpset_contains_batch_device(self._h, C.c_void_p(d_fps_ptr), n, C.c_void_p(d_out_ptr),
                                                            C.c_void_p(stream or 0)))

    def size(self):
        n = C.c_uint64()
        check(capi.load().vsrmc_fpset_size(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            capi.load().vsrmc_fpset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class StateQueue:
    """≙ tlc2.tool.queue.StateQueue: a FIFO of state records in HBM (sEnqueue(TLCState[]) / sDequeue(int) / size)."""

    def __init__(self, capacity_words=1 << 24, capacity_states=1 << 20, device=0):
        self._h = C.c_void_p()
        check(capi.load().vsrmc_queue_create(device, capacity_words, capacity_states, C.byref(self._h)))

    def s_enqueue(self, words, off):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        off = np.ascontiguousarray(off, dtype=np.uint64)
        check(capi.load().vsrmc_queue_enqueue_batch(self._h, _p(words), _p(off), len(off) - 1))

    def s_dequeue(self, max_states, cap_words=None):
        cap_words = cap_words or 256 * max_states
        words = np.zeros(cap_words, dtype=np.uint64)
        off = np.zeros(max_states + 1, dtype=np.uint64)
        n = C.c_uint64()
        check(capi.load().vsrmc_queue_dequeue_batch(self._h, max_states, _p(words), cap_words, _p(off), C.byref(n)))
        return words[: int(off[n.value])].copy(), off[: n.value + 1].copy()

    def size(self):
        n = C.c_uint64()
        check(capi.load().vsrmc_queue_size(self._h, C.byref(n)))
        return n.value

    def close(self):
        if self._h:
            capi.load().vsrmc_queue_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ModelChecker:
    """≙ tlc2.tool.ModelChecker: level-synchronous BFS; `step()` = every Worker draining one level of the StateQueue."""

    @classmethod
    def auto(cls, model, device=0, **kw):
        """every buffer sized from the free memory of the device (vsrmc_options with zeros): seen-set = the largest power of two of
        slots within 30 % of it, the rest in two equal record buffers; pass table_log2 / frontier_words / .. to pin one of them"""
        kw = dict(dict(table_log2=0, frontier_words=0, frontier_states=0, pending_entries=0), **kw)
        return cls(model, device=device, **kw)

    def __init__(self, model, device=0, table_log2=24, frontier_words=1 << 25, frontier_states=1 << 20,
                 pending_entries=1 << 21, keep_trace=True, trace_entries=0, exact_ties=False, recover=None, host_frontier=False, frontier_words_b=0):
        """recover: path of a checkpoint written by save() — continue that search (≙ `tlc2.TLC -recover`).
        table_log2 = 0 / frontier_words = 0 / frontier_states = 0 / pending_entries = 0: sized from the free device memory (auto())."""
        self.model = model
        o = capi.Options()
        capi.load().vsrmc_options_default(C.byref(o))
        o.device, o.table_log2 = device, table_log2
        o.frontier_words, o.frontier_states, o.pending_entries = frontier_words, frontier_states, pending_entries
        o.keep_trace = int(keep_trace)
        o.trace_entries = trace_entries
        o.exact_ties = int(exact_ties)
        o.frontier_words_b = frontier_words_b         # second record buffer (levels 2, 4, ...); 0 = frontier_words
        # bit mask: record buffer 0 (levels 1, 3, ...) / 1 (levels 2, 4, ...) in pinned host memory (≙ DiskStateQueue); True = both
        o.host_frontier = 3 if host_frontier is True else int(host_frontier)
        self.options = o
        self._h = C.c_void_p()
        if recover is None:
            check(capi.load().vsrmc_checker_create(model._h, C.byref(o), C.byref(self._h)))
            self._fresh()
        else:
            check(capi.load().vsrmc_checker_load(model._h, C.byref(o), os.fsencode(recover), C.byref(self._h)))
            info = capi.LevelInfo()
            check(capi.load().vsrmc_checker_status(self._h, C.byref(info)))
            self.level, self.n_frontier, self.distinct = info.level, info.n_new, info.distinct
            self.levels = [dict(level=info.level, n_new=info.n_new, generated=0, deadlocks=0, recovered=True)]
            self.violation = None
            self.rebased = []
            self.depth = self.level + info.reserved0               # a deep search: the levels beyond the stored one came along in the seen-set
        check(capi.load().vsrmc_checker_options(self._h, C.byref(o)))       # sizes left 0 were derived from the free device memory

    def save(self, path):
        """Checkpoint between two levels (≙ TLC's checkpoint of FPSet + StateQueue + TLCTrace): one file."""
        check(capi.load().vsrmc_checker_save(self._h, os.fsencode(path)))

    def _fresh(self):
        self.depth = 1
        self.level = 1
        self.n_frontier = 1
        self.distinct = 1
        self.levels = [dict(level=1, n_new=1, generated=0, deadlocks=0)]
        self.violation = None
        self.rebased = []

    def reset(self):
        """Back to Init with an empty seen-set; keeps the HBM allocations (a fresh TLC run on the same model)."""
        check(capi.load().vsrmc_checker_reset(self._h))
        self._fresh()

    def step(self):
        info = capi.LevelInfo()
        check(capi.load().vsrmc_checker_step(self._h, C.byref(info)))
        d = info.as_dict()
        self.level, self.n_frontier, self.distinct = d["level"], d["n_new"], d["distinct"]
        self.depth = self.level
        if d["n_new"]:
            self.levels.append(d)
        if d["viol_mask"] and self.violation is None:
            self.violation = dict(level=d["level"], index=d["viol_index"], fp=d["viol_fp"], mask=d["viol_mask"])
        return d

    def check(self, max_depth=0, max_seconds=0.0):
        """≙ ModelChecker.runTLC in one native call (vsrmc_check: the automatic level scheme)
        -> "exhausted" | "violation" | "max-depth" | "max-seconds" | "seen-set-full"."""
        reason = C.c_int32()
        info = capi.LevelInfo()
        check(capi.load().vsrmc_check(self._h, max_depth, max_seconds, C.byref(reason), C.byref(info)))
        d = info.as_dict()
        self.depth = d["level"]                                           # the deepest level that is complete (stored or in the seen-set only)
        self.distinct = d["distinct"]
        st = capi.LevelInfo()
        check(capi.load().vsrmc_checker_status(self._h, C.byref(st)))     # the newest STORED level: what level_fps / frontier / trace(index) address
        self.level, self.n_frontier = st.level, st.n_new
        check(capi.load().vsrmc_checker_options(self._h, C.byref(self.options)))   # (the seen-set may have grown)
        if reason.value == 1:
            self.violation = dict(level=d["level"], index=d["viol_index"], fp=d["viol_fp"], mask=d["viol_mask"],
                                  probed=d["viol_index"] == (1 << 64) - 1)
        return ["exhausted", "violation", "max-depth", "max-seconds", "seen-set-full"][reason.value]

    def _deep_dict(self, info):
        d = info.as_dict()
        d["ancestor_index"] = d.pop("viol_index")
        if d["level"] and d["viol_mask"] and self.violation is None:
            self.violation = dict(level=d["level"], index=None, fp=d["viol_fp"], mask=d["viol_mask"], probed=True)
        return d

    def deepen(self):
        """One more level beyond the record buffers (vsrmc_checker_deepen): the next level is inserted into the seen-set only — counted,
        checked, checksummed, its records regenerated from the newest stored level whenever they are needed — and the level after it is
        probed.  -> (inserted dict, probed dict or None).  `self.depth` = the deepest level that is complete in the seen-set."""
        a, b = capi.LevelInfo(), capi.LevelInfo()
        check(capi.load().vsrmc_checker_deepen(self._h, C.byref(a), C.byref(b)))
        da, db = self._deep_dict(a), self._deep_dict(b)
        self.depth, self.distinct = da["level"], da["distinct"]
        return da, (db if db["level"] else None)

    def advance(self):
        """One unit of progress of the automatic level scheme (vsrmc_checker_advance): an ordinary level while the next one is
        predicted to fit the record buffers, else a deepen() pass.  -> ("level", dict, None) | ("deep", inserted dict, probed dict or None)."""
        a, b = capi.LevelInfo(), capi.LevelInfo()
        what = C.c_int32()
        check(capi.load().vsrmc_checker_advance(self._h, C.byref(a), C.byref(b), C.byref(what)))
        while what.value == 3:                       # re-based: the newest seen-set-only level became the stored base (no new level): go on
            self.level, self.n_frontier = a.level, a.n_new
            self.rebased.append(dict(level=a.level, n=a.n_new, seconds=a.seconds, launches=a.pending))
            check(capi.load().vsrmc_checker_advance(self._h, C.byref(a), C.byref(b), C.byref(what)))
        if what.value == 1:
            d = a.as_dict()
            self.level, self.n_frontier, self.distinct = d["level"], d["n_new"], d["distinct"]
            self.depth = self.level
            if d["n_new"]:
                self.levels.append(d)
            if d["viol_mask"] and self.violation is None:
                self.violation = dict(level=d["level"], index=d["viol_index"], fp=d["viol_fp"], mask=d["viol_mask"])
            return "level", d, None
        da, db = self._deep_dict(a), self._deep_dict(b)
        self.depth, self.distinct = da["level"], da["distinct"]
        return "deep", da, (db if db["level"] else None)

    def room(self):
        """The seen-set before the next advance(): 0 = room enough, 1 = just re-hashed into a table of twice the slots (self.options.table_log2 is
        refreshed), 2 = more than 85 % full and no device memory to grow into: the search is incomplete at the depth reached."""
        st = C.c_int32()
        check(capi.load().vsrmc_checker_room(self._h, C.byref(st)))
        if st.value == 1:
            check(capi.load().vsrmc_checker_options(self._h, C.byref(self.options)))
        return st.value

    def run(self, max_depth=None, max_seconds=None, stop_on_violation=True):
        """Worker.run until the queue is empty, an invariant is violated, or a bound is hit — the automatic level scheme: levels are
        stored while they fit the record buffers, the search goes on beyond them through the seen-set alone (deepen)."""
        import time
        t0 = time.time()
        while True:
            if max_depth is not None and self.depth >= max_depth:
                return "max-depth"
            if max_seconds is not None and time.time() - t0 > max_seconds:
                return "max-seconds"
            if self.room() == 2:
                return "seen-set-full"
            kind, d, p = self.advance()
            if d["n_new"] == 0:
                return "exhausted"
            if self.violation is not None and stop_on_violation:
                return "violation"

    def violation_trace(self):
        """the counter-example of the violation run() / advance() reported, wherever it was found"""
        v = self.violation
        return self.probe_trace() if v.get("probed") else self.trace(v["level"], v["index"])

    def level_fps(self):
        out = np.zeros(max(1, self.n_frontier), dtype=np.uint64)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_level_fps(self._h, _p(out), len(out), C.byref(n)))
        return out[: n.value].copy()

    def tlc_level_fps(self, fetch=True):
        """FP64 (TLC's fingerprint, [TLC-RECALLED]) of every state of the newest level, from the frontier in HBM -> (sorted fps or None, kernel ms)."""
        out = np.zeros(max(1, self.n_frontier), dtype=np.uint64) if fetch else None
        n = C.c_uint64()
        ms = C.c_double()
        check(capi.load().vsrmc_checker_tlc_level_fps(self._h, _p(out) if fetch else None, len(out) if fetch else 0, C.byref(n), C.byref(ms)))
        return (out[: n.value].copy() if fetch else None), ms.value

    def frontier(self):
        """The newest level in wire layout -> (words, offsets)."""
        lay = self.model.layout
        cap_w = max(1, self.n_frontier) * int(lay.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(self.n_frontier + 1, dtype=np.uint64)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_frontier(self._h, _p(words), cap_w, _p(off), len(off), C.byref(n)))
        return words[: int(off[n.value])].copy(), off[: n.value + 1].copy()

    def level_checksum(self):
        """(xor, sum mod 2^64, count) of the newest level's fingerprints, computed on the device."""
        x, s_, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_checker_level_checksum(self._h, C.byref(x), C.byref(s_), C.byref(n)))
        return x.value, s_.value, n.value

    def select(self, action_mask, max_states=4096):
        """States of the newest level in which an action of `action_mask` (bit a = action id a) is enabled
        -> (words, off, n_matching): at most max_states wire records, and how many such states the level has."""
        cap_w = int(max_states) * int(self.model.layout.max_record_words) + 16
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(int(max_states) + 1, dtype=np.uint64)
        n, total = C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_checker_select(self._h, int(action_mask), int(max_states), _p(words), cap_w, _p(off), C.byref(n),
                                               C.byref(total)))
        return words[: int(off[n.value])], off[: n.value + 1], total.value

    def find_fp(self, fp):
        """Index (in the newest level's index range) of the state with fingerprint `fp`, or None."""
        idx = C.c_uint64()
        check(capi.load().vsrmc_checker_find_fp(self._h, int(fp), C.byref(idx)))
        return None if idx.value == (1 << 64) - 1 else idx.value

    def trace_fp(self, level, fp):
        """TLCTrace.getTrace for the level-`level` state with fingerprint `fp` (any completed level)."""
        lay = self.model.layout
        cap_w = (level + 1) * int(lay.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(level + 2, dtype=np.uint64)
        acts = np.zeros(level + 2, dtype=np.int32)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_trace_fp(self._h, level, int(fp), _p(words), cap_w, _p(off), _p(acts), len(off), C.byref(n)))
        return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(n.value)]

    def lookup(self, key, level=0, by_low_bits=False):
        """one step of a trace walk through the seen-set -> (fingerprint, meta) or None; by_low_bits: a third element = the number
        of states of that level whose fingerprint ends in those 45 bits (> 1: the predecessor pointer is ambiguous)"""
        found, fp, meta = C.c_int32(), C.c_uint64(), C.c_uint64()
        check(capi.load().vsrmc_checker_lookup(self._h, int(key), int(level), int(by_low_bits), C.byref(found), C.byref(fp), C.byref(meta)))
        if not found.value:
            return None
        return (fp.value, meta.value, found.value) if by_low_bits else (fp.value, meta.value)

    def trace(self, level, index):
        """TLCTrace.getTrace -> list of (action name, record words) from Init to state `index` of the newest level."""
        lay = self.model.layout
        cap_w = (level + 1) * int(lay.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(level + 2, dtype=np.uint64)
        acts = np.zeros(level + 2, dtype=np.int32)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_trace(self._h, level, index, _p(words), cap_w, _p(off), _p(acts), len(off),
                                              C.byref(n)))
        return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(n.value)]

    def probe(self):
        """Expand the newest level without storing it: invariants of every successor that is not an earlier-level state are
        checked, nothing is inserted or written (no frontier memory needed; the search cannot continue afterwards).  Also valid
        right after a step() that failed with "frontier full".  -> dict(level, generated, viol_fp, viol_mask, parent_index, ...)"""
        info = capi.LevelInfo()
        check(capi.load().vsrmc_checker_probe(self._h, C.byref(info)))
        d = info.as_dict()
        d["parent_index"] = d.pop("viol_index")
        if d["viol_mask"] and self.violation is None:
            self.violation = dict(level=d["level"], index=None, fp=d["viol_fp"], mask=d["viol_mask"], probed=True)
        return d

    def probe2(self):
        """Two levels beyond the newest one without storing either: level+1 as a virtual level (fingerprints claimed, exact count,
        invariants), level+2 as a probe over slices of regenerated level+1 states.  -> (virtual level dict, probe level dict)"""
        v, p = capi.LevelInfo(), capi.LevelInfo()
        check(capi.load().vsrmc_checker_probe2(self._h, C.byref(v), C.byref(p)))
        dv, dp = v.as_dict(), p.as_dict()
        for d in (dv, dp):
            d["ancestor_index"] = d.pop("viol_index")
            if d["viol_mask"] and self.violation is None:
                self.violation = dict(level=d["level"], index=None, fp=d["viol_fp"], mask=d["viol_mask"], probed=True)
        return dv, dp

    def probe3(self):
        """Three levels beyond the newest one without storing any: level+1 and level+2 as virtual levels, level+3 as a probe over
        regenerated sub-slices (level is expanded three times, level+1 twice).  -> (virtual dict, virtual dict, probe dict); a
        violation in a virtual level ends the call early (the later dicts then have level 0)."""
        v1, v2, p = capi.LevelInfo(), capi.LevelInfo(), capi.LevelInfo()
        check(capi.load().vsrmc_checker_probe3(self._h, C.byref(v1), C.byref(v2), C.byref(p)))
        out = [v1.as_dict(), v2.as_dict(), p.as_dict()]
        for d in out:
            d["ancestor_index"] = d.pop("viol_index")
            if d["viol_mask"] and self.violation is None:
                self.violation = dict(level=d["level"], index=None, fp=d["viol_fp"], mask=d["viol_mask"], probed=True)
        return tuple(out)

    def probe_violators(self):
        """fingerprints (ascending) of the distinct violating states of the level the last probe / deepen / advance call probed"""
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_probe_violators(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(1, n.value), dtype=np.uint64)
        check(capi.load().vsrmc_checker_probe_violators(self._h, _p(out), len(out), C.byref(n)))
        return [int(x) for x in out[: n.value]]

    def probe_trace(self):
        """The counter-example of the violation probe() reported: [(action name, record)] from Init to the violator."""
        lay = self.model.layout
        n_max = max(self.level, self.depth) + 3                 # (the deepest level in the seen-set + the probed state beyond it)
        cap_w = (n_max + 1) * int(lay.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(n_max + 2, dtype=np.uint64)
        acts = np.zeros(n_max + 2, dtype=np.int32)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_probe_trace(self._h, _p(words), cap_w, _p(off), _p(acts), len(off), C.byref(n)))
        return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(n.value)]

    def trace_to_violator(self, fp):
        """The counter-example that ends in the violating state `fp` of the last probed level (one of probe_violators()), as the search reached it:
        [(action name, record)] from Init.  probe_trace() is this for probe_violators()[0]."""
        lay = self.model.layout
        n_max = max(self.level, self.depth) + 3
        cap_w = (n_max + 1) * int(lay.max_record_words)
        words = np.zeros(cap_w, dtype=np.uint64)
        off = np.zeros(n_max + 2, dtype=np.uint64)
        acts = np.zeros(n_max + 2, dtype=np.int32)
        n = C.c_uint64()
        check(capi.load().vsrmc_checker_trace_to_violator(self._h, int(fp), _p(words), cap_w, _p(off), _p(acts), len(off), C.byref(n)))
        return [(ACTION_NAMES[acts[t]], words[int(off[t]): int(off[t + 1])].copy()) for t in range(n.value)]

    def close(self):
        if self._h:
            capi.load().vsrmc_checker_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

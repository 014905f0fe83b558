"""ctypes binding of libvsrmc.so (include/vsrmc.h).  No fallback: if the library is missing, loading raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VSRMC_LIB") or os.path.join(HERE, "libvsrmc.so")   # VSRMC_LIB: an experimental build of the same library (tools/ab_build.sh)

u64p = C.POINTER(C.c_uint64)


class Layout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "replica_count", "client_count", "value_count", "start_view_on_timer_limit", "symmetry", "invariant_mask",
        "assume_commit_number", "check_deadlock", "words_per_replica", "fixed_words", "permutations", "max_bag",
        "max_record_words", "module")] + [("reserved", C.c_int32 * 2)]


class Options(C.Structure):
    _fields_ = [("device", C.c_int32), ("table_log2", C.c_int32), ("frontier_words", C.c_uint64),
                ("frontier_states", C.c_uint64), ("pending_entries", C.c_uint64), ("keep_trace", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("trace_entries", C.c_uint64), ("exact_ties", C.c_int32), ("filter_log2", C.c_int32), ("host_frontier", C.c_int32), ("reserved0", C.c_int32), ("frontier_words_b", C.c_uint64)]


class LevelInfo(C.Structure):
    _fields_ = [("level", C.c_int32), ("error_code", C.c_int32), ("frontier", C.c_uint64), ("generated", C.c_uint64),
                ("n_new", C.c_uint64), ("distinct", C.c_uint64), ("total_generated", C.c_uint64),
                ("deadlocks", C.c_uint64), ("pending", C.c_uint64), ("probes", C.c_uint64), ("words_new", C.c_uint64), ("record_words", C.c_uint64),
                ("max_bag", C.c_uint64), ("viol_fp", C.c_uint64), ("viol_index", C.c_uint64), ("viol_mask", C.c_int32),
                ("reserved0", C.c_int32), ("seconds", C.c_double), ("expand_ms", C.c_double),
                ("materialize_ms", C.c_double), ("act_generated", C.c_uint64 * 16), ("phase_cycles", C.c_uint64 * 8),
                ("fp_xor", C.c_uint64), ("fp_sum", C.c_uint64), ("limit_rechecked", C.c_uint64)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n not in ("act_generated", "reserved0", "phase_cycles")}
        d["act_generated"] = list(self.act_generated)
        d["phase_cycles"] = list(self.phase_cycles)
        return d


class ShardIO(C.Structure):
    _fields_ = [("cand_send", C.c_void_p), ("cand_cap", C.c_uint64)]


A2AV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_void_p, C.POINTER(C.c_uint64),
                       C.POINTER(C.c_uint64), C.c_uint32, C.c_void_p)
AG_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32)


class Comm(C.Structure):
    """vsrmc_comm: the transport of the native sharded level loop (include/vsrmc.h)"""
    _fields_ = [("ctx", C.c_void_p), ("rank", C.c_int32), ("world", C.c_int32), ("host_buffers", C.c_int32), ("reserved0", C.c_int32),
                ("alltoallv", A2AV_FN), ("allgather", AG_FN)]


class SimResult(C.Structure):
    _fields_ = [("found", C.c_int32), ("viol_mask", C.c_int32), ("viol_steps", C.c_int32), ("reserved", C.c_int32),
                ("steps", C.c_uint64), ("walks", C.c_uint64), ("seconds", C.c_double), ("ords", C.c_uint32 * 512)]


# every symbol include/vsrmc.h declares: name -> (restype, argtypes)
V = C.c_void_p
SYMBOLS = {
    "vsrmc_last_error": (C.c_char_p, []),
    "vsrmc_version": (C.c_int32, []),
    "vsrmc_device_count": (C.c_int32, []),
    "vsrmc_model_load": (C.c_int32, [C.c_char_p, C.c_char_p, C.POINTER(V)]),
    "vsrmc_model_from_constants": (C.c_int32, [C.c_int32] * 8 + [C.POINTER(V)]),
    "vsrmc_model2_from_constants": (C.c_int32, [C.c_int32] * 6 + [C.POINTER(V)]),
    "vsrmc_model3_from_constants": (C.c_int32, [C.c_int32] * 6 + [C.POINTER(V)]),
    "vsrmc_model_info": (C.c_int32, [V, C.POINTER(Layout)]),
    "vsrmc_model_set_fp_seed": (C.c_int32, [V, C.c_uint64]),
    "vsrmc_model_fp_seed": (C.c_uint64, [V]),
    "vsrmc_model_init_state": (C.c_int32, [V, V, C.c_int32, C.POINTER(C.c_int32)]),
    "vsrmc_model_format_state": (C.c_int32, [V, V, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
    "vsrmc_action_name": (C.c_char_p, [C.c_int32]),
    "vsrmc_model_destroy": (None, [V]),
    "vsrmc_fpset_create": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(V)]),
    "vsrmc_fpset_put_batch": (C.c_int32, [V, V, C.c_uint64, V]),
    "vsrmc_fpset_contains_batch": (C.c_int32, [V, V, C.c_uint64, V]),
    "vsrmc_fpset_put_batch_device": (C.c_int32, [V, V, C.c_uint64, V, V]),
    "vsrmc_fpset_contains_batch_device": (C.c_int32, [V, V, C.c_uint64, V, V]),
    "vsrmc_fpset_size": (C.c_int32, [V, C.POINTER(C.c_uint64)]),
    "vsrmc_fpset_destroy": (None, [V]),
    "vsrmc_expand_batch": (C.c_int32, [V, C.c_int32, V, V, C.c_uint64, V, C.c_uint64, V, C.c_uint64,
                                       C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vsrmc_fingerprint_batch": (C.c_int32, [V, C.c_int32, V, V, C.c_uint64, V, V]),
    "vsrmc_tlc_fingerprint_batch": (C.c_int32, [V, C.c_int32, V, V, C.c_uint64, V]),
    "vsrmc_tlc_view_bytes": (C.c_int32, [V, V, C.c_int32, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_tlc_min_permutation": (C.c_int32, [V, V, C.POINTER(C.c_int32)]),
    "vsrmc_fp64_new": (C.c_uint64, []),
    "vsrmc_fp64_extend": (C.c_uint64, [C.c_uint64, V, C.c_uint64]),
    "vsrmc_checker_tlc_level_fps": (C.c_int32, [V, V, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]),
    "vsrmc_options_default": (None, [C.POINTER(Options)]),
    "vsrmc_checker_create": (C.c_int32, [V, C.POINTER(Options), C.POINTER(V)]),
    "vsrmc_checker_reset": (C.c_int32, [V]),
    "vsrmc_checker_step": (C.c_int32, [V, C.POINTER(LevelInfo)]),
    "vsrmc_checker_level_fps": (C.c_int32, [V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_frontier": (C.c_int32, [V, V, C.c_uint64, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_trace": (C.c_int32, [V, C.c_int32, C.c_uint64, V, C.c_uint64, V, V, C.c_uint64,
                                        C.POINTER(C.c_uint64)]),
    "vsrmc_checker_destroy": (None, [V]),
    "vsrmc_model_replay": (C.c_int32, [V, C.c_int32, V, C.c_int32, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_model_replay_fps": (C.c_int32, [V, C.c_int32, V, C.c_int32, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_trace_fp": (C.c_int32, [V, C.c_int32, C.c_uint64, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_lookup": (C.c_int32, [V, C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                         C.POINTER(C.c_uint64)]),
    "vsrmc_checker_level_checksum": (C.c_int32, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vsrmc_checker_select": (C.c_int32, [V, C.c_uint32, C.c_uint64, V, C.c_uint64, V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vsrmc_checker_find_fp": (C.c_int32, [V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_check": (C.c_int32, [V, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(LevelInfo)]),
    "vsrmc_queue_create": (C.c_int32, [C.c_int32, C.c_uint64, C.c_uint64, C.POINTER(V)]),
    "vsrmc_queue_enqueue_batch": (C.c_int32, [V, V, V, C.c_uint64]),
    "vsrmc_queue_dequeue_batch": (C.c_int32, [V, C.c_uint64, V, C.c_uint64, V, C.POINTER(C.c_uint64)]),
    "vsrmc_queue_size": (C.c_int32, [V, C.POINTER(C.c_uint64)]),
    "vsrmc_queue_destroy": (None, [V]),
    "vsrmc_simulate": (C.c_int32, [V, C.c_int32, C.c_uint32, C.c_int32, C.c_uint64, C.c_double, C.POINTER(SimResult)]),
    "vsrmc_checker_probe": (C.c_int32, [V, C.POINTER(LevelInfo)]),
    "vsrmc_checker_probe2": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo)]),
    "vsrmc_checker_probe3": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo), C.POINTER(LevelInfo)]),
    "vsrmc_checker_deepen": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo)]),
    "vsrmc_checker_options": (C.c_int32, [V, C.POINTER(Options)]),
    "vsrmc_checker_advance": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo), C.POINTER(C.c_int32)]),
    "vsrmc_checker_room": (C.c_int32, [V, C.POINTER(C.c_int32)]),
    "vsrmc_checker_probe_candidates": (C.c_int32, [V, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_seen_batch": (C.c_int32, [V, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p]),
    "vsrmc_checker_probe_violators": (C.c_int32, [V, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_probe_trace": (C.c_int32, [V, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_bench_staging": (C.c_int32, [V, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "vsrmc_checker_trace_to_violator": (C.c_int32, [V, C.c_uint64, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_checker_save": (C.c_int32, [V, C.c_char_p]),
    "vsrmc_checker_status": (C.c_int32, [V, C.POINTER(LevelInfo)]),
    "vsrmc_checker_load": (C.c_int32, [V, C.POINTER(Options), C.c_char_p, C.POINTER(V)]),
    "vsrmc_model_parse_states": (C.c_int32, [V, C.c_char_p, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64)]),
    "vsrmc_model_check_trace": (C.c_int32, [V, C.c_int32, V, V, C.c_uint64, V, V, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "vsrmc_shard_expand": (C.c_int32, [V, C.POINTER(ShardIO), V]),
    "vsrmc_shard_claim": (C.c_int32, [V, V, C.c_uint64, V]),
    "vsrmc_shard_materialize": (C.c_int32, [V, C.POINTER(ShardIO), V]),
    "vsrmc_shard_count": (C.c_int32, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vsrmc_shard_export": (C.c_int32, [V, C.c_uint64, C.c_uint64, V, C.c_uint64, V, V, C.c_uint64, C.POINTER(C.c_uint64),
                                      C.POINTER(C.c_uint64)]),
    "vsrmc_shard_append": (C.c_int32, [V, V, C.c_uint64, V, V, C.c_uint64]),
    "vsrmc_shard_commit": (C.c_int32, [V, C.POINTER(LevelInfo)]),
    "vsrmc_shard_local_step": (C.c_int32, [V, C.POINTER(LevelInfo)]),
    "vsrmc_shard_partition": (C.c_int32, [V, C.POINTER(C.c_uint64)]),
    "vsrmc_shard_set_max_bag": (C.c_int32, [V, C.c_uint64]),
    "vsrmc_comm_rccl_unique_id": (C.c_int32, [V]),
    "vsrmc_comm_rccl_create": (C.c_int32, [V, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.POINTER(Comm))]),
    "vsrmc_comm_rccl_destroy": (None, [C.POINTER(Comm)]),
    "vsrmc_shard_loop_create": (C.c_int32, [V, C.POINTER(Comm), C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.POINTER(V)]),
    "vsrmc_shard_loop_destroy": (None, [V]),
    "vsrmc_shard_loop_step": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo)]),
    "vsrmc_shard_loop_run": (C.c_int32, [V, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.POINTER(LevelInfo)]),
    "vsrmc_shard_loop_room": (C.c_int32, [V, C.POINTER(C.c_int32)]),
    "vsrmc_shard_loop_overlap_stats": (C.c_int32, [V, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "vsrmc_shard_loop_save": (C.c_int32, [V, C.c_char_p]),
    "vsrmc_shard_loop_restore": (C.c_int32, [V, V, C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p, C.POINTER(C.c_void_p)]),
    "vsrmc_shard_loop_status": (C.c_int32, [V, C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int32),
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64),
                                            C.POINTER(C.c_uint64)]),
    "vsrmc_shard_loop_trace_fps": (C.c_int32, [V, C.c_int32, C.c_uint64, V]),
    "vsrmc_shard_loop_deepen": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo)]),
    "vsrmc_shard_loop_advance": (C.c_int32, [V, C.POINTER(LevelInfo), C.POINTER(LevelInfo), C.POINTER(C.c_int32)]),
    "vsrmc_shard_loop_probe_trace_fps": (C.c_int32, [V, V, C.c_int32, C.POINTER(C.c_int32)]),
}

_lib = None


class VsrmcError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "vsrmc error %d: %s" % (code, msg))
        self.code = code
        self.message = msg


def load():
    """Load libvsrmc.so and declare every exported symbol; raises if the library or a symbol is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python vsr_tlaplus_amd/build.py` (hipcc, gfx950); "
                              "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise VsrmcError(rc, load().vsrmc_last_error().decode())

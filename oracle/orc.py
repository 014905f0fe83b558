"""oracle/orc.py — ctypes binding of the C++ CPU oracle (TEST INFRASTRUCTURE, NOT PRODUCT CODE).

Builds oracle/build/liborc.so on demand (g++, seconds).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "build", "liborc.so")
BIN = os.path.join(HERE, "build", "vsr_oracle")
BIN_MT = os.path.join(HERE, "build", "vsr_oracle_mt")


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in ("vsr_oracle.cpp", "vsr_oracle_bfs.cpp", "vsr_oracle_mt.cpp", "vsr_oracle.hpp", "Makefile",
                                            "vrst_oracle.cpp", "vrst_oracle.hpp", "vras_oracle.cpp", "vras_oracle.hpp")]
    outs = (LIB, BIN, BIN_MT, os.path.join(HERE, "build", "liborc2.so"), os.path.join(HERE, "build", "vrst_oracle_mt"),
            os.path.join(HERE, "build", "liborc3.so"), os.path.join(HERE, "build", "vras_oracle_mt"))
    stale = force or not all(os.path.exists(o) for o in outs) or any(
        os.path.getmtime(s) > min(os.path.getmtime(o) for o in outs) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", HERE, "-s"], check=True)
    return LIB


LIB_TLC = os.path.join(HERE, "build", "liborc_tlc.so")
_tlc = None


def tlc_lib():
    """oracle/tlc_fp64.cpp: TLC-style fingerprints from the unpacked state (value tree + bit-serial FP64)."""
    global _tlc
    if _tlc is None:
        src = [os.path.join(HERE, f) for f in ("tlc_fp64.cpp", "vsr_oracle.cpp", "vsr_oracle.hpp")]
        if not os.path.exists(LIB_TLC) or any(os.path.getmtime(f) > os.path.getmtime(LIB_TLC) for f in src):
            subprocess.run(["make", "-C", HERE, "-s", "build/liborc_tlc.so"], check=True)
        L = C.CDLL(LIB_TLC)
        L.orc_tlc_last_error.restype = C.c_char_p
        L.orc_tlc_view_bytes.restype = C.c_longlong
        L.orc_tlc_view_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        L.orc_tlc_fingerprint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_tlc_permute_record.restype = C.c_longlong
        L.orc_tlc_permute_record.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        L.orc_fp64_bytes.restype = C.c_uint64
        L.orc_fp64_bytes.argtypes = [C.c_void_p, C.c_longlong]
        _tlc = L
    return _tlc


def tlc_view_bytes(params, record, perm=0):
    rec = np.ascontiguousarray(record, dtype=np.uint64)
    L = tlc_lib()
    n = L.orc_tlc_view_bytes(params.arr.ctypes.data, rec.ctypes.data, perm, None, 0)
    if n < 0:
        raise OracleError(-1, L.orc_tlc_last_error().decode())
    out = np.zeros(n, dtype=np.uint8)
    L.orc_tlc_view_bytes(params.arr.ctypes.data, rec.ctypes.data, perm, out.ctypes.data, n)
    return out.tobytes()


def tlc_fingerprint(params, record, with_perm=False):
    """FP64 of the view of the permuted state TLC fingerprints (the smallest by compareTo, variable by variable); with_perm: -> (fp, permutation number)"""
    rec = np.ascontiguousarray(record, dtype=np.uint64)
    fp = C.c_uint64()
    perm = C.c_int()
    L = tlc_lib()
    if L.orc_tlc_fingerprint(params.arr.ctypes.data, rec.ctypes.data, C.byref(fp), C.byref(perm)):
        raise OracleError(-1, L.orc_tlc_last_error().decode())
    return (fp.value, perm.value) if with_perm else fp.value


def tlc_permute_record(params, record, perm):
    """the wire record of the state under value permutation number `perm`: another member of its symmetry class"""
    rec = np.ascontiguousarray(record, dtype=np.uint64)
    out = np.zeros(len(rec) + 8, dtype=np.uint64)
    L = tlc_lib()
    n = L.orc_tlc_permute_record(params.arr.ctypes.data, rec.ctypes.data, perm, out.ctypes.data, len(out))
    if n < 0:
        raise OracleError(-1, L.orc_tlc_last_error().decode())
    return out[:n].copy()


def fp64_bytes(data):
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    return tlc_lib().orc_fp64_bytes(buf.ctypes.data, len(data))


ACTIONS = ["Initial predicate", "TimerSendSVC", "ReceiveHigherSVC", "ReceiveMatchingSVC", "SendDVC",
           "ReceiveHigherDVC", "ReceiveMatchingDVC", "SendSV", "ReceiveSV", "ReceiveClientRequest",
           "ReceivePrepareMsg", "ReceivePrepareOkMsg", "ExecuteOp", "SendGetState", "ReceiveGetState",
           "ReceiveNewState"]   # action ids in Next order (VSR.tla:896-913)
ACTION_NAMES = ACTIONS

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB)
        L.orc_last_error.restype = C.c_char_p
        L.orc_bfs_create.restype = C.c_void_p
        L.orc_bfs_create.argtypes = [C.c_void_p]
        L.orc_bfs_destroy.argtypes = [C.c_void_p]
        L.orc_bfs_step.restype = C.c_longlong
        L.orc_bfs_step.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_bfs_level_seconds.restype = C.c_double
        L.orc_bfs_level_seconds.argtypes = [C.c_void_p]
        L.orc_bfs_level_fps.restype = C.c_longlong
        L.orc_bfs_level_fps.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_longlong]
        L.orc_bfs_frontier.restype = C.c_longlong
        L.orc_bfs_frontier.argtypes = [C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
        L.orc_bfs_trace_fps.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_int]
        L.orc_set_fp_seed.argtypes = [C.c_uint64]
        L.orc_fp_seed.restype = C.c_uint64
        assert L.orc_fp_version() == FP_VERSION, "oracle/orc.py and oracle/vsr_oracle.hpp disagree on FP_VERSION"
        _lib = L
    return _lib


def set_fp_seed(seed):
    """second-hash audit: a seed xor-ed into every salt of the oracle's fingerprint (process-global; 0 = the fixtures' function)"""
    lib().orc_set_fp_seed(C.c_uint64(int(seed) & (2 ** 64 - 1)))


def fp_seed():
    return int(lib().orc_fp_seed())


FP_VERSION = 2     # == vsr_oracle.hpp FP_VERSION (checked when the library loads): which fingerprint function fixtures were made with


class OracleError(Exception):
    def __init__(self, code, msg):
        Exception.__init__(self, "oracle error %d: %s" % (code, msg))
        self.code = code


class Params:
    """Model constants in the order the C API expects."""

    def __init__(self, R=3, C=1, n=2, L=2, restart_limit=0, assume_commit_number=False, symmetry=True,
                 invariant_mask=1):
        self.R, self.C, self.n, self.L = R, C, n, L
        self.arr = np.array([R, C, n, L, restart_limit, int(assume_commit_number), int(symmetry), invariant_mask],
                            dtype=np.int32)

    @property
    def ptr(self):
        return self.arr.ctypes.data

    def wpr(self):
        return 1 + (self.R + 2) // 2

    def fixed_words(self):
        return 1 + self.R * self.wpr()


def _err(code):
    raise OracleError(code, lib().orc_last_error().decode())


def init_record(P):
    out = np.zeros(512, dtype=np.uint64)
    n = lib().orc_init_record(C.c_void_p(P.ptr), C.c_void_p(out.ctypes.data), 512)
    if n < 0:
        _err(n)
    return out[:n].copy()


def fingerprint(P, rec):
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    fp = C.c_uint64()
    ak = C.c_uint32()
    r = lib().orc_fingerprint(C.c_void_p(P.ptr), C.c_void_p(rec.ctypes.data), C.byref(fp), C.byref(ak))
    if r < 0:
        _err(r)
    return fp.value, ak.value


def invariants(P, rec):
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    r = lib().orc_invariants(C.c_void_p(P.ptr), C.c_void_p(rec.ctypes.data))
    if r < 0:
        _err(r)
    return r


def normalise(P, rec):
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    out = np.zeros(512, dtype=np.uint64)
    n = lib().orc_normalise(C.c_void_p(P.ptr), C.c_void_p(rec.ctypes.data), C.c_void_p(out.ctypes.data), 512)
    if n < 0:
        _err(n)
    return out[:n].copy()


def successors(P, rec):
    """-> list of dict(action, fp, auxkey, inv, words=np.uint64 array) in Next order."""
    rec = np.ascontiguousarray(rec, dtype=np.uint64)
    cap_w, cap_s = 1 << 16, 512
    words = np.zeros(cap_w, dtype=np.uint64)
    meta = np.zeros(5 * cap_s, dtype=np.uint64)
    used = C.c_int()
    n = lib().orc_successors(C.c_void_p(P.ptr), C.c_void_p(rec.ctypes.data), C.c_void_p(words.ctypes.data), cap_w,
                             C.c_void_p(meta.ctypes.data), cap_s, C.byref(used))
    if n < 0:
        _err(n)
    out, off = [], 0
    for k in range(n):
        a, fp, ak, nw, inv = (int(x) for x in meta[5 * k: 5 * k + 5])
        out.append(dict(action=a, fp=fp, auxkey=ak, inv=inv, words=words[off: off + nw].copy()))
        off += nw
    return out


class Bfs:
    """Level-synchronous BFS handle (Init = level 1)."""
    INFO = ["depth", "n_new", "generated", "ties", "deadlocks", "distinct", "total_generated", "viol_mask",
            "viol_gid", "error_code", "max_bag", "frontier_words"]

    def __init__(self, P):
        self.P = P
        self.h = lib().orc_bfs_create(C.c_void_p(P.ptr))
        if not self.h:
            _err(-2)
        self.info = dict(depth=1, n_new=1, generated=0, ties=0, deadlocks=0, distinct=1, total_generated=0,
                         viol_mask=0, viol_gid=2 ** 64 - 1, error_code=0, max_bag=0, frontier_words=0)

    def step(self):
        info = np.zeros(12, dtype=np.uint64)
        nn = lib().orc_bfs_step(self.h, C.c_void_p(info.ctypes.data))
        self.info = {k: int(v) for k, v in zip(self.INFO, info)}
        if self.info["error_code"]:
            code = self.info["error_code"] - (1 << 64)
            raise OracleError(code, lib().orc_last_error().decode())
        return nn

    def level_seconds(self):
        return lib().orc_bfs_level_seconds(self.h)

    def level_fps(self, level, cap=None):
        cap = cap or max(1, self.info["distinct"])
        out = np.zeros(cap, dtype=np.uint64)
        n = lib().orc_bfs_level_fps(self.h, level, C.c_void_p(out.ctypes.data), cap)
        if n < 0:
            raise OracleError(int(n), "level_fps")
        return out[:n].copy()

    def frontier(self):
        """-> (words, offsets) of the newest level."""
        nw = max(1, self.info["frontier_words"]) if self.info["depth"] > 1 else 512
        ns = max(1, self.info["n_new"]) + 1
        words = np.zeros(nw, dtype=np.uint64)
        off = np.zeros(ns, dtype=np.uint64)
        n = lib().orc_bfs_frontier(self.h, C.c_void_p(words.ctypes.data), nw, C.c_void_p(off.ctypes.data), ns)
        if n < 0:
            raise OracleError(int(n), "frontier buffers too small")
        return words[: int(off[n])].copy(), off[: n + 1].copy()

    def trace_fps(self, gid):
        out = np.zeros(4096, dtype=np.uint64)
        n = lib().orc_bfs_trace_fps(self.h, C.c_uint64(gid), C.c_void_p(out.ctypes.data), 4096)
        return out[:n].copy()

    def close(self):
        if self.h:
            lib().orc_bfs_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

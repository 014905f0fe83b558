"""ctypes binding of oracle/build/liborc3.so — the CPU ORACLE of the THIRD model (VR_APP_STATE.tla; oracle/vras_oracle.cpp
behind the same driver and C API as oracle/orc.py).  TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
import ctypes as C
import os

import numpy as np

from . import orc as _orc

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "build", "liborc3.so")
ACTIONS = [("PrimaryExecuteOp" if a == "ExecuteOp" else a) for a in _orc.ACTIONS]   # Next order, VR_APP_STATE.tla:811-831
OracleError = _orc.OracleError
_lib = None


def lib():
    global _lib
    if _lib is None:
        _orc.build()
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
        _lib = L
    return _lib


class Params:
    """VR_APP_STATE.cfg:4-7 constants; invariant_mask bits: 1 AcknowledgedWriteNotLost, 2 AcknowledgedWritesExistOnMajority,
    4 NoLogDivergence, 8 CommitNumberNeverHigherThanOpNumber, 16 NoAppStateDivergence (the shipped cfg checks 2 + 4 + 8 + 16 = 30)"""

    def __init__(self, R=3, n=2, L=2, no_progress_limit=0, symmetry=False, invariant_mask=30):
        self.R, self.C, self.n, self.L = R, 0, n, L
        self.arr = np.array([R, 0, n, L, no_progress_limit, 0, int(symmetry), invariant_mask], dtype=np.int32)

    @property
    def ptr(self):
        return self.arr.ctypes.data

    def fixed_words(self):
        return 1 + 2 * self.R


def _bind(name):
    """the functions of oracle/orc.py, run against liborc3.so"""
    f = getattr(_orc, name)

    def g(*a, **k):
        saved = _orc._lib
        _orc._lib = lib()
        try:
            return f(*a, **k)
        finally:
            _orc._lib = saved
    g.__name__ = name
    return g


init_record = _bind("init_record")
fingerprint = _bind("fingerprint")
invariants = _bind("invariants")
normalise = _bind("normalise")
successors = _bind("successors")


class Bfs(_orc.Bfs):
    def __init__(self, P):
        saved = _orc._lib
        _orc._lib = lib()
        try:
            super().__init__(P)
        finally:
            _orc._lib = saved
        self._L = lib()

    def _with(self, f, *a):
        saved = _orc._lib
        _orc._lib = self._L
        try:
            return f(*a)
        finally:
            _orc._lib = saved

    def step(self):
        return self._with(super().step)

    def level_fps(self, level, cap=None):
        return self._with(super().level_fps, level, cap)

    def frontier(self):
        return self._with(super().frontier)

    def close(self):
        return self._with(super().close)

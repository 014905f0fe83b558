"""ctypes bindings of the CPU ORACLES of the two analysis models — oracle/build/liborc2.so (VR_STATE_TRANSFER.tla; vrst_oracle.cpp)
and liborc3.so (VR_APP_STATE.tla; vras_oracle.cpp) — behind the same driver and C API as oracle/orc.py: one parameterised binding,
instantiated by oracle/orc2.py and oracle/orc3.py.  TEST INFRASTRUCTURE, NOT PRODUCT CODE."""
import ctypes as C
import os

import numpy as np

from . import orc as _orc

HERE = os.path.dirname(os.path.abspath(__file__))


def install(ns, libname, actions, default_mask, words_per_replica, params_doc):
    """Fill the module namespace `ns` with LIB, ACTIONS, OracleError, lib(), Params, Bfs and the per-state functions of oracle/orc.py
    run against `libname`."""
    state = dict(lib=None)
    LIB = os.path.join(HERE, "build", libname)

    def lib():
        if state["lib"] is None:
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
            L.orc_set_fp_seed.argtypes = [C.c_uint64]
            L.orc_fp_seed.restype = C.c_uint64
            state["lib"] = L
        return state["lib"]

    class Params:
        __doc__ = params_doc

        def __init__(self, R=3, n=2, L=2, no_progress_limit=0, symmetry=False, invariant_mask=default_mask):
            self.R, self.C, self.n, self.L = R, 0, n, L
            self.arr = np.array([R, 0, n, L, no_progress_limit, 0, int(symmetry), invariant_mask], dtype=np.int32)

        @property
        def ptr(self):
            return self.arr.ctypes.data

        def fixed_words(self):
            return 1 + words_per_replica * self.R

    def with_lib(f, *a, **k):
        saved = _orc._lib
        _orc._lib = lib()
        try:
            return f(*a, **k)
        finally:
            _orc._lib = saved

    def bind(name):
        f = getattr(_orc, name)

        def g(*a, **k):
            return with_lib(f, *a, **k)
        g.__name__ = name
        g.__doc__ = "oracle/orc.py's %s, run against %s" % (name, libname)
        return g

    class Bfs(_orc.Bfs):
        def __init__(self, P):
            with_lib(super().__init__, P)

        def step(self):
            return with_lib(super().step)

        def level_fps(self, level, cap=None):
            return with_lib(super().level_fps, level, cap)

        def frontier(self):
            return with_lib(super().frontier)

        def close(self):
            return with_lib(super().close)

    ns.update(LIB=LIB, ACTIONS=actions, OracleError=_orc.OracleError, lib=lib, Params=Params, Bfs=Bfs,
              **{name: bind(name) for name in ("init_record", "fingerprint", "invariants", "normalise", "successors", "set_fp_seed", "fp_seed")})

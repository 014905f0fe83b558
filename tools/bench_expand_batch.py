#!/usr/bin/env python3
"""Throughput of the host-buffer drop-ins on a real frontier: vsrmc_expand_batch (Tool.getNextStates over a batch; records
cross PCIe both ways and are re-laid out on the host) and vsrmc_fingerprint_batch, on level 16 of config 2 (838 162 states)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
mc = vt.ModelChecker(m, table_log2=24, frontier_words=1 << 27, frontier_states=1 << 22)
while mc.level < 16:
    mc.step()
words, off = mc.frontier()
n = len(off) - 1
import ctypes as C  # noqa: E402

from vsr_tlaplus_amd import capi  # noqa: E402

words = np.ascontiguousarray(words, dtype=np.uint64)
off = np.ascontiguousarray(off, dtype=np.uint64)
cap_succ, cap_words = 8 * n, 8 * n * 48
ow = np.zeros(cap_words, dtype=np.uint64)
om = np.zeros(8 * cap_succ, dtype=np.uint64)
n_out, w_out = C.c_uint64(), C.c_uint64()
p = lambda a: C.c_void_p(a.ctypes.data)  # noqa: E731
t0 = time.perf_counter()
capi.check(capi.load().vsrmc_expand_batch(m._h, 0, p(words), p(off), n, p(ow), cap_words, p(om), cap_succ, C.byref(n_out), C.byref(w_out)))
t1 = time.perf_counter()
fps = np.zeros(n, dtype=np.uint64)
aks = np.zeros(n, dtype=np.uint32)
capi.check(capi.load().vsrmc_fingerprint_batch(m._h, 0, p(words), p(off), n, p(fps), p(aks)))
t2 = time.perf_counter()
print(json.dumps(dict(states=n, successors=n_out.value, successor_words=w_out.value, expand_batch_s=round(t1 - t0, 3),
                      parents_per_s=round(n / (t1 - t0), 1), successors_per_s=round(n_out.value / (t1 - t0), 1),
                      fingerprint_batch_s=round(t2 - t1, 3), fingerprints_per_s=round(n / (t2 - t1), 1))))

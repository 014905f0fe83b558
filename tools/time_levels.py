#!/usr/bin/env python3
"""Timing only (no assertions): BASELINE configs[1] level by level, k_expand's HIP-event time per level and in total, `reps` runs.
    VSRMC_LIB=<an experimental build> python tools/time_levels.py [reps=3]   -> one JSON line per run"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
mc = vt.ModelChecker.auto(m, device=0, table_log2=31)
for r in range(reps + 1):
    mc.reset()
    ms = []
    while mc.level < 28:
        d = mc.step()
        ms.append(round(d["expand_ms"], 3))
    if r:
        print(json.dumps(dict(lib=os.environ.get("VSRMC_LIB", "product"), k_expand_ms=round(sum(ms), 3), small_levels_ms=ms[:12], last_levels_ms=ms[-4:])), flush=True)
mc.close()

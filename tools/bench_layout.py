#!/usr/bin/env python3
"""SoA or AoS?  k_expand's staging over one real level of BASELINE configs[1], once over the records as the checker stores them (variable-length records
behind a ref array) and once over the same level as fixed-stride columns (the layout BASELINE.json's north_star names): vsrmc_checker_bench_staging
(csrc/vsr_bench_layout.hpp).  Same tile loop, same LDS tile, same consumer; what differs is how a tile's bytes come out of the HBM.
    python tools/bench_layout.py [level=24] [reps=20]   -> one JSON line"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402
from vsr_tlaplus_amd import capi  # noqa: E402

level = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
mc = vt.ModelChecker.auto(m, device=0, table_log2=31)
d = None
while mc.level < level:
    d = mc.step()
rows = {}
for name, lay in (("records (AoS, the product's layout)", 0), ("fixed-stride columns (SoA)", 1)):
    ms, nb = C.c_double(), C.c_uint64()
    capi.check(capi.load().vsrmc_checker_bench_staging(mc._h, lay, reps, C.byref(ms), C.byref(nb)))
    rows[name] = dict(ms_per_pass=round(ms.value, 4), bytes_read=int(nb.value), GBs=round(nb.value / ms.value / 1e6, 1))
pipe = {}
if os.environ.get("BENCH_LAYOUT_PIPE", "1") != "0":                # round 6: the tile loop of the records layout, software-pipelined / at eight blocks per CU
    for name, lay in (("refs one tile ahead", 2), ("refs two, words one tile ahead", 3), ("as it is, 8 blocks per CU", 4), ("refs ahead, 8 blocks per CU", 5),
                      ("refs and words ahead, 8 blocks per CU", 6), ("as it is, 4 tiles per draw from the cursor", 7), ("refs and words ahead, 4 tiles per draw", 8),
                      ("refs and words ahead, 4 tiles per draw, 8 blocks per CU", 9), ("as it is, 16 tiles per draw", 10)):
        ms, nb = C.c_double(), C.c_uint64()
        capi.check(capi.load().vsrmc_checker_bench_staging(mc._h, lay, reps, C.byref(ms), C.byref(nb)))
        pipe[name] = dict(ms_per_pass=round(ms.value, 4), GBs=round(nb.value / ms.value / 1e6, 1))
a, b = rows["records (AoS, the product's layout)"], rows["fixed-stride columns (SoA)"]
print(json.dumps(dict(workload="BASELINE configs[1], level %d: %d states, %.1f B per record on average, LDS stride %d words" %
                               (mc.level, d["n_new"], 8.0 * d["record_words"] / d["n_new"], (int(m.layout.fixed_words) + int(m.layout.permutations) + d["max_bag"]) | 1),
                      k_expand_ms_of_this_level_for_scale=round(d["expand_ms"], 3), staging_pass=rows, records_pipelined=pipe, soa_over_aos_time=round(b["ms_per_pass"] / a["ms_per_pass"], 3),
                      note="the columns' padding (stride x states words) is allocated but never fetched; a stored SoA frontier would hold 1.16 x the bytes")))
mc.close()

#!/usr/bin/env python3
"""Shader-clock phase split of k_expand (wave 0 of every block; thread 0 inside the apply loop) over the large stored levels of a workload:
   python tools/phase_split.py [config2|readme]   -> one JSON line: shares of stage / enumerate / sort / apply / tail and of gen / hash / probe+claim / write"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
R, C, n, L = (3, 1, 2, 2) if wl == "config2" else (3, 1, 3, 3)
max_level = 28 if wl == "config2" else 21
m = vt.Model.from_constants(R=R, C_=C, n=n, L=L)
mc = vt.ModelChecker.auto(m, device=0, table_log2=31 if wl == "config2" else 0)
pc = [0] * 9
ms = 0.0
tiles = 0
while mc.level < max_level:
    d = mc.step()
    if not d["n_new"]:
        break
    if d["frontier"] >= 1 << 20:
        for i in range(8):
            pc[i] += int(d["phase_cycles"][i])
        pc[8] += int(d["act_generated"][0])
        ms += d["expand_ms"]
        tiles += (d["frontier"] + 63) // 64
    if mc.violation:
        break
tot = float(sum(pc[:5])) or 1.0
inner = float(sum(pc[5:9])) or 1.0
print(json.dumps(dict(workload=wl, lib=os.environ.get("VSRMC_LIB", "default"), k_expand_ms=round(ms, 2), tiles=tiles,
                      wave0_cycles_per_tile=round(tot / max(1, tiles)),
                      phases=dict(zip(("stage", "enumerate", "sort", "apply", "tail"), [round(x / tot, 3) for x in pc[:5]])),
                      apply_split=dict(zip(("gen", "hash", "probe_claim", "write"), [round(x / inner, 3) for x in pc[5:9]])),
                      thread0_apply_cycles_per_tile=round(inner / max(1, tiles)))))
mc.close()

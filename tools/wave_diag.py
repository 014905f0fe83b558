#!/usr/bin/env python3
"""Barrier-wait accounting of k_expand's waves (needs a library built with -DVSR_WAVE_DIAG=1 or =2: tools/ab_build.sh diag1 "-DVSR_WAVE_DIAG=1").
   VSRMC_LIB=vsr_tlaplus_amd/ab/libvsrmc_diag1.so python tools/wave_diag.py [config2|readme]
Prints, per wave index 0..3 of a block, the share of its residency spent inside block barriers — summed over the stored levels of the workload.
=1: every barrier of the tile loop; =2: only the barrier that closes the apply loop (the wait for the block's slowest wave of the apply phase)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "config2"
by_group = len(sys.argv) > 2 and sys.argv[2] == "groups"        # a -DVSR_WAVE_DIAG=3 build: wave 1's waits per barrier group, its residency in act_generated[0]
GROUPS = ["top/bottom of the tile loop", "after the ref load", "after staging", "after work-list fill + parent fps", "inside enumeration",
          "after enumeration", "sort / reservations", "apply-closing"]
R, C, n, L = (3, 1, 2, 2) if wl == "config2" else (3, 1, 3, 3)
max_level = 28 if wl == "config2" else 21
m = vt.Model.from_constants(R=R, C_=C, n=n, L=L)
mc = vt.ModelChecker.auto(m, device=0, table_log2=31 if wl == "config2" else 0)
wait = [0] * (8 if by_group else 4)
total = [0] * 4
ms = 0.0
rows = []
while mc.level < max_level:
    d = mc.step()
    if not d["n_new"]:
        break
    pc = [int(x) for x in d["phase_cycles"]]
    if by_group and d["frontier"] >= 1 << 20:
        for g in range(8):
            wait[g] += pc[g]
        total[0] += int(d["act_generated"][0])
        ms += d["expand_ms"]
    elif d["frontier"] >= 1 << 20:                                # the levels that matter (full grids)
        for w in range(4):
            wait[w] += pc[w]
            total[w] += pc[4 + w]
        ms += d["expand_ms"]
        rows.append(dict(level=d["level"], frontier=d["frontier"], ms=round(d["expand_ms"], 3),
                         wait_share=[round(pc[w] / max(1, pc[4 + w]), 4) for w in range(4)]))
    if mc.violation:
        break
for r in rows[-6:]:
    print(json.dumps(r))
if by_group:
    print(json.dumps(dict(workload=wl, lib=os.environ.get("VSRMC_LIB", "default"), k_expand_ms=round(ms, 2),
                          wave1_wait_share_by_barrier_group={GROUPS[g]: round(wait[g] / max(1, total[0]), 4) for g in range(8)},
                          wave1_wait_share_all=round(sum(wait) / max(1, total[0]), 4))))
    mc.close()
    sys.exit(0)
print(json.dumps(dict(workload=wl, lib=os.environ.get("VSRMC_LIB", "default"), k_expand_ms=round(ms, 2),
                      barrier_share_per_wave=[round(wait[w] / max(1, total[w]), 4) for w in range(4)],
                      barrier_share_all=round(sum(wait) / max(1, sum(total)), 4))))
mc.close()

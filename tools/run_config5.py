#!/usr/bin/env python3
"""BASELINE.json configs[4] — ReplicaCount=5, ClientCount=1, Values={v1,v2}, StartViewOnTimerLimit=2, the "288 GB/GPU FPSet sizing
stress" — on ONE MI355X as deep as its HBM allows: levels 1-12 materialised (every figure asserted against the CPU oracle's fixture,
tests/golden/oracle_levels_config5.json), level 13 as a virtual level (oracle-pinned: 596 058 668 new states), level 14 streamed
(inserted, never stored; oracle-pinned since the memory-lean driver reached it: 2 403 817 813 new states), level 15 probed
(vsrmc_checker_probe3).  Prints one JSON object: per-level figures, seen-set load, record bytes per state, states/s.  Level 15 has no CPU
counterpart (3.7e10 successors): GPU-sourced, labelled so.

    python tools/run_config5.py [--table-log2 33] [--probe-from 12]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vsr_tlaplus_amd as vt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--table-log2", type=int, default=33)
ap.add_argument("--probe-from", type=int, default=12, help="newest materialised level; 13 / 14 / 15 become virtual / streamed / probed")
ap.add_argument("--words-a", type=float, default=3.0e9, help="record buffer of the odd levels (words)")
ap.add_argument("--words-b", type=float, default=11.6e9, help="record buffer of the even levels (words): level 12 = 8.8e9 + the blocks' unfinished chunks")
ap.add_argument("--states", type=float, default=1.6e8)
a = ap.parse_args()
with open(os.path.join(ROOT, "tests", "golden", "oracle_levels_config5.json")) as f:
    g = json.load(f)
m = vt.Model.from_constants(R=5, C_=1, n=2, L=2)
t0 = time.time()
mc = vt.ModelChecker(m, device=0, table_log2=a.table_log2, frontier_words=int(a.words_a), frontier_words_b=int(a.words_b),
                     frontier_states=int(a.states), pending_entries=1 << 16)
setup = time.time() - t0
levels = []
t0 = time.time()
kms = 0.0
while mc.level < a.probe_from:
    d = mc.step()
    w = g["levels"][d["level"] - 1]
    assert (d["n_new"], d["generated"], d["deadlocks"], d["max_bag"]) == (w["new"], w["generated"], w["deadlocks"], w["max_bag"]), d["level"]
    assert d["viol_mask"] == 0
    kms += d["expand_ms"]
    levels.append(dict(level=d["level"], new=d["n_new"], generated=d["generated"], expand_ms=round(d["expand_ms"], 2), record_words=d["record_words"], source="oracle-pinned"))
t_mat = time.time() - t0
n_mat, words_last, n_last = mc.distinct, levels[-1]["record_words"], levels[-1]["new"]
v1, v2, p = mc.probe3()
dt = time.time() - t0
for v, kind in ((v1, "virtual"), (v2, "streamed"), (p, "probed")):
    w = g["levels"][v["level"] - 1] if 0 < v["level"] <= len(g["levels"]) else None
    if w is not None and kind != "probed":
        assert (v["n_new"], v["generated"], v["deadlocks"], v["max_bag"]) == (w["new"], w["generated"], w["deadlocks"], w["max_bag"]), v["level"]
        assert [int(x) for x in v["act_generated"][1:16]] == w["act_generated"][1:16], v["level"]
        if g.get("fp_version") == 2 and v["fp_xor"]:
            assert ("%016x" % v["fp_xor"], "%016x" % v["fp_sum"]) == (w["fp_xor"], w["fp_sum"]), v["level"]
    levels.append(dict(level=v["level"], kind=kind, new=v["n_new"], generated=v["generated"], deadlocks=v["deadlocks"], max_bag=v["max_bag"],
                       viol_mask=v["viol_mask"], seconds=round(v["seconds"], 3), expand_ms=round(v["expand_ms"], 1),
                       source="oracle-pinned" if (w is not None and kind != "probed") else "gpu"))
distinct = v2["distinct"] if v2["level"] else (v1["distinct"] if v1["level"] else mc.distinct)
slots = 1 << a.table_log2
print(json.dumps(dict(
    workload="VSR.tla BFS, ReplicaCount=5 ClientCount=1 Values={v1,v2} StartViewOnTimerLimit=2 (BASELINE configs[4]), VIEW+SYMMETRY, one MI355X: "
             "levels 1-%d materialised, %d virtual, %d streamed, %d probed" % (a.probe_from, a.probe_from + 1, a.probe_from + 2, a.probe_from + 3),
    distinct=distinct, seconds=round(dt, 3), distinct_states_per_s=round(distinct / dt, 1),
    setup_s=round(setup, 2), seen_set=dict(slots_log2=a.table_log2, bytes=16 * slots, load=round(distinct / slots, 4)),
    record_bytes_per_state=round(8.0 * words_last / n_last, 1), generated_per_expanded=round(v1["generated"] / n_last, 2) if v1["level"] else None,
    materialised=dict(levels=a.probe_from, distinct=n_mat, seconds=round(t_mat, 3), k_expand_ms=round(kms, 1)),
    violation=dict(level=p["level"], mask=p["viol_mask"], fp="%016x" % p["viol_fp"]) if p["viol_mask"] else None,
    hbm=dict(table_gb=round(16 * slots / 1e9, 1), records_a_gb=round(8 * a.words_a / 1e9, 1), records_b_gb=round(8 * a.words_b / 1e9, 1)),
    levels=levels[-6:])))
mc.close()

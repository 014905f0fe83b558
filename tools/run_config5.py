#!/usr/bin/env python3
"""BASELINE.json configs[4] — ReplicaCount=5, ClientCount=1, Values={v1,v2}, StartViewOnTimerLimit=2, the "288 GB/GPU FPSet sizing
stress" — on ONE MI355X as deep as its HBM allows, through the automatic level scheme (no level number, no buffer size from here): the
checker sizes itself from the free HBM, stores levels while the next one is predicted to fit (1-12), then goes on through the seen-set
alone (13 virtual, 14 streamed, 15 probed) until the seen-set would overfill.  Every figure of the 14 levels the CPU oracle's fixture
holds (tests/golden/oracle_levels_config5.json; level 14 from the memory-lean driver) is asserted; the probed level 15 has no CPU
counterpart (3.7e10 successors): GPU-sourced, labelled so.  Prints one JSON object.

    python tools/run_config5.py [--table-log2 N]      (0 = from the free memory: 2^32 slots; 33 = the round-3 figure, load 0.37)"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vsr_tlaplus_amd as vt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--table-log2", type=int, default=0)
ap.add_argument("--seed", type=lambda s: int(s, 0), default=0, help="fingerprint seed (vsrmc_model_set_fp_seed): the second-hash audit of the GPU-only probe of level 15")
a = ap.parse_args()
with open(os.path.join(ROOT, "tests", "golden", "oracle_levels_config5.json")) as f:
    g = json.load(f)
m = vt.Model.from_constants(R=5, C_=1, n=2, L=2)
if a.seed:
    m.set_fp_seed(a.seed)
t0 = time.time()
mc = vt.ModelChecker.auto(m, table_log2=a.table_log2)
setup = time.time() - t0
slots = 1 << int(mc.options.table_log2)
levels, stop, probed = [], None, None
t0 = time.time()
while stop is None:
    if mc.distinct > 0.85 * slots:
        stop = "seen-set-full"
        break
    # the next pass inserts one more level: will it overfill the seen-set?  (the last level's growth bounds the next one's)
    if len(levels) >= 2 and levels[-1]["kind"] != "stored" and mc.distinct + levels[-1]["new"] * levels[-1]["new"] / max(1, levels[-2]["new"]) > 0.85 * slots:
        stop = "seen-set-full (the next level would not fit: %d slots)" % slots
        break
    kind, d, b = mc.advance()
    if d["n_new"] == 0:
        stop = "exhausted"
        break
    w = g["levels"][d["level"] - 1] if d["level"] <= len(g["levels"]) else None
    if w is not None:
        assert (d["n_new"], d["generated"], d["deadlocks"], d["max_bag"]) == (w["new"], w["generated"], w["deadlocks"], w["max_bag"]), d["level"]
        assert [int(x) for x in d["act_generated"][1:16]] == w["act_generated"][1:16], d["level"]
        if kind == "deep" and g.get("fp_version") == 2 and not a.seed:
            assert ("%016x" % d["fp_xor"], "%016x" % d["fp_sum"]) == (w["fp_xor"], w["fp_sum"]), d["level"]
    levels.append(dict(level=d["level"], kind="stored" if kind == "level" else "seen-set only", new=d["n_new"], generated=d["generated"],
                       deadlocks=d["deadlocks"], max_bag=d["max_bag"], seconds=round(d["seconds"], 3), expand_ms=round(d["expand_ms"], 1),
                       record_words=d["record_words"], source="oracle-pinned" if w is not None else "gpu"))
    if b is not None:
        probed = dict(level=b["level"], generated=b["generated"], deadlocks=b["deadlocks"], viol_mask=b["viol_mask"], seconds=round(b["seconds"], 3), source="gpu")
    if mc.violation is not None:
        stop = "violation"
dt = time.time() - t0
stored = [lv for lv in levels if lv["kind"] == "stored"]
print(json.dumps(dict(
    fp_seed=hex(a.seed),
    config="BASELINE configs[4]: ReplicaCount=5 ClientCount=1 Values={v1,v2} StartViewOnTimerLimit=2, VIEW+SYMMETRY, AcknowledgedWriteNotLost",
    stop=stop, depth=mc.depth, distinct=mc.distinct, seconds=round(dt, 3), setup_s=round(setup, 2), distinct_states_per_s=round(mc.distinct / dt, 1),
    sized_from_free_hbm=dict(table_log2=int(mc.options.table_log2), frontier_words=int(mc.options.frontier_words), frontier_states=int(mc.options.frontier_states)),
    seen_set=dict(slots_log2=int(mc.options.table_log2), bytes=slots * 16, load=round(mc.distinct / slots, 3)),
    stored_levels=len(stored) + 1, record_bytes_per_state_last_stored=round(8.0 * stored[-1]["record_words"] / stored[-1]["new"], 1),
    probed=probed, violation=mc.violation, oracle_pinned_levels=len(g["levels"]), levels=levels)))
mc.close()

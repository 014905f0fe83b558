#!/usr/bin/env python3
"""After tools/merge_lean_levels.py has put the lean oracle's levels 22-23 and its probe of level 24 into
tests/golden/oracle_levels_config3.json: bring tests/golden/config3_violation.json (the counter-example fixture, whose level table
came partly from GPU runs) in line — every level figure must EQUAL the oracle's before its source is relabelled, the violating
fingerprint and the probe's generated count likewise.  Nothing is copied from the GPU to the oracle side."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")
with open(os.path.join(G, "oracle_levels_config3.json")) as f:
    o = json.load(f)
with open(os.path.join(G, "config3_violation.json")) as f:
    fx = json.load(f)
assert len(o["levels"]) >= len(fx["levels"]) == 23 and o.get("probe"), "the oracle fixture does not reach level 23 + probe yet"
for lv, ol in zip(fx["levels"], o["levels"]):
    assert (lv["level"], lv["n_new"], lv["generated"]) == (ol["level"], ol["new"], ol["generated"]), lv["level"]
    if lv.get("deadlocks") is not None:
        assert lv["deadlocks"] == ol["deadlocks"], lv["level"]
    lv["deadlocks"], lv["max_bag"], lv["source"] = ol["deadlocks"], ol["max_bag"], "oracle"
p = o["probe"]
assert fx["probe"]["generated"] == p["generated"] and fx["probe"]["deadlocks"] == p["deadlocks"], (fx["probe"], p)
assert fx["viol_fp"] == p["viol_fp"] and fx["viol_mask"] == p["viol_mask"], (fx["viol_fp"], p["viol_fp"])
assert fx["distinct_through_level_23"] == o["distinct"]
fx["probe"]["source"] = "oracle"
fx["probe"]["violating_successors_oracle"] = p["violating_successors"]
fx["levels_source"] = "levels 1-23, the probe of level 24 and the violating fingerprint: CPU oracle (oracle_levels_config3.json; levels 22-23 and the probe " \
                      "from the memory-lean driver oracle/vsr_oracle_lean); the 24-state trace: the GPU run, validated step by step by both CPU restatements"
with open(os.path.join(G, "config3_violation.json"), "w") as f:
    json.dump(fx, f, indent=1)
print("config3_violation.json: 23 levels + probe + violating fingerprint %s equal to the oracle's; relabelled" % p["viol_fp"])

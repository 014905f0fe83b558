#!/usr/bin/env python3
"""Re-derive what tests/golden/config{2,3}_violation.json hold besides the counter-example itself (CPU only, uses the oracle):

  * `viol_fp` = the ORACLE's fingerprint of the trace's last state under the current fingerprint function (`fp_version`);
  * `levels`  = the CPU oracle's per-level figures (tests/golden/oracle_levels_<config>*.json) as deep as the oracle went; deeper
                levels keep the figures of the GPU run that found the trace and are marked `"source": "gpu"`.

The traces themselves are behaviours found by the GPU BFS; tests/test_host_cpu.py and tests/test_config3_trace.py validate every
step of them with both CPU restatements.  Optional: --trace2 FILE / --trace3 FILE = output of tools/run_bfs.py holding a new
{"trace": [...]} line to install as the counter-example."""
import argparse
import glob
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def oracle_fixture(key):
    best = None
    for path in sorted(glob.glob(os.path.join(GOLDEN, "oracle_levels_%s*.json" % key))):
        with open(path) as f:
            d = json.load(f)
        if best is None or d.get("fp_version", 1) == orc.FP_VERSION:
            best = d
            best["file"] = os.path.basename(path)
    return best


def new_trace(path):
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line.startswith('{"trace"'):
                return json.loads(line)["trace"]
    raise SystemExit("no {\"trace\": ...} line in " + path)


def refresh(name, key, P, trace_file):
    path = os.path.join(GOLDEN, name)
    with open(path) as f:
        fx = json.load(f)
    if trace_file:
        fx["trace"] = new_trace(trace_file)
        fx["depth"] = len(fx["trace"])
    last = np.array([int(w, 16) for w in fx["trace"][-1]["words"]], dtype=np.uint64)
    fp, _ = orc.fingerprint(P, last)
    fx["viol_fp"] = "%016x" % fp
    fx["fp_version"] = orc.FP_VERSION
    o = oracle_fixture(key)
    if o:
        for i, lv in enumerate(o["levels"]):
            row = dict(level=lv["level"], n_new=lv["new"], generated=lv["generated"], deadlocks=lv["deadlocks"], max_bag=lv["max_bag"],
                       source="oracle")
            if i < len(fx["levels"]):
                old = fx["levels"][i]
                assert (old["n_new"], old["generated"], old["deadlocks"]) == (row["n_new"], row["generated"], row["deadlocks"]), (name, i)
                fx["levels"][i] = row
            else:
                fx["levels"].append(row)
        for lv in fx["levels"][len(o["levels"]):]:
            lv["source"] = "gpu"
        fx["levels_source"] = "levels 1-%d: CPU oracle (%s); deeper levels: the GPU run that found the trace" % (len(o["levels"]), o["file"])
    with open(path, "w") as f:
        json.dump(fx, f, indent=1)
    print(name, "viol_fp", fx["viol_fp"], "levels", len(fx["levels"]), fx.get("levels_source"))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--trace2")
    ap.add_argument("--trace3")
    a = ap.parse_args()
    refresh("config2_violation.json", "config2", orc.Params(3, 1, 2, 2), a.trace2)
    refresh("config3_violation.json", "config3", orc.Params(3, 1, 3, 3), a.trace3)

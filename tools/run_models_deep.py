#!/usr/bin/env python3
"""The two analysis models under their shipped cfgs (VR_STATE_TRANSFER.cfg, VR_APP_STATE.cfg: 3 replicas, two values, limit 2) as far as one
MI355X goes: the automatic level scheme to exhaustion, to a violation, or until the seen-set is 85 % full.  Their cfgs promise "no
violation"; the oracle fixtures pin the first 22 levels (asserted here), everything deeper is GPU-sourced and said so.  Every run is made
under two members of the fingerprint family (vsrmc_model_set_fp_seed): the per-level counts of the two runs must be equal.
    python tools/run_models_deep.py [--table-log2 33] [--max-seconds 600]   -> one JSON line per model"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--table-log2", type=int, default=33)
    ap.add_argument("--max-seconds", type=float, default=600.0)
    ap.add_argument("--models", default="model2,model3")
    ap.add_argument("--one-seed", action="store_true", help="only the default fingerprint function (no audit run)")
    ap.add_argument("--max-depth", type=int, default=0, help="stop after this level (0 = no bound): both seeds then end at the same level")
    a = ap.parse_args()
    import vsr_tlaplus_amd as vt
    for label, make in (("model2", vt.Model.second_model), ("model3", vt.Model.third_model)):
        if label not in a.models.split(","):
            continue
        with open(os.path.join(ROOT, "tests", "golden", "oracle_levels_%s.json" % label)) as f:
            g = json.load(f)
        p = g["params"]
        runs = []
        for seed in ((0,) if a.one_seed else (0, 0x5EED5EED5EED5EED)):
            m = make(R=p["R"], n=p["n"], L=p["L"], invariant_mask=p["inv_mask"])
            if seed:
                m.set_fp_seed(seed)
            mc = vt.ModelChecker.auto(m, table_log2=a.table_log2)
            rows, t0, stop = [], time.perf_counter(), None
            while stop is None:
                if time.perf_counter() - t0 > a.max_seconds:
                    stop = "max-seconds"
                    break
                if mc.distinct > 0.85 * (1 << a.table_log2):
                    stop = "seen-set-full"
                    break
                if a.max_depth and mc.depth >= a.max_depth:
                    stop = "max-depth"
                    break
                kind, d, b = mc.advance()
                if d["n_new"] == 0:
                    stop = "exhausted"
                    break
                rows.append((d["level"], d["n_new"], d["generated"], d["deadlocks"], kind))
                if d["level"] <= len(g["levels"]) and not seed:   # the fixture's counts are those of ITS fingerprint function (seed 0): model 2's level 26 is one
                    lv = g["levels"][d["level"] - 1]              # state short under it (the 64-bit collision of tests/test_fp_collision.py) — the other seed shows it below
                    assert (d["n_new"], d["generated"], d["deadlocks"]) == (lv["new"], lv["generated"], lv["deadlocks"]), d["level"]
                if mc.violation is not None:
                    stop = "violation"
            dt = time.perf_counter() - t0
            runs.append(dict(seed=hex(seed), stop=stop, depth=mc.depth, distinct=mc.distinct, seconds=round(dt, 3), rows=rows,
                             violation=mc.violation, stored_levels=sum(1 for r in rows if r[4] == "level") + 1, rebased=list(mc.rebased)))
            print(json.dumps(dict(run=g["label"], seed=hex(seed), stop=stop, depth=mc.depth, distinct=mc.distinct, seconds=round(dt, 3),
                                  rebased=[dict(level=r["level"], states=r["n"], seconds=round(r["seconds"], 2), launches=r["launches"]) for r in mc.rebased],
                                  kinds="".join("s" if r[4] == "level" else "d" for r in rows))), flush=True)
            mc.close()
        if a.one_seed:
            runs.append(runs[0])
        common = min(len(runs[0]["rows"]), len(runs[1]["rows"]))   # (a time bound can end the two runs one pass apart)
        diff = [(x[:4], y[:4]) for x, y in zip(runs[0]["rows"][:common], runs[1]["rows"][:common]) if x[:4] != y[:4]]
        same = not diff
        last = runs[0]["rows"][-1]
        print(json.dumps(dict(model=g["label"], stop=runs[0]["stop"], depth=runs[0]["depth"], distinct=runs[0]["distinct"], seconds=runs[0]["seconds"],
                              distinct_states_per_s=round(runs[0]["distinct"] / runs[0]["seconds"], 1), oracle_pinned_levels=len(g["levels"]),
                              stored_levels=runs[0]["stored_levels"], rebased_at=[r["level"] for r in runs[0]["rebased"]], last_level=dict(level=last[0], n_new=last[1], generated=last[2]),
                              violation=runs[0]["violation"], second_seed=dict(seed=runs[1]["seed"], counts_equal=same, levels_compared=common + 1, seconds=runs[1]["seconds"],
                                                                               depth=runs[1]["depth"], distinct=runs[1]["distinct"], first_differences=diff[:3]),
                              collision_estimate_n2_over_2_65=round(float(runs[0]["distinct"]) ** 2 / 2.0 ** 65, 3),
                              level_sizes=[r[1] for r in runs[0]["rows"]])))
        if not same:
            print("WARNING: per-level counts differ between the two fingerprint functions from level %d on: at %.2g states a 64-bit collision is no longer "
                  "unlikely (n^2 / 2^65 above) — TLC prints the same warning" % (diff[0][0][0], runs[0]["distinct"]))


if __name__ == "__main__":
    main()

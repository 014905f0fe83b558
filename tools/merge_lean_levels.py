#!/usr/bin/env python3
"""Extend a whole-workload fixture of the CPU ORACLE (tests/golden/oracle_levels_*.json, written by tools/make_oracle_levels.py from
oracle/vsr_oracle_mt) with the deeper levels of the memory-lean driver (oracle/vsr_oracle_lean: the same restatement, levels beyond
a stored base level regenerated from it) — never with anything from the GPU path.  The lean run's figures for the levels the
fixture already holds must equal the fixture's (two drivers, one answer) before anything is appended.

    python tools/merge_lean_levels.py tests/golden/oracle_levels_config3.json gpurun_out/lean_config3.jsonl "<command line of the lean run>" """
import json
import sys

KEYS = ("level", "new", "generated", "ties", "deadlocks", "max_bag", "fp_xor", "fp_sum", "act_generated")


def main():
    fixture, lean_path, cmd = sys.argv[1:4]
    with open(fixture) as f:
        g = json.load(f)
    lines = [json.loads(l) for l in open(lean_path) if l.strip()]
    levels = [d for d in lines if "level" in d]
    probe = [d for d in lines if "probe_level" in d]
    summary = [d for d in lines if d.get("summary")]
    have = len(g["levels"])
    for old, new in zip(g["levels"], levels):
        assert {k: old[k] for k in KEYS} == {k: new[k] for k in KEYS}, ("the two oracle drivers disagree at level %d" % old["level"])
    added = 0
    for d in levels[have:]:
        e = {k: d[k] for k in KEYS}
        e["seconds"] = d["seconds"]
        e["driver"] = "vsr_oracle_lean"
        g["levels"].append(e)
        added += 1
    g["depth"] = len(g["levels"])
    g["distinct"] = sum(lv["new"] for lv in g["levels"])
    g["generated"] = sum(lv["generated"] for lv in g["levels"])
    g["max_bag"] = max(lv["max_bag"] for lv in g["levels"])
    g["count_only_from"] = None
    if probe:
        p = probe[0]
        g["probe"] = dict(level=p["probe_level"], generated=p["generated"], deadlocks=p["deadlocks"], violating_successors=p["violating_successors"],
                          viol_fp=p["viol_fp"], viol_mask=p["viol_mask"], seconds=p["seconds"],
                          note="the level after the last one: successors looked up, not inserted; a successor in no earlier level gets its invariants "
                               "checked (TLC reports a violation while expanding the level before); viol_fp = smallest violating fingerprint")
        g["viol_mask"], g["viol_fp"], g["stop"] = p["viol_mask"], p["viol_fp"], ("violation" if p["viol_mask"] else "probe")
    if summary:
        g["lean_run"] = {k: summary[0][k] for k in ("seconds", "threads", "base_level", "slots", "stop", "depth", "distinct") if k in summary[0]}
    g["source"] = g["source"].split(" || ")[0] + " || levels %d-%d and the probe of level %d: oracle/vsr_oracle_lean (memory-lean driver over the same " \
        "restatement: levels beyond the base level are regenerated from its records; its per-level figures for levels 1-%d equal the ones above) — `%s`" \
        % (have + 1, len(g["levels"]), len(g["levels"]) + 1, have, cmd)
    with open(fixture, "w") as f:
        json.dump(g, f, indent=1)
    print("levels 1-%d checked equal, %d appended, probe %s" % (have, added, "yes" if probe else "no"))


if __name__ == "__main__":
    main()

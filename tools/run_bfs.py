#!/usr/bin/env python3
"""Exploratory driver: run the GPU BFS on one config and print one JSON line per level."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vsr_tlaplus_amd as vt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("R", type=int); ap.add_argument("C", type=int); ap.add_argument("n", type=int); ap.add_argument("L", type=int)
ap.add_argument("--table-log2", type=int, default=30)
ap.add_argument("--frontier-words-log2", type=float, default=31)
ap.add_argument("--frontier-states-log2", type=float, default=27)
ap.add_argument("--pending-log2", type=int, default=28)
ap.add_argument("--max-seconds", type=float, default=120)
ap.add_argument("--max-depth", type=int, default=10 ** 6)
ap.add_argument("--inv-mask", type=int, default=1)
ap.add_argument("--no-symmetry", action="store_true")
ap.add_argument("--assume-commit-number", action="store_true")
ap.add_argument("--no-trace", action="store_true")
ap.add_argument("--exact-ties", action="store_true")
ap.add_argument("--host-frontier", type=int, default=0, help="bit mask: buffer 0 / 1 in pinned host memory")
ap.add_argument("--probe2-at", type=int, default=0, help="when the newest level is N-1: virtual level N + probe level N+1")
ap.add_argument("--frontier-b-words-log2", type=float, default=0)
ap.add_argument("--trace-log2", type=float, default=0)
ap.add_argument("--probe-at", type=int, default=0)
a = ap.parse_args()
m = vt.Model.from_constants(R=a.R, C_=a.C, n=a.n, L=a.L, symmetry=not a.no_symmetry, invariant_mask=a.inv_mask,
                            assume_commit_number=a.assume_commit_number)
t0 = time.time()
mc = vt.ModelChecker(m, table_log2=a.table_log2, frontier_words=int(2 ** a.frontier_words_log2),
                     frontier_states=int(2 ** a.frontier_states_log2), pending_entries=1 << a.pending_log2,
                     keep_trace=not a.no_trace, exact_ties=a.exact_ties, host_frontier=a.host_frontier,
                     frontier_words_b=int(2 ** a.frontier_b_words_log2) if a.frontier_b_words_log2 else 0,
                     trace_entries=int(2 ** a.trace_log2) if a.trace_log2 else 0)
print(json.dumps(dict(setup_seconds=round(time.time() - t0, 3))))
t0 = time.time()
why = "exhausted"
try:
    while True:
        if mc.level >= a.max_depth:
            why = "max-depth"; break
        if time.time() - t0 > a.max_seconds:
            why = "max-seconds"; break
        if a.probe2_at and mc.level + 1 == a.probe2_at:
            v, p = mc.probe2()
            for d in (v, p):
                print(json.dumps({k: (round(x, 3) if isinstance(x, float) else x) for k, x in d.items() if k not in ("act_generated", "phase_cycles")}), flush=True)
            why = "probe2"
            if v["viol_mask"] or p["viol_mask"]:
                tr = mc.probe_trace()
                print("probe trace length", len(tr), [t[0] for t in tr])
                print(m.format_state(tr[-1][1]))
                print(json.dumps(dict(trace=[dict(action=t[0], words=["%016x" % int(w) for w in t[1]]) for t in tr])))
            break
        if a.probe_at and mc.level + 1 == a.probe_at:
            p = mc.probe()
            print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in p.items() if k not in ("act_generated", "phase_cycles")}))
            why = "probe"
            if p["viol_mask"]:
                tr = mc.probe_trace()
                print("probe trace length", len(tr), [t[0] for t in tr])
                print(m.format_state(tr[-1][1]))
                print(json.dumps(dict(trace=[dict(action=t[0], words=["%016x" % int(w) for w in t[1]]) for t in tr])))
            break
        d = mc.step()
        if d["n_new"]:
            acts = d.pop("act_generated"); ph = d.pop("phase_cycles")
            if d["level"] >= 26:
                print(json.dumps(dict(level=d["level"], act_generated=acts, phase_cycles=ph)))
            print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in d.items()
                              if k in ("level", "frontier", "generated", "n_new", "distinct", "deadlocks", "pending", "probes",
                                       "words_new", "max_bag", "seconds", "expand_ms", "materialize_ms", "viol_mask")}), flush=True)
        if d["n_new"] == 0:
            break
        if mc.violation:
            why = "violation"; break
except vt.VsrmcError as e:
    why = "error: %s" % e
dt = time.time() - t0
print(json.dumps(dict(summary=True, stop=why, depth=mc.level, distinct=mc.distinct, seconds=round(dt, 3),
                      states_per_s=round(mc.distinct / dt, 1), violation=mc.violation)))
if mc.violation and not a.no_trace and not mc.violation.get("probed"):
    tr = mc.trace(mc.violation["level"], mc.violation["index"])
    print("trace length", len(tr), [t[0] for t in tr])
    print(m.format_state(tr[-1][1]))
    print(json.dumps(dict(trace=[dict(action=t[0], words=["%016x" % int(w) for w in t[1]]) for t in tr])))

#!/usr/bin/env python3
"""Whole-workload fixtures from the CPU ORACLE (never from the GPU path): runs oracle/build/vsr_oracle_mt — the multi-threaded
driver over oracle/vsr_oracle.cpp, the line-cited restatement of /root/reference/vsr-revisited/paper/VSR.tla — on one
configuration and writes its per-level figures (new, generated, per-action generated, deadlocks, largest bag, xor and sum of
the new canonical fingerprints), the violating fingerprint and the stop reason as JSON.

    python tools/make_oracle_levels.py --R 3 --C 1 --n 2 --L 2 --out gpurun_out/oracle_levels_config2.json
    python tools/make_oracle_levels.py --R 3 --C 1 --n 3 --L 3 --count-only-from 21 --max-seconds 600 --out ...

The big ones need the GPU box's host (hundreds of hardware threads, > 100 GB of RAM): run there through gpurun, then copy the
file into tests/golden/ (tests/test_gpu_parity.py::test_whole_workload_against_the_oracle and bench.py read it from there).
Needs nothing of /root/reference at run time."""
import argparse
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FP_VERSION = 2          # oracle/vsr_oracle.hpp FP_VERSION (only used for the partial file of a killed run)


def main():
    ap = argparse.ArgumentParser()
    for k in "RCnL":
        ap.add_argument("--" + k, type=int, required=True)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--max-depth", type=int, default=0)
    ap.add_argument("--max-seconds", type=float, default=0)
    ap.add_argument("--count-only-from", type=int, default=0)
    ap.add_argument("--inv-mask", type=int, default=0, help="default: the shipped cfg's invariants (model 1: 1, model 2: 14, model 3: 30)")
    ap.add_argument("--model", type=int, default=1, help="1 = VSR.tla (vsr_oracle_mt), 2 = analysis/03-state-transfer/VR_STATE_TRANSFER.tla (vrst_oracle_mt), 3 = analysis/04-application-state/VR_APP_STATE.tla (vras_oracle_mt)")
    ap.add_argument("--no-symmetry", action="store_true")
    ap.add_argument("--assume-commit-number", action="store_true",
                    help="policy for VSR.tla:421 (`m.commit`, a field PrepareMsg does not have: TLC aborts there when ClientCount >= 2): read m.commit_number instead")
    ap.add_argument("--label", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    exe = os.path.join(ROOT, "oracle", "build", {1: "vsr_oracle_mt", 2: "vrst_oracle_mt", 3: "vras_oracle_mt"}[a.model])
    a.inv_mask = a.inv_mask or {1: 1, 2: 14, 3: 30}[a.model]
    cmd = [exe, str(a.R), str(a.C), str(a.n), str(a.L), "--inv-mask", str(a.inv_mask)]
    if a.threads:
        cmd += ["--threads", str(a.threads)]
    if a.max_depth:
        cmd += ["--max-depth", str(a.max_depth)]
    if a.max_seconds:
        cmd += ["--max-seconds", str(a.max_seconds)]
    if a.count_only_from:
        cmd += ["--count-only-from", str(a.count_only_from)]
    if a.no_symmetry:
        cmd += ["--no-symmetry"]
    if a.assume_commit_number:
        cmd += ["--assume-commit-number"]
    t0 = time.time()
    levels = []

    def write(summary):
        """(re)write the output file; before the oracle's summary line arrives the file says stop = "running" — a run that is
        killed (timeout) still leaves every completed level behind"""
        out = dict(
            source="CPU oracle: oracle/%s (multi-threaded driver over oracle/%s) — " % (
                       {1: ("vsr_oracle_mt", "vsr_oracle.cpp, the restatement of VSR.tla"),
                        2: ("vrst_oracle_mt", "vrst_oracle.cpp, the restatement of VR_STATE_TRANSFER.tla"),
                        3: ("vras_oracle_mt", "vras_oracle.cpp, the restatement of VR_APP_STATE.tla")}[a.model]) +
                   "`%s`, %s threads on %s (%d logical CPUs), %.1f s; written by tools/make_oracle_levels.py"
                   % (" ".join([os.path.basename(exe)] + cmd[1:]), summary.get("threads", a.threads or "all"), platform.processor() or platform.machine(),
                      os.cpu_count() or 0, time.time() - t0),
            label=a.label or "(%d,%d,%d values,%d)" % (a.R, a.C, a.n, a.L),
            model=a.model, params=dict(R=a.R, C=a.C, n=a.n, L=a.L, symmetry=(not a.no_symmetry) and a.model == 1, inv_mask=a.inv_mask,
                        **({"assume_commit_number": True} if a.assume_commit_number else {})),
            stop=summary.get("stop", "running"), depth=summary.get("depth", len(levels)),
            distinct=summary.get("distinct", sum(lv["new"] for lv in levels)),
            generated=summary.get("generated", sum(lv["generated"] for lv in levels)),
            max_bag=summary.get("max_bag", max([lv["max_bag"] for lv in levels] or [0])), viol_mask=summary.get("viol_mask", 0),
            viol_fp=summary.get("viol_fp", "0" * 16), error=summary.get("error", ""), fp_version=summary.get("fp_version", FP_VERSION),
            count_only_from=a.count_only_from or None, oracle_seconds=summary.get("seconds", time.time() - t0),
            oracle_states_per_s=summary.get("states_per_s", 0.0),
            levels=[{k: v for k, v in lv.items() if k != "distinct"} for lv in levels])
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out + ".tmp", "w") as f:
            json.dump(out, f, indent=1)
        os.replace(a.out + ".tmp", a.out)
        return out

    summary = None
    with subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True) as p:
        for line in p.stdout:
            d = json.loads(line)
            if d.get("summary"):
                summary = d
            else:
                levels.append(d)
                print("level %d: new %d generated %d (%.1f s)" % (d["level"], d["new"], d["generated"], d["seconds"]), file=sys.stderr, flush=True)
                if d["seconds"] > 1.0:
                    write({})
    if summary is None:
        write({})
        raise SystemExit("the oracle did not finish (exit code %s); the completed levels are in %s" % (p.returncode, a.out))
    out = write(summary)
    print(json.dumps({k: v for k, v in out.items() if k != "levels"}))


if __name__ == "__main__":
    main()

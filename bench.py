#!/usr/bin/env python3
"""bench.py — distinct states/s of the GPU BFS model checker on BASELINE.json's 1-GPU configuration.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one complete run of the hot path: level-synchronous BFS of VSR.tla under the shipped VSR.cfg constants
(BASELINE.json configs[1]: ReplicaCount=3, ClientCount=1, Values={v1,v2}, StartViewOnTimerLimit=2, VIEW + SYMMETRY on,
INVARIANT AcknowledgedWriteNotLost) from Init until the level that contains the first invariant violation is complete —
28 levels, 319 228 361 distinct states, 885 M successors generated; the seen-set is cleared at the start of every step,
HBM allocations are reused.  Inputs are the model itself (deterministic, no data files): "synthetic" in the contract's
sense.  Timed region: barrier + torch.cuda.synchronize() on both sides, max over ranks, exactly K steps.

N > 1: the seen-set is sharded by the high fingerprint bits, one rank per GPU, successors routed to their owner with
an all-to-all per level (vsr-tlaplus_amd/sharded.py); total work is fixed as N grows ("strong").

Extra objects on the JSON line: `roofline` for the dominant kernel (k_expand, HIP-event time on the checker's stream)
and `cpu_baseline` = the CPU oracle (a port, not TLC) timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIG = dict(R=3, C=1, n=2, L=2)                     # BASELINE.json configs[1] = /root/reference/.../VSR.cfg:4-8
FP_VERSION = 2                                        # fingerprint function of this build (oracle/vsr_oracle.hpp FP_VERSION)


def load_expect():
    """What a run must reproduce, from the CPU ORACLE's run of the whole workload (tools/make_oracle_levels.py ->
    tests/golden/oracle_levels_config2*.json): level sizes, generated / deadlock counts and largest bags of all 28 levels (valid for
    any fingerprint function), and — from a fixture made with this build's fingerprint function — the per-level xor / sum of the
    fingerprints and the violating fingerprint."""
    g = os.path.join(ROOT, "tests", "golden")
    for name in ("oracle_levels_config2.json", "oracle_levels_config2_fpv1.json"):
        if os.path.exists(os.path.join(g, name)):
            with open(os.path.join(g, name)) as f:
                d = json.load(f)
            same_fp = d.get("fp_version", 1) == FP_VERSION
            return dict(distinct=d["distinct"], depth=d["depth"], levels=d["levels"], fixture=name,
                        viol_fp=int(d["viol_fp"], 16) if same_fp else None, checksums=same_fp)
    raise SystemExit("no oracle fixture for the bench workload under tests/golden/")


EXPECT = load_expect()
HBM_PEAK_GBS = 8000.0                                 # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
TABLE_LOG2 = int(os.environ.get("VSR_BENCH_TABLE_LOG2", 31))   # seen-set: 2^31 slots x 16 B = 32 GiB of the 288 GB (load 0.15 at the end)


def usable_cpus():
    """Hardware threads this process may actually use: the affinity mask, capped by the container's CPU quota (cgroup v2
    cpu.max / v1 cfs quota).  The GPU box shows 256 logical CPUs but grants 16 CPUs' worth of time: 256 oracle threads then
    run SLOWER than 16 (measured: 1.0e6 vs 2.4e6 states/s)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return min(n, 256)


def cpu_baseline(seconds=15.0):
    """CPU oracle (oracle/vsr_oracle_mt: the restatement of VSR.tla spread over std::thread workers sharing one lock-free
    seen-set, the way TLC spreads Worker threads over an FPSet) on the same config, on every CPU the container may use, for a
    bounded time."""
    exe = os.path.join(ROOT, "oracle", "build", "vsr_oracle_mt")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    threads = usable_cpus()
    out = subprocess.run([exe, str(CONFIG["R"]), str(CONFIG["C"]), str(CONFIG["n"]), str(CONFIG["L"]), "--threads", str(threads),
                          "--max-seconds", str(seconds), "--quiet"], capture_output=True, text=True, check=True).stdout
    s = json.loads(out.strip().splitlines()[-1])
    return dict(value=round(s["states_per_s"], 1), unit="distinct states/s", cores=int(s["threads"]), kind="port",
                sample="oracle/vsr_oracle_mt (multi-threaded C++ restatement of VSR.tla, not TLC) on the same config for "
                       "%.0f s: %d distinct states, %d BFS levels" % (s["seconds"], s["distinct"], s["depth"]))


def run_single(args):
    import torch
    import vsr_tlaplus_amd as vt
    torch.cuda.set_device(0)
    m = vt.Model.from_constants(R=CONFIG["R"], C_=CONFIG["C"], n=CONFIG["n"], L=CONFIG["L"])
    mc = vt.ModelChecker(m, device=0, table_log2=TABLE_LOG2, frontier_words=1 << 32, frontier_states=1 << 27,
                         pending_entries=1 << 28, keep_trace=True, trace_entries=1 << 29)
    S = dict(expand_ms=0.0, mat_ms=0.0, launches=0, alg_bytes=0.0, distinct=0, generated=0, ttfv=[], words=0)

    def one_run(record, verify=False):
        mc.reset()
        t0 = time.perf_counter()
        cur_words = int(m.layout.fixed_words) + int(m.layout.permutations)      # Init record, device layout
        while True:
            d = mc.step()
            want = EXPECT["levels"][d["level"] - 1] if d["n_new"] else None     # every run, every level: the oracle's figures
            assert want is None or (d["n_new"], d["generated"], d["deadlocks"], d["max_bag"]) == \
                (want["new"], want["generated"], want["deadlocks"], want["max_bag"]), (d["level"], d["n_new"], d["generated"])
            if verify and want is not None:                                    # untimed run: per-action counts, fingerprint checksums
                assert [int(x) for x in d["act_generated"][1:16]] == want["act_generated"][1:16], d["level"]
                x, sm, cnt = mc.level_checksum()
                assert cnt == want["new"]
                if EXPECT["checksums"]:
                    assert ("%016x" % x, "%016x" % sm) == (want["fp_xor"], want["fp_sum"]), d["level"]
            if record and d["frontier"]:
                S["expand_ms"] += d["expand_ms"]
                S["mat_ms"] += d["materialize_ms"]
                S["launches"] += 1
                # algorithmic bytes of one k_expand launch: every frontier record read once, one 8-byte
                # key read per generated successor, one 8-byte key write per newly inserted fingerprint, every new record
                # written once (single-pass levels: k_expand also materialises the successors)
                S["alg_bytes"] += 8.0 * cur_words + 8.0 * d["generated"] + 8.0 * d["n_new"] + 8.0 * d["record_words"]
                S["generated"] += d["generated"]
                S["words"] += d["record_words"]
            cur_words = d["record_words"]
            if d["n_new"] == 0 or mc.violation is not None:
                break
        if mc.violation is not None:
            tr = mc.trace(mc.violation["level"], mc.violation["index"])       # counter-example reconstructed = found
            assert len(tr) == mc.violation["level"]
        dt = time.perf_counter() - t0
        assert mc.distinct == EXPECT["distinct"] and mc.level == EXPECT["depth"], (mc.distinct, mc.level)
        assert mc.violation and (EXPECT["viol_fp"] is None or mc.violation["fp"] == EXPECT["viol_fp"])
        if record:
            S["distinct"] += mc.distinct
            S["ttfv"].append(dt)

    if not args.no_verify:
        one_run(False, verify=True)                                            # untimed: the whole workload against the oracle fixture
    for _ in range(args.warmup):
        one_run(False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_run(True)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    mc.close()                                                                 # the HBM goes to the config-3 leg
    return elapsed, S, m


def config3_to_violation(dump_trace=None):
    """BASELINE configs[2] = the reference README's defect configuration (3 replicas, {v1,v2,v3}, limit 3; README:13-18) on ONE GPU,
    BFS to its first violation at depth 24 — everything in HBM: levels 1-21 are materialised (level 21: 261 M states, 91 GB of
    records), level 22 is a VIRTUAL level (seen-set entries only, regenerated slice by slice), level 23 is streamed through a
    scratch buffer (inserted, never kept), level 24 is PROBED (vsrmc_checker_probe3); the counter-example is reconstructed in the timed region.  Untimed setup (~245 GB of
    device allocations), one timed pass.  Levels 1-21 — every level that is stored — are asserted against the CPU
    oracle's fixture (tests/golden/oracle_levels_config3.json), the virtual / streamed levels 22-23 against
    tests/golden/config3_violation.json (GPU runs of two rounds, two fingerprint functions, three level schemes)."""
    import numpy as np
    import vsr_tlaplus_amd as vt
    with open(os.path.join(ROOT, "tests", "golden", "oracle_levels_config3.json")) as f:
        g = json.load(f)
    with open(os.path.join(ROOT, "tests", "golden", "config3_violation.json")) as f:
        fx = json.load(f)
    deep = fx["levels"]
    m = vt.Model.from_constants(R=3, C_=1, n=3, L=3)
    t0 = time.perf_counter()
    mc = vt.ModelChecker(m, device=0, table_log2=32, frontier_words=int(12.8e9), frontier_words_b=int(7.0e9), frontier_states=int(2.85e8),
                         pending_entries=1 << 16)
    try:
        setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        kernel_ms, alg_bytes, gen = 0.0, 0.0, 0
        cur_words = int(m.layout.fixed_words) + int(m.layout.permutations)
        while mc.level < 21:
            d = mc.step()
            lv = d["level"]
            if lv <= len(g["levels"]):
                want = g["levels"][lv - 1]
                assert (d["n_new"], d["generated"], d["deadlocks"], d["max_bag"]) == (want["new"], want["generated"], want["deadlocks"], want["max_bag"]), lv
            else:
                assert (d["n_new"], d["generated"]) == (deep[lv - 1]["n_new"], deep[lv - 1]["generated"]), lv
            assert d["viol_mask"] == 0
            kernel_ms += d["expand_ms"]
            alg_bytes += 8.0 * cur_words + 8.0 * d["generated"] + 8.0 * d["n_new"] + 8.0 * d["record_words"]
            cur_words = d["record_words"]
            gen += d["generated"]
        t_mat = time.perf_counter() - t0
        n_mat = mc.distinct
        v1, v2, p = mc.probe3()
        tr = mc.probe_trace()
        dt = time.perf_counter() - t0
        for v in (v1, v2):
            assert (v["n_new"], v["generated"], v["viol_mask"]) == (deep[v["level"] - 1]["n_new"], deep[v["level"] - 1]["generated"], 0), v
        assert p["level"] == 24 and p["viol_mask"] == 1 and len(tr) == 24 and p["generated"] == fx["probe"]["generated"]
        assert fx.get("fp_version") != FP_VERSION or p["viol_fp"] == int(fx["viol_fp"], 16)
        fps, _ = m.fingerprints(tr[-1][1], np.array([0, len(tr[-1][1])], dtype=np.uint64))
        assert int(fps[0]) == p["viol_fp"]                                   # the reconstructed path ends in the reported violator
        same_trace = None                                                    # the counter-example is a function of the state space alone
        if fx.get("fp_version") == FP_VERSION and fx.get("trace"):          # (min-merged keys): the same 24 states as the round's host-frontier run
            same_trace = [(a, ["%016x" % int(w) for w in rec]) for a, rec in tr] == [(t["action"], t["words"]) for t in fx["trace"]]
        if dump_trace:                                                       # refresh of tests/golden/config3_violation.json (tools/refresh_violation_fixtures.py)
            with open(dump_trace, "w") as f:
                f.write(json.dumps(dict(trace=[dict(action=a, words=["%016x" % int(w) for w in rec]) for a, rec in tr])) + "\n")
        out = dict(workload="VSR.tla BFS, ReplicaCount=3 ClientCount=1 Values={v1,v2,v3} StartViewOnTimerLimit=3 (BASELINE configs[2], README:13-18), "
                            "VIEW+SYMMETRY, to the first violation at depth 24, all in HBM: levels 1-21 materialised, 22 virtual, 23 streamed, 24 probed",
                   time_to_first_violation_s=round(dt, 4), depth=24, distinct_through_level_23=v2["distinct"],
                   distinct_states_per_s=round(v2["distinct"] / dt, 1), generated=gen + v1["generated"] + v2["generated"] + p["generated"],
                   setup_s=round(setup, 2), oracle_pinned_levels=len(g["levels"]), pcie_bound=False, trace_equals_fixture=same_trace,
                   materialised=dict(levels=21, distinct=n_mat, seconds=round(t_mat, 4), states_per_s=round(n_mat / t_mat, 1),
                                     k_expand_ms=round(kernel_ms, 2), roofline_frac=round(alg_bytes / (kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 5)),
                   probe3=dict(virtual_22_s=round(v1["seconds"], 4), virtual_23_s=round(v2["seconds"], 4), probe_24_s=round(p["seconds"], 4),
                               k_expand_ms=round(v1["expand_ms"] + v2["expand_ms"] + p["expand_ms"], 2),
                               slices=v2["pending"] >> 32, sub_slices=v2["pending"] & 0xFFFFFFFF, expansions=dict(level_21=2, level_22=1, level_23=1)))
    finally:
        mc.close()
    return out


def run_config3(args):
    """--workload config3: only the README defect configuration (BASELINE configs[2]) to its first violation, as the bench line."""
    c3 = config3_to_violation(os.environ.get("VSR_BENCH_DUMP_TRACE"))
    dt = c3["time_to_first_violation_s"]
    print(json.dumps({
        "metric": "time-to-first-violation, VSR 3-replica README defect config (BFS, trace reconstructed)", "value": round(dt, 3), "unit": "s",
        "n_gpus": 1, "steps": 1, "warmup": 0, "ms_per_step": round(1e3 * dt, 1), "higher_is_better": False, "scaling": "strong",
        "vs_baseline": None, "dtype": "u64", "data": "synthetic", "config": {"workload": c3["workload"]}, "detail": c3}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the untimed verification run (profiling: one run = 27 k_expand launches)")
    ap.add_argument("--no-config3", action="store_true", help="skip the config-3 leg (the README defect configuration to its depth-24 violation; ~245 GB of HBM)")
    ap.add_argument("--workload", choices=["config2", "config3"], default="config2",
                    help="config2 (default) = BASELINE's 1-GPU configuration; config3 = only the README defect config to its violation")
    args = ap.parse_args()
    if args.workload == "config3":
        return run_config3(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("VSR_BENCH_SHARDED"):   # VSR_BENCH_SHARDED=1: the N > 1 leg on one rank
        from vsr_tlaplus_amd import sharded_bench
        return sharded_bench.main(args, CONFIG, EXPECT)
    elapsed, S, m = run_single(args)
    value = S["distinct"] / elapsed
    avg_launch_s = S["expand_ms"] / 1e3 / max(1, S["launches"])
    achieved = S["alg_bytes"] / max(1, S["launches"]) / avg_launch_s / 1e9
    g = S["generated"] / S["distinct"]
    s_bytes = 8.0 * S["words"] / S["distinct"]
    out = {
        "metric": "distinct states/sec (whole node), VSR 3-replica", "value": round(value, 1), "unit": "distinct states/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": "VSR.tla BFS, ReplicaCount=3 ClientCount=1 Values={v1,v2} StartViewOnTimerLimit=2 "
                               "(BASELINE configs[1] = shipped VSR.cfg), VIEW+SYMMETRY, to first violation: 28 levels, "
                               "319228361 distinct states", "table_slots_log2": TABLE_LOG2, "trace": "predecessor pointers in the seen-set, counter-example reconstructed in the timed region"},
        "time_to_first_violation_s": round(sum(S["ttfv"]) / len(S["ttfv"]), 4),
        "generated_per_distinct": round(g, 3), "record_bytes": round(s_bytes, 1),
        "roofline": {"bound": "hbm", "kernel": "k_expand", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                     "avg_launch_ms": round(1e3 * avg_launch_s, 4), "launches": S["launches"],
                     "B_alg_per_state": round(2 * s_bytes + 8 * g + 8, 1),
                     "kernel_ms_per_step": {"k_expand": round(S["expand_ms"] / args.steps, 3)}},
    }
    # HBM traffic of the same kernel from the committed PMC passes (FETCH_SIZE / WRITE_SIZE cannot be read live)
    tpath = os.path.join(ROOT, "profiles", "r02_traffic.json")
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "r01g_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as f:
            t = json.load(f)
        out["roofline"]["traffic"] = t["hbm_bytes_per_launch"]
        out["roofline"]["traffic_unit"] = "bytes per launch (PMC, %s)" % t["source"].split(" (")[0]
        out["roofline"]["alg_bytes_per_launch"] = round(S["alg_bytes"] / max(1, S["launches"]))
    if not args.no_config3:
        try:
            out["config3"] = config3_to_violation()
        except Exception as e:                                   # e.g. less than 245 GB of free HBM: the headline figures above stay valid
            out["config3"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()

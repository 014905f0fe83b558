"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the sharded level loop (vsr_tlaplus_amd/sharded.py) over a
CPU stand-in engine built on the oracle; the union of the shards' per-level fingerprint sets must equal the
single-process oracle BFS, for any world size.  (`-m gpu` has the same test over the HIP engine.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(engine, world, params, max_depth, tmp_path, port, replicate_below=0, **legs):
    out = str(tmp_path / ("shard_%s_w%d" % (engine, world)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker.py"), engine] + \
          [str(x) for x in params] + [str(max_depth), out, str(replicate_below)]
    env = dict(os.environ, OMP_NUM_THREADS="1", **{k: str(v) for k, v in legs.items()})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return [json.load(open("%s.rank%d.json" % (out, k))) for k in range(world)]


def check_probe_and_checkpoint(ranks, params, inv_mask, depth):
    """the checkpoint leg ran, the probed level reports the oracle's violation, the counter-example replays in the oracle"""
    from oracle import orc
    P = orc.Params(*params, invariant_mask=inv_mask)
    ob = orc.Bfs(P)
    while ob.info["depth"] < depth:
        ob.step()
    assert ob.info["viol_mask"] == inv_mask
    words, off = ob.frontier()                                          # level `depth`: the violators' smallest fingerprint
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    assert all(r["restored"] for r in ranks)
    pr = ranks[0]["probe"]
    assert all(r["probe"] == pr for r in ranks)                          # every rank reports the same violation and path
    assert pr["level"] == depth and pr["viol_fp"] == "%016x" % viol and pr["viol_mask"] == inv_mask
    assert pr["generated"] == ob.info["generated"] and pr["deadlocks"] == ob.info["deadlocks"]
    path = [int(f, 16) for f in pr["fps"]]
    rec = orc.init_record(P)
    assert len(path) == depth and orc.fingerprint(P, rec)[0] == path[0]
    for f in path[1:]:
        nxt = [s for s in orc.successors(P, rec) if s["fp"] == f]
        assert nxt, "a state of the counter-example is not a successor of its predecessor"
        rec, inv = nxt[0]["words"], nxt[0]["inv"]
    assert inv == inv_mask


def check_against_oracle(ranks, params, max_depth):
    from oracle import orc
    from vsr_tlaplus_amd import sharded
    R, C_, n, L = params
    P = orc.Params(R, C_, n, L)
    ob = orc.Bfs(P)
    world = len(ranks)
    nlev = len(ranks[0]["levels"])
    assert all(len(r["levels"]) == nlev for r in ranks)
    for li in range(nlev):
        want = [int(x) for x in ob.level_fps(li + 1)]
        got = []
        assert all(r["levels"][li]["replicated"] == ranks[0]["levels"][li]["replicated"] for r in ranks)
        if ranks[0]["levels"][li]["replicated"]:                    # replicated phase: every rank holds the whole level
            for r in ranks:
                assert sorted(int(x, 16) for x in r["levels"][li]["fps"]) == want, "level %d rank %d" % (li + 1, r["rank"])
            got = want
        for r in ranks:
            if not r["levels"][li]["replicated"]:
                got += [int(x, 16) for x in r["levels"][li]["fps"]]     # records live with their generator, not their owner
        assert sorted(got) == want, "level %d" % (li + 1)
        assert ranks[0]["levels"][li]["n_new"] == len(want)
        if li > 0:
            assert ranks[0]["levels"][li]["generated"] == ob.info["generated"]
            assert ranks[0]["levels"][li]["deadlocks"] == ob.info["deadlocks"]
        if li + 1 < nlev or ranks[0]["depth"] < max_depth:
            ob.step()
    assert all(r["distinct"] == ob.info["distinct"] for r in ranks)
    return ob


def replay_walks_with_oracle(ranks, params):
    """every distributed trace walk is a valid path: from Init, every state of the path is a successor of the one before"""
    from oracle import orc
    P = orc.Params(*params)
    for w in ranks[0]["walks"]:
        rec = orc.init_record(P)
        path = [int(f, 16) for f in w["fps"]]
        assert len(path) == w["level"] and orc.fingerprint(P, rec)[0] == path[0]
        for f in path[1:]:
            nxt = [s["words"] for s in orc.successors(P, rec) if s["fp"] == f]
            assert nxt, "a state of the walk is not a successor of its predecessor"
            rec = nxt[0]
        assert w["fp"] == w["fps"][-1] and w["fp"] in ranks[w["rank"]]["levels"][-1]["fps"]


@pytest.mark.parametrize("world,params,max_depth,rb", [(2, (2, 1, 2, 2), 40, 0), (3, (2, 1, 1, 1), 40, 0), (2, (3, 1, 2, 2), 7, 0),
                                                       (2, (2, 1, 2, 2), 40, 30), (3, (2, 1, 2, 2), 40, 10 ** 9),
                                                       (4, (2, 1, 2, 2), 40, 20)])
def test_sharded_level_loop_matches_single_process_oracle(tmp_path, world, params, max_depth, rb):
    """rb = replicate_below: 0 = sharded from Init on, 30 = the ranks explore the first levels on their own and partition the
    first level with >= 30 new states, 10^9 = never sharded (every rank explores everything)"""
    ranks = run_world("fake", world, params, max_depth, tmp_path, 29640 + world, rb)
    check_against_oracle(ranks, params, max_depth)
    replay_walks_with_oracle(ranks, params)
    assert all(r["walks"] == ranks[0]["walks"] for r in ranks)
    if rb >= 10 ** 9:
        assert all(lv["replicated"] for r in ranks for lv in r["levels"])
    elif rb:
        flags = [lv["replicated"] for lv in ranks[0]["levels"]]
        assert flags[0] and not flags[-1] and flags == sorted(flags, reverse=True)     # one switch, early
    if world > 1 and rb == 0:
        assert sum(r["bytes_sent"] for r in ranks) > 0
        biggest = max(sum(len(r["levels"][li]["fps"]) for r in ranks) for li in range(len(ranks[0]["levels"])))
        if biggest >= 64 * world:                                       # Init lives on one rank: rebalancing must have spread it
            assert sum(r["moved"] for r in ranks) > 0
        last = [len(r["levels"][-2]["fps"]) for r in ranks] if len(ranks[0]["levels"]) > 2 else None
        if last and sum(last) >= 64 * world:
            assert max(last) <= 1.6 * sum(last) / world                 # and the frontier stays balanced


@pytest.mark.parametrize("params,max_depth,rb", [((3, 1, 2, 2), 9, 200), ((3, 1, 3, 3), 8, 100)])
def test_eight_ranks_on_the_cpu_stand_in(tmp_path, params, max_depth, rb):
    """The target world size (BASELINE: 8 x MI355X) rehearsed on the CPU stand-in: prefixes of config 2 and of the README defect configuration on EIGHT
    ranks — the replicated small levels, the partition at the first level with >= rb new states, eight-way owner buckets (owner_of(fp, 8)), the
    all-to-all-v over eight peers, bulk rebalancing, the collective trace walks — per-level fingerprint sets, successor and deadlock counts against the
    single-process oracle.  (Rounds 2-5 never ran the collectives beyond world 4.)"""
    from vsr_tlaplus_amd.sharded import owner_of
    ranks = run_world("fake", 8, params, max_depth, tmp_path, 29740 + params[2], rb)
    check_against_oracle(ranks, params, max_depth)
    replay_walks_with_oracle(ranks, params)
    assert all(r["walks"] == ranks[0]["walks"] for r in ranks) and len(ranks[0]["walks"]) == 8
    flags = [lv["replicated"] for lv in ranks[0]["levels"]]
    assert flags[0] and not flags[-1] and flags == sorted(flags, reverse=True)         # one switch, early
    assert sum(r["bytes_sent"] for r in ranks) > 0 and all(r["bytes_sent"] > 0 for r in ranks)
    last = [len(r["levels"][-1]["fps"]) for r in ranks]
    assert min(last) > 0 and max(last) <= 1.6 * sum(last) / 8                         # every rank holds work, the frontier is balanced
    # the owner function spreads the last level's fingerprints over all eight shards
    owners = [owner_of(int(f, 16), 8) for r in ranks for f in r["levels"][-1]["fps"]]
    assert sorted(set(owners)) == list(range(8))


@pytest.mark.parametrize("world,rb", [(2, 0), (3, 500)])
def test_sharded_checkpoint_and_probe_level(tmp_path, world, rb):
    """(3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority: 146 935 states, violated at depth 19.  The run is checkpointed after
    level 9 (every rank its shard + the loop state), continued from the files by fresh engines, stepped to level 18, and level 19
    is PROBED (nothing stored; violating successors shown to their owners): the oracle's violating fingerprint, a 19-state path."""
    params, depth = (3, 1, 2, 1), 19
    ranks = run_world("fake", world, params, depth - 1, tmp_path, 29660 + world, rb, SHARD_INV_MASK=2, SHARD_CHECKPOINT_AT=9,
                      SHARD_PROBE_AT=depth)
    check_against_oracle(ranks, params, depth - 1)
    check_probe_and_checkpoint(ranks, params, 2, depth)


def test_balance_plan_is_deterministic_and_conservative():
    from vsr_tlaplus_amd.sharded import balance_plan
    assert balance_plan([10, 0]) == []                                  # tiny frontiers are left alone
    assert balance_plan([1000, 1010, 990, 1005]) == []                  # already balanced
    plan = balance_plan([4000, 0, 0, 0])
    moved = {}
    for src, dst, k in plan:
        assert src == 0 and k > 0
        moved[dst] = moved.get(dst, 0) + k
    assert moved == {1: 1000, 2: 1000, 3: 1000}
    counts = [900, 100, 500]
    after = list(counts)
    for src, dst, k in balance_plan(counts):
        after[src] -= k
        after[dst] += k
    assert sum(after) == sum(counts) and max(after) - min(after) <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("engine,world,params,depth,rb", [
    ("hip", 2, (3, 1, 2, 2), 11, 0), ("hip", 3, (3, 1, 3, 3), 9, 0), ("hip", 2, (5, 1, 2, 2), 6, 0),
    ("hip", 2, (3, 1, 2, 2), 12, 200), ("hip", 3, (3, 1, 2, 2), 9, 10 ** 9),
    ("hip-exact", 2, (3, 1, 2, 2), 11, 0), ("hip-exact", 3, (3, 1, 3, 3), 8, 50),
    # the level loop in C++ (csrc/vsr_shard_loop.hpp) over two gloo callbacks, buckets staged through pinned host memory
    ("native", 2, (3, 1, 2, 2), 11, 0), ("native", 3, (3, 1, 3, 3), 9, 0), ("native", 2, (3, 1, 2, 2), 12, 200), ("native", 3, (3, 1, 2, 2), 9, 10 ** 9)])
def test_sharded_hip_engine_on_one_gpu(tmp_path, engine, world, params, depth, rb):
    """All ranks share device 0 and exchange over gloo (staged through the host): every HIP kernel of the sharded protocol
    runs.  "hip" = single-pass levels (k_expand<true> with the sent-filter and speculative writes, k_claim_batch_fused,
    k_apply_verdict), "hip-exact" = two-kernel levels (k_expand<false> bucketing, k_claim_batch, k_verdict, k_materialize);
    rb > 0 adds the replicated phase (vsrmc_shard_local_step, k_partition); k_export / k_append_fixup when ranks drift."""
    ranks = run_world(engine, world, params, depth, tmp_path, 29650 + world, rb)
    check_against_oracle(ranks, params, depth)
    # trace walks: ordinals of the HIP engine replay on the GPU to a state of the right level
    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    m = vt.Model.from_constants(R=params[0], C_=params[1], n=params[2], L=params[3])
    for w in ranks[0]["walks"]:
        tr = sharded.replay_fps(m, [int(f, 16) for f in w["fps"]])
        assert len(tr) == w["level"]
        words = tr[-1][1]
        fps, _ = m.fingerprints(words, np.array([0, len(words)], dtype=np.uint64))
        assert "%016x" % int(fps[0]) in ranks[w["rank"]]["levels"][-1]["fps"]


@pytest.mark.gpu
@pytest.mark.parametrize("world,params,depth,rb,slices", [(2, (3, 1, 2, 2), 12, 0, 3), (3, (3, 1, 3, 3), 9, 50, 2), (2, (5, 1, 2, 2), 6, 0, 5), (4, (3, 1, 2, 2), 11, 100, 4)])
def test_overlapped_exchange_in_slices(tmp_path, world, params, depth, rb, slices):
    """Round 6 (the review's item 2): a sharded level runs in slices — the exchange of slice k (count all-gather, all-to-all of candidates, owners' claims,
    verdict bytes, withdrawal of the losers; second stream, second set of buckets) while k_expand of slice k + 1 runs; the slices append to one next
    frontier and add up in one control block.  Forced onto these small spaces (the default starts at 2^20 states per rank): every level in `slices` slices
    or more — per-level fingerprint SETS over the ranks, new / generated / deadlock counts = the single-process oracle's; the trace walks replay; and the
    sequential level (VSRMC_OVERLAP=0) leaves the same."""
    ranks = run_world("native", world, params, depth, tmp_path, 29750 + world, rb, VSRMC_OVERLAP_MIN_STATES=8, VSRMC_OVERLAP_MIN_SLICES=slices)
    check_against_oracle(ranks, params, depth)
    replay_walks_with_oracle_gpu = [w for w in ranks[0]["walks"]]
    assert all(r["walks"] == ranks[0]["walks"] for r in ranks) and replay_walks_with_oracle_gpu
    lv, sl = ranks[0]["overlap"]
    assert all(r["overlap"] == ranks[0]["overlap"] for r in ranks)                 # every rank cuts every level into the same number of slices
    sharded_levels = sum(1 for x in ranks[0]["levels"][1:] if not x["replicated"])
    assert lv >= sharded_levels - 4 and sl >= slices * lv, (lv, sl, sharded_levels)    # (the first levels hold fewer than 8 states per rank)
    seq = run_world("native", world, params, depth, tmp_path, 29760 + world, rb, VSRMC_OVERLAP=0)
    assert seq[0]["overlap"] == [0, 0]
    for a, b in zip(ranks[0]["levels"], seq[0]["levels"]):
        assert (a["level"], a["n_new"], a["generated"], a["deadlocks"]) == (b["level"], b["n_new"], b["generated"], b["deadlocks"])
    assert sorted(f for r in ranks for f in r["levels"][-1]["fps"]) == sorted(f for r in seq for f in r["levels"][-1]["fps"])


@pytest.mark.gpu
@pytest.mark.parametrize("world,rb", [(2, 0), (3, 500)])
def test_sharded_hip_checkpoint_and_probe_level(tmp_path, world, rb):
    """the same legs over the HIP engine (ranks share device 0, gloo): vsrmc_checker_save / _load of a shard, vsrmc_checker_probe
    on a sharded checker, vsrmc_checker_probe_candidates, vsrmc_checker_seen_batch"""
    params, depth = (3, 1, 2, 1), 19
    ranks = run_world("hip", world, params, depth - 1, tmp_path, 29670 + world, rb, SHARD_INV_MASK=2, SHARD_CHECKPOINT_AT=9,
                      SHARD_PROBE_AT=depth)
    check_against_oracle(ranks, params, depth - 1)
    check_probe_and_checkpoint(ranks, params, 2, depth)


@pytest.mark.gpu
def test_sharded_cli_two_ranks_on_one_gpu(tmp_path):
    """`vsrmc` on N ranks (vsr_tlaplus_amd/sharded_cli.py): config 1 to completion, and the shipped cfg to its violation
    with the counter-example printed in TLC's syntax."""
    from test_host_cpu import _cfg

    def run(cfg, *extra):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29671", "-m", "vsr_tlaplus_amd.sharded_cli", "-config", cfg, "-noTLA", "-backend", "gloo",
               "-tableLog2", "20", "-frontierGiB", "0.05"] + list(extra)
        return subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"))

    # the default: no sizes on the command line — the C++ level loop, every rank sized from (its share of) the free HBM, automatic scheme
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29673",
                        "-m", "vsr_tlaplus_amd.sharded_cli", "-config", _cfg(tmp_path, R=2, vals="v1, v2", L=2), "-noTLA", "-backend", "gloo", "-replicateBelow", "16"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert "Model checking completed. No error has been found." in r.stdout and "2073 distinct states found" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "C++ level loop" in r.stdout and "search is 27" in r.stdout and "[sharded]" in r.stdout
    r = run(_cfg(tmp_path, R=2, vals="v1", L=1), "-replicateBelow", "8")
    assert "Model checking completed. No error has been found." in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "76 distinct states found" in r.stdout and "search is 14" in r.stdout and "[sharded]" in r.stdout
    # the same through `vsrmc -gpus 2` (the C++ front end re-executes itself under torch.distributed.run)
    r = subprocess.run([os.path.join(ROOT, "vsr_tlaplus_amd", "vsrmc"), "-config", _cfg(tmp_path, R=2, vals="v1", L=1), "-noTLA", "-gpus", "2",
                        "-backend", "gloo", "-tableLog2", "20", "-frontierGiB", "0.05", "-replicateBelow", "8"], capture_output=True, text=True,
                       timeout=600, env=dict(os.environ, OMP_NUM_THREADS="1", MASTER_PORT="29672"))
    assert "76 distinct states found" in r.stdout and "[sharded]" in r.stdout and "2 rank(s)" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    # checkpoint after every level to depth 10, then a second run recovers it and probes level 19 of (3,1,{v1,v2},1) with the stricter
    # invariant: violated there (109 878 states up to level 18), 19-state counter-example
    d3 = tmp_path / "c3"
    d3.mkdir()
    cfg3 = _cfg(d3, L=1, extra="AcknowledgedWritesExistOnMajority")
    chk = str(d3 / "chk")
    r = run(cfg3, "-replicateBelow", "100", "-maxDepth", "10", "-checkpoint", chk, "-checkpointMinutes", "0")
    assert "Checkpointing of run %s completed (level 10)." % chk in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert os.path.exists(chk + ".json") and os.path.exists(chk + ".rank1of2")
    r = run(cfg3, "-recover", chk, "-probeAt", "19")
    assert "Recovered from checkpoint" in r.stdout and "Probe(19):" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "Error: Invariant AcknowledgedWritesExistOnMajority is violated." in r.stdout
    assert "State 19: <" in r.stdout and "State 20: <" not in r.stdout and "109878 distinct states found" in r.stdout, r.stdout[-1500:]
    # the AUTOMATIC scheme (C++ level loop) with -checkpoint / -recover, cut AFTER the search has gone beyond the ranks' record buffers (round 6:
    # vsrmc_shard_loop_save / _restore): 1/100 of the device each (2.8 GB: seen-set shards of 2^25 slots, 0.6-GB record buffers), the shipped constants —
    # level 19 no longer fits, "Virtual(..)" lines follow; checkpointed before every unit of progress up to depth 21; recovered to depth 22
    d4 = tmp_path / "c4"
    d4.mkdir()
    chk4 = str(d4 / "chk")
    cfg4 = _cfg(d4)
    def auto(*extra):
        return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29674",
                               "-m", "vsr_tlaplus_amd.sharded_cli", "-config", cfg4, "-noTLA", "-backend", "gloo"] + list(extra),
                              capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1", VSRMC_AUTOSIZE_SHARE="100"))
    r = auto("-maxDepth", "21", "-checkpoint", chk4, "-checkpointMinutes", "0")
    assert "Virtual(20)" in r.stdout and "Checkpointing of run %s completed (depth 20)." % chk4 in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.index("Virtual(") < r.stdout.index("completed (depth 20)")          # the last checkpoint was cut after the search had left its record buffers
    assert os.path.exists(chk4 + ".rank1of2") and os.path.exists(chk4 + ".rank0of2.loop")
    r = auto("-recover", chk4, "-maxDepth", "22")
    assert "Recovered from checkpoint %s: depth 20" % chk4 in r.stdout and "Virtual(21)" in r.stdout and "Virtual(22)" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    with open(os.path.join(ROOT, "tests", "golden", "oracle_levels_config2.json")) as f:
        lv2 = json.load(f)["levels"]
    assert "%d distinct states found" % sum(l["new"] for l in lv2[:22]) in r.stdout, r.stdout[-1500:]
    # the shipped VSR.cfg constants: the run ends in the depth-28 violation of AcknowledgedWriteNotLost (319 M states)
    d2 = tmp_path / "c2"
    d2.mkdir()
    r = run(_cfg(d2), "-tableLog2", "29", "-frontierGiB", "18")
    assert "Error: Invariant AcknowledgedWriteNotLost is violated." in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "State 1: <Initial predicate>" in r.stdout and "State 28: <" in r.stdout and "State 29: <" not in r.stdout
    assert "319228361 distinct states found" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("world,workload", [(2, "readme"), (4, "config2")])
def test_bench_sharded_leg_at_full_scale(world, workload):
    """`bench.py --gpus N` (the driver's scaling run) with N ranks sharing this GPU over gloo — the level loop in C++ (on a multi-GPU
    node it talks RCCL directly; here its two collectives are gloo callbacks and the buckets are staged through the host).  world 2:
    the README defect configuration, the leg's default — every rank sizes itself from (its share of) the free HBM, the last levels live
    in the seen-sets only; world 4: config 2 (319 M states) the same way.  bench asserts every level's figures, the distinct-state
    count, the depth, the violating fingerprint and the trace replay itself."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29680 + world), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline"] + (["--workload", "config2"] if workload == "config2" else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="1", VSR_BENCH_BACKEND="gloo"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == world and out["value"] > 0 and out["roofline"]["launches"] >= 20
    assert out["config"]["level_loop"].startswith("native C++")
    assert ("README" in out["config"]["workload"]) == (workload == "readme")
    if workload == "readme":
        assert len(out["deep_passes"]) >= 2 and out["deep_passes"][-1]["probed_level"] == 24


@pytest.mark.gpu
def test_reset_clears_the_sent_filter(tmp_path):
    """the sent-filter of a sharded single-pass checker (announced (fingerprint, auxkey) tags, any level) must not survive
    vsrmc_checker_reset: a second run announces exactly what the first one did.  (Own process: torch has to be loaded before
    libvsrmc.so — both bind libamdhip64.so.7 — and this pytest process has used the library already.)"""
    script = tmp_path / "sent_filter.py"
    script.write_text("""
import sys
sys.path.insert(0, %r)
import numpy as np
from vsr_tlaplus_amd import sharded
import vsr_tlaplus_amd as vt
m = vt.Model.from_constants(R=3, C_=1, n=2, L=2)
eng = sharded.HipShardEngine(m, 0, 2, device=0, table_log2=20, frontier_words=1 << 22, frontier_states=1 << 17,
                             pending_entries=1 << 19, cand_cap=1 << 18, rec_cap=1 << 17, rec_words_cap=1 << 22, filter_log2=16)
runs = []
for _ in range(2):
    for _ in range(7):
        eng.local_step()                                # replicated phase: levels 2-8 on this rank alone
    eng.partition()
    cands, err = eng.expand()                           # rank 0 of 2: what it would announce to rank 1 for level 9
    assert err == 0
    # 0 = unused entry of a block's chunk; the filter is a lossy cache, so a tag can be announced twice: compare the SETS
    runs.append(sorted(set(int(x) for x in cands[1][:, 0].cpu().numpy().view(np.uint64) if x)))
    eng.reset()
assert len(runs[0]) > 100 and runs[0] == runs[1], (len(runs[0]), len(runs[1]))
eng.close()
print("OK", len(runs[0]))
""" % ROOT)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


# ---------------------------------------------------------------------------------------------------------------------
# the automatic level scheme on a sharded run: levels beyond the ranks' record buffers (vsrmc_shard_loop_advance / _deepen)
# ---------------------------------------------------------------------------------------------------------------------
def run_deep_world(world, params, inv_mask, max_depth, tmp_path, port, fw_log2=0, replicate_below=0, **env):
    out = str(tmp_path / ("deep_w%d" % world))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "shard_deep_worker.py")] + [str(x) for x in params] + \
          [str(inv_mask), str(max_depth), out, str(fw_log2), str(replicate_below)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=dict(os.environ, OMP_NUM_THREADS="1", **{k: str(v) for k, v in env.items()}))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ranks = [json.load(open("%s.rank%d.json" % (out, k))) for k in range(world)]
    # seconds / launches are the rank's own; act_generated[0] is not a count (a shader clock rides in that slot)
    strip = lambda lv: {k: (v[1:] if k == "act_generated" else v) for k, v in lv.items() if k not in ("seconds", "launches")}   # noqa: E731
    for r_ in ranks[1:]:                                                 # every figure is the level's, the same on every rank
        assert ([strip(x) for x in r_["levels"]], r_["probed"], r_["violation"], r_["path"], r_["distinct"]) == \
            ([strip(x) for x in ranks[0]["levels"]], ranks[0]["probed"], ranks[0]["violation"], ranks[0]["path"], ranks[0]["distinct"])
    ranks[0]["stopped"] = [r_.get("stopped") for r_ in ranks]            # (why each rank's loop ended early, if it did)
    return ranks[0]


@pytest.mark.gpu
@pytest.mark.parametrize("world,rb", [(2, 0), (3, 300)])
def test_sharded_deep_levels_against_the_oracle(tmp_path, world, rb):
    """(3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority (violated at depth 19, 109 878 states) on 2 / 3 ranks with record buffers
    of 2^17 words: a few sharded levels are stored, the rest live in the ranks' seen-sets only — virtual, regenerated (every rank rebuilds the
    states ITS candidates inserted — its winner set — once per descent, asking nobody: round 5), streamed and probed levels, each
    held against the oracle: new states, successors in total and per action, deadlocks, largest bag, and for the levels that are never
    stored the xor / sum of their fingerprints.  The violation is found by a probe pass; its counter-example replays in the oracle."""
    from oracle import orc
    params, inv = (3, 1, 2, 1), 2
    got = run_deep_world(world, params, inv, 19, tmp_path, 29690 + world, fw_log2=17, replicate_below=rb)
    P = orc.Params(*params, invariant_mask=inv)
    ob = orc.Bfs(P)
    kinds = [lv["kind"] for lv in got["levels"]]
    assert "level" in kinds and kinds.count("deep") >= 5 and kinds == sorted(kinds, key=lambda k: k == "deep"), kinds
    for lv in got["levels"]:
        n = ob.step()
        assert lv["level"] == ob.info["depth"]
        assert (lv["n_new"], lv["generated"], lv["deadlocks"]) == (n, ob.info["generated"], ob.info["deadlocks"]), lv["level"]
        if lv["kind"] == "deep":
            fps = ob.level_fps(lv["level"])
            assert lv["fp_xor"] == "%016x" % int(np.bitwise_xor.reduce(fps)), lv["level"]
            assert lv["fp_sum"] == "%016x" % (int(fps.astype(object).sum()) & ((1 << 64) - 1)), lv["level"]
    assert got["levels"][-1]["level"] == 18 and got["distinct"] == ob.info["distinct"] == 109878
    ob.step()
    words, off = ob.frontier()
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    assert got["violation"] == dict(level=19, fp="%016x" % viol, mask=2, probed=True)
    assert got["probed"]["generated"] == ob.info["generated"] and got["probed"]["deadlocks"] == ob.info["deadlocks"]
    path = [int(f, 16) for f in got["path"]]
    rec = orc.init_record(P)
    assert len(path) == 19 and orc.fingerprint(P, rec)[0] == path[0]
    for f in path[1:]:
        nxt = [s for s in orc.successors(P, rec) if s["fp"] == f]
        assert nxt, "a state of the counter-example is not a successor of its predecessor"
        rec, bad = nxt[0]["words"], nxt[0]["inv"]
    assert bad == 2


@pytest.mark.gpu
@pytest.mark.parametrize("world,params,inv,fw,save_at,last", [(2, (3, 1, 2, 1), 2, 17, 13, 19), (2, (3, 1, 3, 3), 1, 18, 10, 12), (3, (3, 1, 2, 1), 2, 17, 15, 19)])
def test_sharded_deep_search_is_checkpointed_and_recovered(tmp_path, world, params, inv, fw, save_at, last):
    """Round 6 (the round-5 review's f4 remainder): a SHARDED search that has gone beyond its record buffers is checkpointed between two passes
    (vsrmc_shard_loop_save: every rank its seen-set shard, its part of the stored base level, the descriptors of the seen-set-only levels, its winner
    set; the loop's totals in a sidecar; two phases) and recovered by NEW processes (vsrmc_checker_load + vsrmc_shard_loop_restore), which make every
    later pass as the uninterrupted run does: every level's figures and checksums against the CPU oracle before and after the cut, the violation and
    its counter-example where the configuration has one.  (3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority on 2 / 3 ranks; the README defect
    configuration (six permutations) on 2 ranks with 2 MB record buffers, cut at depth 10."""
    from oracle import orc
    first = run_deep_world(world, params, inv, last, tmp_path, 29705 + world, fw_log2=fw, SHARD_SAVE_AT=save_at)
    assert first["violation"] is None and first["depth"] >= save_at and [lv["kind"] for lv in first["levels"]][-1] == "deep"
    prefix = str(tmp_path / ("deep_w%d" % world)) + ".chk"
    for r in range(world):
        assert os.path.exists("%s.rank%dof%d" % (prefix, r, world)) and os.path.exists("%s.rank%dof%d.loop" % (prefix, r, world))
    second = run_deep_world(world, params, inv, last, tmp_path, 29715 + world, fw_log2=fw, SHARD_RECOVER=prefix)
    levels = first["levels"] + second["levels"]
    assert [lv["level"] for lv in levels] == list(range(2, levels[-1]["level"] + 1)) and second["levels"], "the recovered run goes on where the first one stopped"
    assert all(lv["kind"] == "deep" for lv in second["levels"])
    P = orc.Params(*params, invariant_mask=inv)
    ob = orc.Bfs(P)
    for lv in levels:
        n = ob.step()
        assert (lv["level"], lv["n_new"], lv["generated"], lv["deadlocks"]) == (ob.info["depth"], n, ob.info["generated"], ob.info["deadlocks"]), lv["level"]
        if lv["kind"] == "deep":
            fps = ob.level_fps(lv["level"])
            assert lv["fp_xor"] == "%016x" % int(np.bitwise_xor.reduce(fps)), lv["level"]
            assert lv["fp_sum"] == "%016x" % (int(fps.astype(object).sum()) & ((1 << 64) - 1)), lv["level"]
    assert second["distinct"] == ob.info["distinct"]
    if inv == 2:                                                          # the violation at depth 19 is found by the recovered run's probe
        ob.step()
        words, off = ob.frontier()
        viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
                   if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
        assert second["violation"] == dict(level=19, fp="%016x" % viol, mask=2, probed=True)
        path = [int(f, 16) for f in second["path"]]
        rec = orc.init_record(P)
        assert len(path) == 19 and orc.fingerprint(P, rec)[0] == path[0]
        for f in path[1:]:
            nxt = [s_ for s_ in orc.successors(P, rec) if s_["fp"] == f]
            assert nxt, "a state of the counter-example is not a successor of its predecessor"
            rec = nxt[0]["words"]
    else:
        assert second["violation"] is None and second["depth"] == last


@pytest.mark.gpu
def test_readme_configuration_on_two_ranks(tmp_path, oracle_levels):
    """BASELINE configs[2] (the reference README's defect configuration) through the sharded path at world 2, both ranks on this GPU with
    half of its memory each (sizes from the free HBM, no level numbers): all 23 levels of the CPU oracle's fixture — the stored ones and
    the ones that exist in the two seen-sets only — and the probe's violator 042ca5372e82d6fb at depth 24, with a 24-state path."""
    g = oracle_levels["config3"]
    p = g["params"]
    got = run_deep_world(2, (p["R"], p["C"], p["n"], p["L"]), p["inv_mask"], 23, tmp_path, 29697)
    assert len(got["levels"]) == 22 and "deep" in [lv["kind"] for lv in got["levels"]], (got.get("stopped"), got["sizes"])
    for lv, want in zip(got["levels"], g["levels"][1:]):
        assert (lv["level"], lv["n_new"], lv["generated"], lv["deadlocks"], lv["max_bag"]) == \
            (want["level"], want["new"], want["generated"], want["deadlocks"], want["max_bag"]), want["level"]
        assert lv["act_generated"][1:16] == want["act_generated"][1:16], want["level"]
        if lv["kind"] == "deep" and g["checksums"]:
            assert (lv["fp_xor"], lv["fp_sum"]) == (want["fp_xor"], want["fp_sum"]), want["level"]
    assert got["distinct"] == g["distinct"]
    pr = g["probe"]
    assert (got["probed"]["level"], got["probed"]["generated"], got["probed"]["deadlocks"], got["probed"]["viol_mask"]) == \
        (pr["level"], pr["generated"], pr["deadlocks"], pr["viol_mask"])
    assert got["violation"]["probed"] and got["violation"]["level"] == 24 and len(got["path"]) == 24
    if g["checksums"]:
        assert got["violation"]["fp"] == pr["viol_fp"] == "042ca5372e82d6fb"


@pytest.mark.gpu
def test_config4_on_two_ranks(tmp_path, oracle_levels):
    """BASELINE configs[3] (3,2,{v1,v2,v3},3: two clients) under the documented policy for VSR.tla:421 (`assume_commit_number`; strict TLC
    semantics abort at that line) through the sharded path at world 2, both ranks on this GPU: the first 14 levels of the CPU oracle's
    fixture (tests/golden/oracle_levels_config4.json), every figure per level, on a configuration whose second client doubles the
    ReceiveClientRequest bindings and exercises the client-table branch of ReceivePrepareMsg (VSR.tla:414-421) on both ranks."""
    g = oracle_levels["config4"]
    p = g["params"]
    assert p.get("assume_commit_number")
    got = run_deep_world(2, (p["R"], p["C"], p["n"], p["L"]), p["inv_mask"], 14, tmp_path, 29698, SHARD_ASSUME_COMMIT=1)
    assert [lv["level"] for lv in got["levels"]] == list(range(2, 15)) and got["violation"] is None
    for lv, want in zip(got["levels"], g["levels"][1:]):
        assert (lv["level"], lv["n_new"], lv["generated"], lv["deadlocks"], lv["max_bag"]) == \
            (want["level"], want["new"], want["generated"], want["deadlocks"], want["max_bag"]), want["level"]
        assert lv["act_generated"][1:16] == want["act_generated"][1:16], want["level"]
    assert got["distinct"] == sum(lv["new"] for lv in g["levels"][:14])


@pytest.mark.gpu
@pytest.mark.parametrize("engine", ["hip", "native", "hip-exact"])
def test_violation_of_any_mask_on_a_remotely_owned_successor_is_reported(tmp_path, engine):
    """Round-2 advice (high): a successor that is written speculatively by its generator and owned by another rank carries its
    violated-invariant mask beside its state index (cand_pack) — two bits of it in round 2, which lost the masks 4 / 8 / 16 of the
    analysis models.  No shipped cfg reaches such a violation, so the test hook VSRMC_TEST_FORCE_BAD=<fp>:<mask> makes one: a state of
    level 8 that rank 0 generates and rank 1 owns fails the invariants 4 | 8 | 16.  The run must report exactly that state, at that
    level, with the whole mask — through the Python loop, the C++ loop and the two-kernel (exact) scheme."""
    params, depth = (3, 1, 2, 2), 9
    ranks = run_world(engine, 2, params, depth, tmp_path, 29700, 0)
    assert all(r["violation"] is None for r in ranks)
    lvl8 = ranks[0]["levels"][7]
    assert lvl8["level"] == 8 and not lvl8["replicated"]
    owner = lambda fp: ((fp >> 40) & 0xFFFFFF) % 2                       # noqa: E731  (sharded.owner_of)
    remote = [f for f in lvl8["fps"] if owner(int(f, 16)) == 1]            # generated (and stored) by rank 0, owned by rank 1
    assert remote
    target = remote[len(remote) // 2]
    # the hook lives in libvsrmc_hooks.so only (-DVSRMC_TEST_HOOKS, built beside the product library by vsr_tlaplus_amd/build.py); the product
    # library ignores the variable: the same run through it reports nothing
    hooks = os.path.join(ROOT, "vsr_tlaplus_amd", "libvsrmc_hooks.so")
    assert os.path.exists(hooks), "build it: python vsr_tlaplus_amd/build.py"
    if engine == "hip":
        ranks = run_world(engine, 2, params, depth, tmp_path, 29702, 0, VSRMC_TEST_FORCE_BAD="%s:28" % target)
        assert all(r["violation"] is None for r in ranks)
    ranks = run_world(engine, 2, params, depth, tmp_path, 29701, 0, VSRMC_TEST_FORCE_BAD="%s:28" % target, VSRMC_LIB=hooks)
    for r in ranks:
        assert r["violation"] == dict(level=8, fp=target, mask=28), r["violation"]


@pytest.mark.parametrize("world,rb,deep_at", [(2, 0, 8), (3, 200, 12), (8, 200, 13)])
def test_sharded_deep_protocol_on_the_cpu_stand_in(tmp_path, world, rb, deep_at):
    """The protocol of the levels beyond the ranks' record buffers (virtual level: announce -> first inserter wins -> winners counted;
    regenerated level: local — a rank rebuilds the states of that level its own candidates inserted (winner set: fingerprint -> level, last
    descent), exactly once per descent, no exchange; inserted level + probe with the candidates shown to their owners) as the Python loop runs it over the CPU stand-in
    engine — the reference implementation of what csrc/vsr_shard_loop.hpp does over the HIP engine (`-m gpu`:
    test_sharded_deep_levels_against_the_oracle).  (3,1,{v1,v2},1) with AcknowledgedWritesExistOnMajority: stored sharded levels up to
    `deep_at` - 1, then through the seen-sets alone with slices of 48 states (every loop runs many times, the ranks run out of work at
    different moments) to the violation at depth 19 — every level's size, successors, deadlocks and fingerprint checksums against the
    single-process oracle; a 19-state path that replays."""
    from oracle import orc
    params, inv, depth = (3, 1, 2, 1), 2, 19
    ranks = run_world("fake", world, params, depth, tmp_path, 29720 + world, rb, SHARD_INV_MASK=inv, SHARD_DEEP_AT=deep_at)
    P = orc.Params(*params, invariant_mask=inv)
    ob = orc.Bfs(P)
    for _ in range(deep_at - 2):
        ob.step()
    assert all(r["deep"] == ranks[0]["deep"] and r["violation"] == ranks[0]["violation"] and r["path"] == ranks[0]["path"] for r in ranks)
    got = ranks[0]
    assert len(got["deep"]) == 18 - (deep_at - 1)
    for lv in got["deep"]:
        n = ob.step()
        fps = ob.level_fps(lv["level"])
        assert (lv["level"], lv["n_new"], lv["generated"], lv["deadlocks"]) == (ob.info["depth"], n, ob.info["generated"], ob.info["deadlocks"]), lv["level"]
        assert lv["fp_xor"] == "%016x" % int(np.bitwise_xor.reduce(fps)) and lv["fp_sum"] == "%016x" % (int(fps.astype(object).sum()) & ((1 << 64) - 1))
    assert got["distinct"] == ob.info["distinct"] == 109878
    ob.step()
    words, off = ob.frontier()
    viol = min(orc.fingerprint(P, words[int(off[i]): int(off[i + 1])])[0] for i in range(len(off) - 1)
               if orc.invariants(P, words[int(off[i]): int(off[i + 1])]))
    assert got["violation"] == dict(level=19, fp="%016x" % viol, mask=2, probed=True)
    last = got["deep"][-1]["probed"]
    assert (last["level"], last["generated"], last["deadlocks"]) == (19, ob.info["generated"], ob.info["deadlocks"])
    path = [int(f, 16) for f in got["path"]]
    rec = orc.init_record(P)
    assert len(path) == 19 and orc.fingerprint(P, rec)[0] == path[0]
    for f in path[1:]:
        nxt = [s for s in orc.successors(P, rec) if s["fp"] == f]
        assert nxt, "a state of the counter-example is not a successor of its predecessor"
        rec, bad = nxt[0]["words"], nxt[0]["inv"]
    assert bad == 2


def test_sharded_deep_protocol_readme_configuration_prefix_on_the_cpu_stand_in(tmp_path):
    """the README defect configuration (3,1,{v1,v2,v3},3: six value permutations) at world 2: levels 2-5 stored, 6-9 through the seen-sets
    alone, the probe of level 10 clean — each against the single-process oracle"""
    from oracle import orc
    params = (3, 1, 3, 3)
    ranks = run_world("fake", 2, params, 9, tmp_path, 29731, 0, SHARD_DEEP_AT=6, SHARD_DEEP_TO=9, SHARD_DEEP_SLICE=64)
    ob = orc.Bfs(orc.Params(*params))
    for _ in range(4):
        ob.step()
    got = ranks[0]
    assert ranks[1]["deep"] == got["deep"] and [lv["level"] for lv in got["deep"]] == [6, 7, 8, 9] and got["violation"] is None
    for lv in got["deep"]:
        n = ob.step()
        fps = ob.level_fps(lv["level"])
        assert (lv["n_new"], lv["generated"], lv["deadlocks"]) == (n, ob.info["generated"], ob.info["deadlocks"]), lv["level"]
        assert lv["fp_xor"] == "%016x" % int(np.bitwise_xor.reduce(fps))
    ob.step()
    assert got["deep"][-1]["probed"] == dict(level=10, generated=ob.info["generated"], deadlocks=ob.info["deadlocks"], viol_fp=None, viol_mask=0)
    assert got["distinct"] == ob.info["distinct"] - ob.info["n_new"] if "n_new" in ob.info else True

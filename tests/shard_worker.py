"""TEST INFRASTRUCTURE: one rank of a world_size-N sharded BFS, launched by tests via torch.distributed.run.
    python -m torch.distributed.run ... tests/shard_worker.py <engine: fake|hip|hip-exact> R C n L max_depth out_prefix [replicate_below]
Writes out_prefix.rank<k>.json with the per-level sorted fingerprints of this rank's shard and the global counters."""
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    engine_kind, R, C_, n, L, max_depth, out = sys.argv[1], *(int(x) for x in sys.argv[2:7]), sys.argv[7]
    replicate_below = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    dist.init_process_group("gloo")           # CPU test: gloo; on the one-GPU box both ranks share device 0 over gloo
    rank, world = dist.get_rank(), dist.get_world_size()
    from vsr_tlaplus_amd import sharded
    # optional legs (environment): SHARD_INV_MASK = the invariants checked; SHARD_CHECKPOINT_AT = k: after level k the run is
    # checkpointed, every engine is thrown away and the run continues from the files; SHARD_PROBE_AT = k: level k is probed
    # (ShardedChecker.probe) instead of stepped into
    inv_mask = int(os.environ.get("SHARD_INV_MASK", "1"))
    chk_at, probe_at = int(os.environ.get("SHARD_CHECKPOINT_AT", "0")), int(os.environ.get("SHARD_PROBE_AT", "0"))
    if engine_kind == "fake":
        from fake_shard_engine import FakeShardEngine
        from oracle import orc
        P = orc.Params(R, C_, n, L, invariant_mask=inv_mask)

        def make_engine(recover=None):
            return FakeShardEngine.load(recover, P, rank, world, sharded.owner_of) if recover else FakeShardEngine(P, rank, world, sharded.owner_of)
    else:
        import vsr_tlaplus_amd as vt
        m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L, invariant_mask=inv_mask)

        def make_engine(recover=None):
            return sharded.HipShardEngine(m, rank, world, device=0, table_log2=20, frontier_words=1 << 22, frontier_states=1 << 17,
                                          pending_entries=1 << 19, cand_cap=1 << 18, rec_cap=1 << 17, rec_words_cap=1 << 22,
                                          exact_ties=engine_kind == "hip-exact", filter_log2=16, recover=recover)   # "native": the same engine under the C++ loop
    eng = make_engine()
    if engine_kind == "native":                                # the C++ level loop (csrc/vsr_shard_loop.hpp) over gloo callbacks
        sc = sharded.NativeShardedChecker(eng, sharded.TorchHostComm(), replicate_below=replicate_below)
        sc.x = sharded.Exchanger()
    else:
        sc = sharded.ShardedChecker(eng, sharded.Exchanger(), replicate_below=replicate_below)
    # "replicated": the level's states are on every rank (the ranks explored it on their own), else each state is on one rank
    levels = [dict(level=1, n_new=sc.distinct, generated=0, deadlocks=0, replicated=sc.replicated,
                   fps=["%016x" % int(f) for f in eng.level_fps()])]
    restored = False
    while sc.level < max_depth and not (int(os.environ.get("SHARD_DEEP_AT", "0")) and sc.level == int(os.environ.get("SHARD_DEEP_AT", "0")) - 1):
        if chk_at and sc.level == chk_at and not restored:
            sc.save(out + ".chk")
            if hasattr(eng, "close"):
                eng.close()
            x = sc.x
            del sc, eng
            sc = sharded.ShardedChecker.restore(out + ".chk", make_engine, x)
            eng = sc.e
            restored = True
        d = sc.step()
        if d["n_new"] == 0:
            break
        levels.append(dict(level=d["level"], n_new=d["n_new"], generated=d["generated"], deadlocks=d["deadlocks"],
                           replicated=sc.replicated, fps=["%016x" % int(f) for f in eng.level_fps()]))
    deep = []
    deep_at = int(os.environ.get("SHARD_DEEP_AT", "0"))          # the levels from here on live in the ranks' seen-sets only (ShardedChecker.deepen)
    if deep_at and sc.level == deep_at - 1:
        depth_now = sc.level
        while depth_now < int(os.environ.get("SHARD_DEEP_TO", "99")) and sc.violation is None:
            a, b = sc.deepen(slice_size=int(os.environ.get("SHARD_DEEP_SLICE", "48")))
            if a["n_new"] == 0:
                break
            depth_now = a["level"]
            deep.append(dict(level=a["level"], n_new=a["n_new"], generated=a["generated"], deadlocks=a["deadlocks"], fp_xor="%016x" % a["fp_xor"],
                             fp_sum="%016x" % a["fp_sum"], probed=None if b is None else dict(level=b["level"], generated=b["generated"], deadlocks=b["deadlocks"],
                                                                                              viol_fp=None if b["viol_fp"] is None else "%016x" % b["viol_fp"], viol_mask=b["viol_mask"])))
        path = []
        if sc.violation is not None:
            path = ["%016x" % f for f in (sc.probe_trace_fps() if sc.violation.get("probed") else sc.trace_fps(sc.violation["level"], sc.violation["fp"]))]
        viol = None if sc.violation is None else dict(level=sc.violation["level"], fp="%016x" % sc.violation["fp"], mask=sc.violation["mask"], probed=bool(sc.violation.get("probed")))
        with open("%s.rank%d.json" % (out, rank), "w") as f:
            json.dump(dict(rank=rank, world=world, distinct=sc.distinct, depth=depth_now, levels=levels, deep=deep, violation=viol, path=path), f)
        dist.barrier()
        dist.destroy_process_group()
        return
    probe = None
    if probe_at and sc.level == probe_at - 1:
        probe = sc.probe()
        probe["viol_fp"] = "%016x" % probe["viol_fp"] if probe["viol_fp"] is not None else None
        probe["fps"] = ["%016x" % f for f in sc.probe_trace_fps()] if probe["viol_fp"] else []
    # trace of the last state of the deepest non-empty local level (every rank takes part in every walk)
    walks = []
    for r in range(world):
        # rank r nominates its state with the largest fingerprint; states are addressed by fingerprint
        mine = eng.level_fps()
        fp = int(mine[-1]) if (rank == r and len(mine)) else 0
        hi, lo = sc.x.allreduce([fp >> 32, fp & 0xFFFFFFFF], dist.ReduceOp.MAX) if world > 1 else (fp >> 32, fp & 0xFFFFFFFF)
        # (the two halves come from the same rank: every other rank contributes zeros)
        fp = (hi << 32) | lo
        if fp:
            walks.append(dict(rank=r, level=sc.level, fp="%016x" % fp, fps=["%016x" % f for f in sc.trace_fps(sc.level, fp)]))
    with open("%s.rank%d.json" % (out, rank), "w") as f:
        viol = None
        if sc.violation is not None:
            viol = dict(level=sc.violation["level"], fp="%016x" % sc.violation["fp"], mask=sc.violation["mask"])
        json.dump(dict(rank=rank, world=world, distinct=sc.distinct, depth=sc.level, levels=levels, walks=walks, violation=viol,
                       bytes_sent=sc.x.bytes_sent, moved=sc.moved, probe=probe, restored=restored,
                       overlap=list(sc.overlap_stats()) if hasattr(sc, "overlap_stats") else None), f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

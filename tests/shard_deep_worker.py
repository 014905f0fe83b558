"""TEST INFRASTRUCTURE: one rank of a sharded run through the AUTOMATIC level scheme of the C++ level loop (vsrmc_shard_loop_advance):
ordinary sharded levels while they fit, then levels that live in the ranks' seen-sets only (csrc/vsr_deep.hpp with the collective hooks).
    python -m torch.distributed.run ... tests/shard_deep_worker.py R C n L inv_mask max_depth out_prefix [frontier_words_log2 | 0 = auto]
All ranks share device 0 and talk over gloo (buckets staged through pinned host memory).  Writes out_prefix.rank<k>.json: the figures of
every level over all ranks, which kind of step produced it, the violation and the fingerprints of its counter-example."""
import json
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    R, C_, n, L, inv_mask, max_depth = (int(x) for x in sys.argv[1:7])
    out = sys.argv[7]
    fw_log2 = int(sys.argv[8]) if len(sys.argv) > 8 else 0
    replicate_below = int(sys.argv[9]) if len(sys.argv) > 9 else 0
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    os.environ.setdefault("VSRMC_AUTOSIZE_SHARE", str(world))   # the ranks share one device here
    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    m = vt.Model.from_constants(R=R, C_=C_, n=n, L=L, invariant_mask=inv_mask, assume_commit_number=bool(int(os.environ.get("SHARD_ASSUME_COMMIT", "0"))))
    def make_engine(recover=None):
        if fw_log2:
            return sharded.HipShardEngine(m, rank, world, device=0, table_log2=20, frontier_words=1 << fw_log2, frontier_states=1 << (fw_log2 - 5),
                                          pending_entries=1 << 16, cand_cap=1 << 17, rec_cap=1 << 15, rec_words_cap=1 << 20, filter_log2=16, native_only=True,
                                          recover=recover)
        return sharded.HipShardEngine(m, rank, world, device=0, table_log2=0, frontier_words=0, frontier_states=0, pending_entries=0, cand_cap=0,
                                      rec_cap=1 << 22, rec_words_cap=1 << 28, native_only=True, recover=recover)
    # SHARD_SAVE_AT=<depth>: when the search has reached that depth it is checkpointed (vsrmc_shard_loop_save) and this process ends;
    # SHARD_RECOVER=<prefix>: the run continues from such a checkpoint, in fresh processes
    save_at, recover = int(os.environ.get("SHARD_SAVE_AT", "0")), os.environ.get("SHARD_RECOVER")
    if recover:
        sc = sharded.NativeShardedChecker.restore(recover, make_engine, sharded.TorchHostComm())
        eng = sc.e
    else:
        eng = make_engine()
        sc = sharded.NativeShardedChecker(eng, sharded.TorchHostComm(), replicate_below=replicate_below)
        sc.depth = sc.level
    rows = []
    probed = None
    stopped = None
    while sc.depth < max_depth and sc.violation is None:
        if save_at and sc.depth >= save_at:
            sc.save(out + ".chk")
            break
        if sc.room() == 2:                                      # (collective, as every loop over advance() asks: the seen-set shards — and it grows the winner sets)
            stopped = "no room: " + getattr(sc, "room_note", "")
            break
        kind, a, b = sc.advance()
        if os.environ.get("SHARD_VERBOSE") and rank == 0:
            print("level %d %s: new %d generated %d launches %s seconds %.3f overlap %s" % (a["level"], kind, a["n_new"], a["generated"], a["launches"], a["seconds"],
                                                                                   sc.overlap_stats()), flush=True)
        if a["n_new"] == 0:
            break
        rows.append(dict(kind=kind, level=a["level"], n_new=a["n_new"], generated=a["generated"], deadlocks=a["deadlocks"], max_bag=a["max_bag"],
                         fp_xor="%016x" % a["fp_xor"], fp_sum="%016x" % a["fp_sum"], act_generated=[int(x) for x in a["act_generated"]],
                         launches=a["launches"], seconds=a["seconds"]))
        if b is not None:
            probed = dict(level=b["level"], generated=b["generated"], deadlocks=b["deadlocks"], viol_mask=b["viol_mask"],
                          viol_fp=("%016x" % b["viol_fp"]) if b["viol_mask"] else None)
    viol, path = None, []
    if sc.violation is not None:
        viol = dict(level=sc.violation["level"], fp="%016x" % sc.violation["fp"], mask=sc.violation["mask"], probed=bool(sc.violation.get("probed")))
        path = ["%016x" % f for f in sc.violation_trace_fps()]
    with open("%s.rank%d.json" % (out, rank), "w") as f:
        json.dump(dict(rank=rank, world=world, distinct=sc.distinct, depth=sc.depth, levels=rows, probed=probed, violation=viol, path=path, stopped=stopped,
                       bytes_sent=sc.bytes_sent, sizes=dict(table_log2=int(eng.options.table_log2), frontier_words=int(eng.options.frontier_words),
                                                            cand_cap=int(eng.cand_cap))), f)
    sc.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

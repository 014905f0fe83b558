"""vsrmc on N GPUs — the TLC-style command line of the sharded checker (`vsrmc ... --gpus N`).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m vsr_tlaplus_amd.sharded_cli -config VSR.cfg [VSR.tla] [options]

One rank per GPU (backend "nccl" = RCCL over xGMI; "gloo" stages through the host and lets several ranks share one GPU).
Rank 0 prints TLC's progress lines, the violated invariant and the counter-example in TLC's value syntax; every rank returns
TLC's exit code (0 = no error, 12 = safety violation, 11 = deadlock, 1 = failure).

  -config FILE        TLC configuration (VSR.cfg grammar)              -noTLA   do not read / hash-check the .tla file
  -maxDepth N         stop after N BFS levels (Init = level 1)         -checkDeadlock   stop at the first terminal state
  -tableLog2 N        seen-set slots PER RANK = 2^N x 16 B; -frontierGiB G: size of each of the two record buffers PER RANK.
                      DEFAULT (neither given, no -checkpoint / -recover / -exactTies / -probeAt): the AUTOMATIC scheme over the C++ level
                      loop — every rank sizes its seen-set shard, record and exchange buffers from the free memory of its GPU, levels are
                      stored while every rank's part of the next one fits, then the search goes on through the seen-sets alone
                      (Virtual(L) / Probe(L+1) lines).  With explicit sizes: the Python loop, stored levels only (2^26 slots, 2 GiB)
  -replicateBelow K   levels with fewer than K new states are explored by every rank on its own (default 2^20; 0 = never)
  -exactTies          two-kernel levels that arbitrate same-level VIEW ties like the oracle (default: single-pass levels)
  -backend nccl|gloo  (default nccl)                                   -json    one JSON object per level
  -checkpoint PREFIX  checkpoint between levels, at most every -checkpointMinutes M (default 30; 0 = after every level): every
                      rank writes its shard to PREFIX.rank<r>of<N>, rank 0 the loop state to PREFIX.json
  -recover PREFIX     continue the run a checkpoint stopped at (same constants, same number of ranks)
  -probeAt N          level N is PROBED instead of explored: its states' invariants are checked, nothing is stored (one level beyond
                      what the ranks' buffers hold); the search ends there
"""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

INVARIANTS = ["AcknowledgedWriteNotLost", "AcknowledgedWritesExistOnMajority", "NoLogDivergence", "CommitNumberNeverHigherThanOpNumber",
              "NoAppStateDivergence"]            # invariant_mask bits (include/vsrmc.h)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    opt = dict(cfg=None, tla=None, max_depth=1 << 30, table_log2=0, frontier_gib=0.0, replicate_below=1 << 20, exact=False,
               backend="nccl", json=False, no_tla=False, check_deadlock=False, checkpoint=None, checkpoint_minutes=30.0, recover=None,
               probe_at=0)
    i = 0
    while i < len(argv):
        a = argv[i]
        val = argv[i + 1] if i + 1 < len(argv) else None
        if a == "-config" and val:
            opt["cfg"] = val; i += 1
        elif a == "-maxDepth" and val:
            opt["max_depth"] = int(val); i += 1
        elif a == "-tableLog2" and val:
            opt["table_log2"] = int(val); i += 1
        elif a == "-frontierGiB" and val:
            opt["frontier_gib"] = float(val); i += 1
        elif a == "-replicateBelow" and val:
            opt["replicate_below"] = int(val); i += 1
        elif a == "-backend" and val:
            opt["backend"] = val; i += 1
        elif a == "-checkpoint" and val:
            opt["checkpoint"] = val; i += 1
        elif a == "-checkpointMinutes" and val:
            opt["checkpoint_minutes"] = float(val); i += 1
        elif a == "-recover" and val:
            opt["recover"] = val; i += 1
        elif a == "-probeAt" and val:
            opt["probe_at"] = int(val); i += 1
        elif a == "-exactTies":
            opt["exact"] = True
        elif a == "-json":
            opt["json"] = True
        elif a == "-noTLA":
            opt["no_tla"] = True
        elif a == "-deadlock":
            opt["check_deadlock"] = False
        elif a == "-checkDeadlock":
            opt["check_deadlock"] = True
        elif a == "-workers" and val:
            i += 1                                              # accepted for command-line compatibility
        elif not a.startswith("-"):
            opt["tla"] = a
        else:
            print(__doc__)
            return 2
        i += 1
    if not opt["cfg"]:
        print(__doc__)
        return 2

    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) if opt["backend"] == "nccl" else 0
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if opt["backend"] == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(opt["backend"])
    rank, world = dist.get_rank(), dist.get_world_size()
    say = print if rank == 0 else (lambda *a, **k: None)
    try:
        m = vt.Model.load(opt["cfg"], None if (opt["no_tla"] or not opt["tla"]) else opt["tla"])
    except vt.VsrmcError as e:
        say("Error: %s" % e)
        return 1
    lay = m.layout
    if not (opt["table_log2"] or opt["frontier_gib"] or opt["exact"] or opt["probe_at"]):
        return run_automatic(opt, m, rank, world, local_rank, say)      # (with -checkpoint / -recover too, since round 6: vsrmc_shard_loop_save / _restore)
    opt["table_log2"] = opt["table_log2"] or 26
    opt["frontier_gib"] = opt["frontier_gib"] or 2.0
    words = int(opt["frontier_gib"] * (1 << 30) / 8)
    states = max(1 << 12, words // 24)
    def make_engine(recover=None):
        return sharded.HipShardEngine(
            m, rank, world, device=local_rank, table_log2=opt["table_log2"], frontier_words=words, frontier_states=states,
            pending_entries=max(1 << 16, 3 * states if opt["exact"] else 0), cand_cap=int(1.5 * states / world) + (1 << 16),
            rec_cap=max(1 << 12, states // 8), rec_words_cap=max(1 << 16, words // 8), exact_ties=opt["exact"], recover=recover)

    try:
        if opt["recover"]:
            sc = sharded.ShardedChecker.restore(opt["recover"], make_engine, sharded.Exchanger())
        else:
            sc = sharded.ShardedChecker(make_engine(), sharded.Exchanger(), replicate_below=opt["replicate_below"])
    except (sharded.ShardError, vt.VsrmcError, OSError) as e:
        say("Error: %s" % e)
        return 1
    say("vsrmc: %s lowered: ReplicaCount=%d ClientCount=%d |Values|=%d StartViewOnTimerLimit=%d, %d permutation(s), invariant "
        "mask %d; %d rank(s), backend %s" % (["VSR.tla", "VR_STATE_TRANSFER.tla", "VR_APP_STATE.tla"][lay.module], lay.replica_count, lay.client_count, lay.value_count, lay.start_view_on_timer_limit,
                                             lay.permutations, lay.invariant_mask, world, opt["backend"]))
    if opt["recover"]:
        say("Recovered from checkpoint %s: level %d, %d distinct states found, %d states left on queue."
            % (opt["recover"], sc.level, sc.distinct, sc.n_frontier))
    else:
        say("Finished computing initial states: 1 distinct state generated.")
    t0 = t_chk = time.time()
    total_generated, code, last, deadlocked, probed = 0, 0, dict(n_new=1), False, None
    try:
        while sc.level < opt["max_depth"]:
            if opt["probe_at"] and sc.level + 1 == opt["probe_at"]:
                probed = sc.probe()
                total_generated += probed["generated"]
                say("Probe(%d): %d states generated, %d violating successors kept by their owners. (%.2f s)"
                    % (probed["level"], probed["generated"], probed["candidates"], time.time() - t0))
                if sc.violation is None:
                    say("No violation up to level %d; the search is incomplete beyond it." % probed["level"])
                break
            d = sc.step()
            last = d
            total_generated += d["generated"]
            dt = time.time() - t0
            if opt["json"]:
                say(json.dumps(dict(level=d["level"], generated=d["generated"], new=d["n_new"], distinct=d["distinct"],
                                    deadlocks=d["deadlocks"], replicated=bool(d.get("replicated")), seconds=round(dt, 4))))
            elif d["n_new"]:
                say("Progress(%d): %d states generated, %d distinct states found, %d states left on queue. (%.2f s)%s"
                    % (d["level"], total_generated, d["distinct"], d["n_new"], dt, "" if d.get("replicated") else "  [sharded]"))
            if sc.violation is not None:
                break
            if opt["check_deadlock"] and d["deadlocks"]:
                deadlocked = True
                break
            if d["n_new"] == 0:
                break
            if opt["checkpoint"]:
                # rank 0's clock decides for everybody (a checkpoint is a collective)
                due = 1 if time.time() - t_chk >= 60.0 * opt["checkpoint_minutes"] else 0
                if sc.x.allreduce([due if rank == 0 else 0], dist.ReduceOp.MAX)[0]:
                    sc.save(opt["checkpoint"])
                    say("Checkpointing of run %s completed (level %d)." % (opt["checkpoint"], sc.level))
                    t_chk = time.time()
    except (sharded.ShardError, vt.VsrmcError) as e:
        say("Error: %s" % e)
        code = 1
    dt = time.time() - t0
    if code == 0 and sc.violation is not None:
        v = sc.violation
        path = sc.probe_trace_fps() if v.get("probed") else sc.trace_fps(v["level"], v["fp"])    # every rank takes part in the walk
        if rank == 0:
            tr = sharded.replay_fps(m, path, device=local_rank)
            fps, _ = m.fingerprints(tr[-1][1], np.array([0, len(tr[-1][1])], dtype=np.uint64), device=local_rank)
            assert int(fps[0]) == v["fp"], "trace replay does not end in the violating state"
            mask = int(v["mask"])
            if v.get("probed"):     # the probe's mask is the union over every violating successor the ranks saw: ask the path's last state
                mask = m.check_trace([rec for _, rec in tr], device=local_rank)["inv_mask_last"]
            for b, name in enumerate(INVARIANTS):
                if (mask >> b) & 1:
                    print("Error: Invariant %s is violated." % name)
            print("Error: The behavior up to this point is:")
            for t, (action, rec) in enumerate(tr):
                print("State %d: <%s>\n%s\n" % (t + 1, action, m.format_state(rec)))
        code = 12
    elif code == 0 and deadlocked:
        say("Error: Deadlock reached (%d state(s) of level %d have no successor)." % (last["deadlocks"], sc.level - 1))
        code = 11
    elif code == 0 and last["n_new"] == 0:
        say("Model checking completed. No error has been found.")
    say("%d states generated, %d distinct states found, %d states left on queue." % (total_generated, sc.distinct, last["n_new"]))
    depth = sc.level if last["n_new"] else sc.level - 1
    say("The depth of the complete state graph search is %d.\nFinished in %.3f s (%.3g distinct states/s) on %d rank(s)."
        % (depth, dt, sc.distinct / dt if dt > 0 else 0.0, world))
    dist.barrier()
    dist.destroy_process_group()
    return code


def run_automatic(opt, m, rank, world, local_rank, say):
    """the default: the C++ level loop (direct RCCL, or gloo callbacks) through vsrmc_shard_loop_advance, sizes from the free memory"""
    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    lay = m.layout
    if opt["backend"] != "nccl":
        os.environ.setdefault("VSRMC_AUTOSIZE_SHARE", str(world))   # the ranks share device 0
    def make_engine(recover=None):
        return sharded.HipShardEngine(m, rank, world, device=local_rank, table_log2=0, frontier_words=0, frontier_states=0, pending_entries=0,
                                      cand_cap=0, rec_cap=1 << 22, rec_words_cap=1 << 28, native_only=True, recover=recover)
    try:
        comm = sharded.RcclComm(local_rank) if opt["backend"] == "nccl" else sharded.TorchHostComm()   # before the checker sizes itself
        if opt["recover"]:
            sc = sharded.NativeShardedChecker.restore(opt["recover"], make_engine, comm)
            eng = sc.e
        else:
            eng = make_engine()
            sc = sharded.NativeShardedChecker(eng, comm, replicate_below=opt["replicate_below"])
            sc.depth = sc.level
    except (sharded.ShardError, vt.VsrmcError, OSError) as e:
        say("Error: %s" % e)
        return 1
    say("vsrmc: %s lowered: ReplicaCount=%d ClientCount=%d |Values|=%d StartViewOnTimerLimit=%d, %d permutation(s), invariant mask %d; "
        "%d rank(s), backend %s, C++ level loop; per rank: seen-set 2^%d slots, record buffers 2 x %.1f GiB, %d candidates per peer"
        % (["VSR.tla", "VR_STATE_TRANSFER.tla", "VR_APP_STATE.tla"][lay.module], lay.replica_count, lay.client_count, lay.value_count,
           lay.start_view_on_timer_limit, lay.permutations, lay.invariant_mask, world, opt["backend"], int(eng.options.table_log2),
           int(eng.options.frontier_words) * 8 / (1 << 30), int(eng.cand_cap)))
    if opt["recover"]:
        say("Recovered from checkpoint %s: depth %d (%d level(s) in the seen-sets only), %d distinct states found."
            % (opt["recover"], sc.depth, sc.depth - sc.level, sc.distinct))
    else:
        say("Finished computing initial states: 1 distinct state generated.")
    t0 = t_chk = time.time()
    total_generated, code, last, deadlocked, kind, incomplete = 0, 0, dict(n_new=1, deadlocks=0), False, "level", False
    try:
        while sc.depth < opt["max_depth"]:
            if sc.room() == 2:                                   # (collective: every rank stops here together)
                incomplete = True
                break
            if opt["checkpoint"] and sc.depth > 1:
                # rank 0's clock decides for everybody (a checkpoint is a collective); between two units of progress — also after Virtual(..) lines
                due = torch.tensor([1 if (rank == 0 and time.time() - t_chk >= 60.0 * opt["checkpoint_minutes"]) else 0],
                                   device=("cuda:%d" % local_rank) if opt["backend"] == "nccl" else "cpu")
                dist.all_reduce(due, op=dist.ReduceOp.MAX)
                if int(due.item()):
                    sc.save(opt["checkpoint"])
                    say("Checkpointing of run %s completed (depth %d)." % (opt["checkpoint"], sc.depth))
                    t_chk = time.time()
            kind, d, b = sc.advance()
            last = d
            total_generated += d["generated"]
            dt = time.time() - t0
            if opt["json"]:
                say(json.dumps(dict(level=d["level"], kind=kind, generated=d["generated"], new=d["n_new"], distinct=d["distinct"], deadlocks=d["deadlocks"],
                                    seconds=round(dt, 4))))
            elif d["n_new"] and kind == "level":
                say("Progress(%d): %d states generated, %d distinct states found, %d states left on queue. (%.2f s)%s"
                    % (d["level"], total_generated, d["distinct"], d["n_new"], dt, "" if d.get("replicated") else "  [sharded]"))
            elif d["n_new"]:
                say("Virtual(%d): %d states generated, %d distinct states found, %d states in the level (not stored). (%.2f s)  [sharded]"
                    % (d["level"], total_generated, d["distinct"], d["n_new"], dt))
            if b is not None:
                total_generated += b["generated"]
                say("Probe(%d): %d states generated from the %d states of level %d. (%.2f s)" % (b["level"], b["generated"], d["n_new"], d["level"], dt))
            if sc.violation is not None:
                break
            if opt["check_deadlock"] and d["deadlocks"]:
                deadlocked = True
                break
            if d["n_new"] == 0:
                break
    except (sharded.ShardError, vt.VsrmcError) as e:
        say("Error: %s" % e)
        code = 1
    dt = time.time() - t0
    if code == 0 and sc.violation is not None:
        v = sc.violation
        path = sc.violation_trace_fps()                          # every rank takes part in the walk
        if rank == 0:
            tr = sharded.replay_fps(m, path, device=local_rank)
            fps, _ = m.fingerprints(tr[-1][1], np.array([0, len(tr[-1][1])], dtype=np.uint64), device=local_rank)
            assert int(fps[0]) == v["fp"], "trace replay does not end in the violating state"
            mask = int(v["mask"])
            if v.get("probed"):     # the probe's mask is the union over every violating successor the ranks saw: ask the path's last state
                mask = m.check_trace([rec for _, rec in tr], device=local_rank)["inv_mask_last"]
            for bit, name in enumerate(INVARIANTS):
                if (mask >> bit) & 1:
                    print("Error: Invariant %s is violated." % name)
            print("Error: The behavior up to this point is:")
            for t, (action, rec) in enumerate(tr):
                print("State %d: <%s>\n%s\n" % (t + 1, action, m.format_state(rec)))
        code = 12
    elif code == 0 and deadlocked:
        say("Error: Deadlock reached (%d state(s) of level %d have no successor)." % (last["deadlocks"], sc.depth - 1))
        code = 11
    elif code == 0 and incomplete:
        say("Model checking INCOMPLETE at depth %d: a rank's seen-set shard is more than 85 %% full (no error has been found up to that depth)." % sc.depth)
        code = 4
    elif code == 0 and last["n_new"] == 0:
        say("Model checking completed. No error has been found.")
    say("%d states generated, %d distinct states found, %d states left on queue." % (total_generated, sc.distinct, last["n_new"]))
    depth = sc.depth - (1 if (last["n_new"] == 0 and kind == "level") else 0)   # (an empty sharded level still advances the ranks' level counters)
    say("The depth of the complete state graph search is %d.\nFinished in %.3f s (%.3g distinct states/s) on %d rank(s)."
        % (depth, dt, sc.distinct / dt if dt > 0 else 0.0, world))
    sc.close()
    dist.barrier()
    dist.destroy_process_group()
    return code


if __name__ == "__main__":
    sys.exit(main())

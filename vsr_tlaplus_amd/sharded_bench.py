"""bench.py's N > 1 leg: the README defect configuration (BASELINE configs[2]) with the seen-set sharded over N GPUs, through the
automatic level scheme of the C++ level loop (csrc/vsr_shard_loop.hpp: vsrmc_shard_loop_advance).  One rank per GPU (RANK / LOCAL_RANK /
WORLD_SIZE from torch.distributed.run), backend "nccl" (= RCCL over xGMI, called directly from the loop).

Nothing here names a level or a size: every rank sizes its seen-set shard, record buffers and exchange buffers from the free memory of
its GPU, stores a level while every rank predicts that its part of the next one fits, and goes on through the seen-sets alone
(virtual / regenerated / streamed / probed levels) when one does not — at N >= 4 everything up to the violation is stored, at N = 2
(and N = 1 under VSR_BENCH_SHARDED=1) the last levels are not.  Every level's figures over all ranks are asserted against the CPU oracle's
fixture on every run.  `--workload config2` runs BASELINE configs[1] the same way."""
import json
import os
import time

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0
XGMI_LINK_GBS = 153.0        # one xGMI link of an MI355X, one direction (7 links per GPU: a fully connected 8-GPU node gives every pair its own)
XGMI_ACHIEVABLE = 0.70       # ASSUMPTION (no multi-GPU node was available to measure it): the share of a link's peak grouped ncclSend / ncclRecv of 16-byte
                             # candidates sustain — the modelled critical path below prices the exchange at 107 GB/s per link, not at peak
# levels with fewer new states than this are explored by every rank on its own (no collectives): a level of a million states
# is under a millisecond of kernel time, less than the exchanges of one sharded level
REPLICATE_BELOW = int(os.environ.get("VSR_BENCH_REPLICATE_BELOW", 1 << 20))


def main(args, bench):
    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VSR_BENCH_BACKEND", "nccl")      # "gloo": functional check of this leg on a one-GPU box
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if backend != "nccl":                                       # (all ranks share device 0, buckets staged through the host)
        local_rank = 0
        os.environ.setdefault("VSRMC_AUTOSIZE_SHARE", str(world_env))
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    readme = args.workload != "config2"
    cfg = bench.README if readme else bench.CONFIG
    if readme:
        E = bench.load_readme_expect()
        want_levels = E["levels"]
        expect = dict(distinct=1821858767, depth=24, viol_fp=E["viol_fp"], probe_generated=E["probe_generated"])
    else:
        want_levels = [dict(n_new=lv["new"], generated=lv["generated"], deadlocks=lv["deadlocks"], max_bag=lv["max_bag"], act_generated=lv["act_generated"])
                       for lv in bench.EXPECT["levels"]]
        expect = dict(distinct=bench.EXPECT["distinct"], depth=bench.EXPECT["depth"], viol_fp=bench.EXPECT["viol_fp"], probe_generated=None)
    m = vt.Model.from_constants(R=cfg["R"], C_=cfg["C"], n=cfg["n"], L=cfg["L"])
    # the communicator first: RCCL's own device buffers must be there when the checker sizes itself from what is free
    comm = sharded.RcclComm(local_rank) if backend == "nccl" else sharded.TorchHostComm()
    t0 = time.perf_counter()
    eng = sharded.HipShardEngine(m, rank, world, device=local_rank, table_log2=0, frontier_words=0, frontier_states=0, pending_entries=0, cand_cap=0,
                                 rec_cap=1 << 22, rec_words_cap=1 << 28, native_only=True)
    setup = time.perf_counter() - t0
    S = dict(alg_bytes=0.0, launches=0, distinct=0, ttfv=[], kernel_ms=0.0, stored_ms=0.0, deep_ms=0.0, stored_levels=0, passes=[], sent=0, moved=0)

    def one_run(record):
        eng.reset()
        sc = sharded.NativeShardedChecker(eng, comm, replicate_below=REPLICATE_BELOW)
        sc.depth = sc.level
        t0 = time.perf_counter()
        s_rec, n_prev = 8.0 * (int(m.layout.fixed_words) + int(m.layout.permutations)), 1
        alg, kms, dms, launches, stored, passes, found = 0.0, 0.0, 0.0, 0, 0, [], None
        units = []                                              # per unit of progress: (rank 0's kernel seconds, rank 0's bytes sent)
        while found is None:
            sent0 = sc.bytes_sent
            kind, a, b = sc.advance()
            units.append(((a["expand_ms"] + a["materialize_ms"] + (b["expand_ms"] if b is not None else 0.0)) / 1e3, sc.bytes_sent - sent0))
            assert a["n_new"], "the search is exhausted before the violation"
            want = want_levels[a["level"] - 1]
            assert (a["n_new"], a["generated"]) == (want["n_new"], want["generated"]), (a["level"], a["n_new"], a["generated"])
            if want.get("deadlocks") is not None:
                assert a["deadlocks"] == want["deadlocks"], a["level"]
            if want.get("act_generated") is not None:
                assert [int(x) for x in a["act_generated"][1:16]] == want["act_generated"][1:16], a["level"]
            # algorithmic bytes of the whole job (every state read once and written once, 8 B per generated successor and per new key)
            alg += s_rec * n_prev + 8.0 * a["generated"] + 8.0 * a["n_new"]
            if kind == "level":
                s_rec = 8.0 * a["record_words"] / a["n_new"] if a["record_words"] else s_rec
                alg += s_rec * a["n_new"]
                kms += a["expand_ms"]
                launches += 1
                stored = a["level"]
                if a["viol_mask"]:
                    found = a
            else:
                alg += s_rec * a["n_new"]
                dms += a["expand_ms"] + a["materialize_ms"]
                launches += a["launches"]
                row = dict(level=a["level"], seconds=round(a["seconds"], 4), launches=a["launches"])
                if a["viol_mask"]:
                    found = a
                elif b is not None:
                    alg += s_rec * a["n_new"] + 8.0 * b["generated"]
                    dms += b["expand_ms"]
                    row.update(probed_level=b["level"], probe_seconds=round(b["seconds"], 4))
                    if b["viol_mask"]:
                        found = b
                passes.append(row)
            n_prev = a["n_new"]
        fps = sc.violation_trace_fps()                          # counter-example reconstructed = found (collective walk through the shards)
        if rank == 0:
            tr = sharded.replay_fps(m, fps, device=local_rank)
            assert len(tr) == expect["depth"]
        dt = time.perf_counter() - t0
        assert sc.distinct == expect["distinct"] and found["level"] == expect["depth"], (sc.distinct, found["level"])
        assert expect["viol_fp"] is None or found["viol_fp"] == expect["viol_fp"]
        assert expect["probe_generated"] is None or kind == "level" or b is None or not b["viol_mask"] or b["generated"] == expect["probe_generated"]
        S["moved"] = sc.moved
        S["sent"] += sc.bytes_sent
        S["units"] = units
        S["overlap"] = sc.overlap_stats()
        sc.close()
        if record:
            S["distinct"] += sc.distinct
            S["ttfv"].append(dt)
            S["alg_bytes"] += alg
            S["stored_ms"] += kms
            S["deep_ms"] += dms
            S["launches"] += launches
            S["stored_levels"], S["passes"] = stored, passes

    for _ in range(args.warmup):
        one_run(False)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_run(True)
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        value = S["distinct"] / elapsed
        kernel_ms = S["stored_ms"] + S["deep_ms"]                # rank 0's k_expand time (HIP events on the checker's stream)
        avg_launch_s = kernel_ms / 1e3 / max(1, S["launches"])
        achieved = S["alg_bytes"] / world / max(kernel_ms / 1e3, 1e-12) / 1e9   # per GPU: the job's algorithmic bytes / N over rank 0's kernel time
        out = {
            "metric": "distinct states/sec (whole node) + time-to-first-violation, VSR 3-replica", "value": round(value, 1), "unit": "distinct states/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": ("VSR.tla BFS, ReplicaCount=3 ClientCount=1 Values={v1,v2,v3} StartViewOnTimerLimit=3 (the reference README's state-"
                                    "transfer-defect configuration, README:13-18 = BASELINE configs[2]), VIEW+SYMMETRY, to the first violation at depth 24: "
                                    "1821858767 distinct states" if readme else
                                    "VSR.tla BFS, ReplicaCount=3 ClientCount=1 Values={v1,v2} StartViewOnTimerLimit=2 (BASELINE configs[1] = shipped VSR.cfg), "
                                    "VIEW+SYMMETRY, to first violation: 28 levels, 319228361 distinct states"),
                       "parallelism": "seen-set sharded by fingerprint over %d ranks, all-to-all of (fp, key) candidates per level / pass (RCCL), records "
                                      "stay with their generator, levels below %d new states replicated; automatic level scheme: levels 1-%d stored, %d "
                                      "pass(es) through the seen-sets alone" % (world, REPLICATE_BELOW, S["stored_levels"], len(S["passes"])),
                       "sized_from_free_hbm": dict(table_slots_log2_per_rank=int(eng.options.table_log2), frontier_words_per_buffer=int(eng.options.frontier_words),
                                                   candidates_per_peer=int(eng.cand_cap)),
                       "level_loop": "native C++ (csrc/vsr_shard_loop.hpp), " + ("direct RCCL: grouped ncclSend/ncclRecv" if backend == "nccl" else "gloo callbacks, host-staged")},
            "time_to_first_violation_s": round(sum(S["ttfv"]) / len(S["ttfv"]), 4), "setup_s": round(setup, 2),
            "xgmi_bytes_sent_rank0_per_step": int(S["sent"] / max(1, args.steps + args.warmup)),
            # the exchange priced against the fabric BEFORE any multi-GPU run exists: rank 0's bytes spread over the links to its N - 1 peers at link
            # peak, beside rank 0's kernel time for the same step — which of the two bounds a step (they do not overlap yet: DESIGN.md §7)
            "exchange": (lambda b, ks: {"xgmi_bytes_sent_rank0_per_step": int(b), "links_used": max(1, min(world - 1, 7)), "link_peak_GBs": XGMI_LINK_GBS,
                                        "xgmi_s_at_link_peak": round(b / (XGMI_LINK_GBS * 1e9 * max(1, min(world - 1, 7))), 4),
                                        "kernel_s_per_step": round(ks, 4),
                                        "bound_by": "kernel" if ks >= b / (XGMI_LINK_GBS * 1e9 * max(1, min(world - 1, 7))) else "exchange",
                                        "note": "a regenerated level costs no exchange since round 5 (generator-side winner set): what is left are the (fp, key) "
                                                "announcements of the levels that are inserted, their verdict bytes and the small all-gathers"})(
                S["sent"] / max(1, args.steps + args.warmup), kernel_ms / 1e3 / max(1, args.steps)),
            # the modelled critical path of one run (the last timed one), unit of progress by unit: rank 0's kernel time against its bytes over the links to its
            # N - 1 peers at an ACHIEVABLE rate (XGMI_ACHIEVABLE of link peak: an assumption, stated, not a measurement).  sequential = kernel + exchange
            # (rounds 2-5); overlapped = the larger of the two per unit (round 6: the exchange of slice k runs under k_expand of slice k + 1)
            "exchange_model": (lambda units, rate: {
                "achievable_link_GBs": round(XGMI_LINK_GBS * XGMI_ACHIEVABLE, 1), "links_used": max(1, min(world - 1, 7)), "assumption": "%.0f %% of link peak" % (100 * XGMI_ACHIEVABLE),
                "kernel_s": round(sum(k_ for k_, _ in units), 4), "exchange_s": round(sum(x / rate for _, x in units), 4),
                "critical_path_sequential_s": round(sum(k_ + x / rate for k_, x in units), 4),
                "critical_path_overlapped_s": round(sum(max(k_, x / rate) for k_, x in units), 4),
                "units_bound_by_exchange": sum(1 for k_, x in units if x / rate > k_), "units": len(units),
                "levels_run_in_slices": S.get("overlap", (0, 0))[0], "slices": S.get("overlap", (0, 0))[1]})(
                S.get("units", []), XGMI_LINK_GBS * XGMI_ACHIEVABLE * 1e9 * max(1, min(world - 1, 7))),
            "records_moved_by_rebalancing_rank0": S["moved"],
            "deep_passes": S["passes"],
            "roofline": {"bound": "hbm", "kernel": "k_expand (rank 0; the job's algorithmic bytes / N over rank 0's kernel time)", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": None, "traffic_note": "PMC passes exist for the N = 1 line only (profiles/): no counter run on more than one GPU",
                         "avg_launch_ms": round(1e3 * avg_launch_s, 4), "launches": S["launches"] // max(1, args.steps),
                         "kernel_ms_per_step": {"k_expand": round(kernel_ms / args.steps, 3), "k_expand_stored_levels": round(S["stored_ms"] / args.steps, 3),
                                                "k_expand_deep_passes": round(S["deep_ms"] / args.steps, 3)}},
        }
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = bench.cpu_baseline(args.cpu_seconds, cfg)
        print(json.dumps(out))
    dist.barrier()
    eng.close()
    dist.destroy_process_group()

"""bench.py's N > 1 leg: the same workload (config 2 to the first violation) with the seen-set sharded over N GPUs.
One rank per GPU (RANK / LOCAL_RANK / WORLD_SIZE from torch.distributed.run), backend "nccl" (= RCCL over xGMI)."""
import json
import math
import os
import time

import torch
import torch.distributed as dist

# per-level maxima of the workload (tests/golden/config2_violation.json): sizes every buffer
MAX_NEW, MAX_GENERATED, MAX_WORDS, TOTAL = 80003390, 217755238, 3029987047, 319228361
HBM_PEAK_GBS = 8000.0
# levels with fewer new states than this are explored by every rank on its own (no collectives): a level of a million states
# is under a millisecond of kernel time, less than the exchanges of one sharded level
REPLICATE_BELOW = int(os.environ.get("VSR_BENCH_REPLICATE_BELOW", 1 << 20))


def main(args, CONFIG, EXPECT):
    import vsr_tlaplus_amd as vt
    from vsr_tlaplus_amd import sharded
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VSR_BENCH_BACKEND", "nccl")      # "gloo": functional check of this leg on a one-GPU box
    if backend != "nccl":                                       # (all ranks share device 0, buckets staged through the host)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    # slack 1.4: frontier imbalance between ranks (rebalancing tolerates 25 %) and the speculative successors that lose (≈ 5 %).
    # On top of that every resident block of k_expand leaves one partly used index chunk (<= 8192 indices) and one word chunk
    # (<= 262144 words) behind per level: up to 4 blocks per CU.
    slack = 1.4
    blocks = 4 * torch.cuda.get_device_properties(local_rank).multi_processor_count
    tail_idx, tail_words = blocks * 8192, blocks * 262144
    per_rank = lambda x, extra=0: int(x / world * slack) + extra + (1 << 16)   # noqa: E731
    per_pair = lambda x, extra=0: int(x / world / world * slack) + extra + (1 << 16)   # noqa: E731
    m = vt.Model.from_constants(R=CONFIG["R"], C_=CONFIG["C"], n=CONFIG["n"], L=CONFIG["L"])
    table_log2 = max(20, int(math.ceil(math.log2(4.4 * TOTAL / world))))
    eng = sharded.HipShardEngine(
        m, rank, world, device=local_rank, table_log2=table_log2, frontier_words=per_rank(MAX_WORDS, 2 * tail_words),
        frontier_states=per_rank(MAX_NEW, 2 * tail_idx), pending_entries=1 << 16,   # single-pass levels: no pending list
        cand_cap=per_pair(MAX_GENERATED), filter_log2=max(20, int(math.ceil(math.log2(2.0 * TOTAL / world)))),
        # records stay with their generator; these two only bound one rebalancing move to one peer (early, small levels)
        rec_cap=1 << 22, rec_words_cap=1 << 28,
        keep_trace=True, trace_entries=per_rank(TOTAL, 16 * tail_idx))
    x = sharded.Exchanger()
    # the level loop: native (C++, csrc/vsr_shard_loop.hpp) over RCCL called directly — the default on "nccl" — or the Python loop over
    # torch.distributed (VSR_BENCH_PYLOOP=1; always on "gloo" unless VSR_BENCH_NATIVE=1 asks for the native loop over gloo callbacks)
    native = (backend == "nccl" and not os.environ.get("VSR_BENCH_PYLOOP")) or bool(os.environ.get("VSR_BENCH_NATIVE"))
    comm = None
    if native:
        comm = sharded.RcclComm(local_rank) if backend == "nccl" else sharded.TorchHostComm()
    S = dict(alg_bytes=0.0, launches=0, distinct=0, ttfv=[])
    moved = [0]
    sent = [0]

    def one_run(record):
        eng.reset()
        eng.kernel_ms = dict(expand=0.0, materialize=0.0)
        if native:
            sc = sharded.NativeShardedChecker(eng, comm, replicate_below=REPLICATE_BELOW)
        else:
            sc = sharded.ShardedChecker(eng, x, replicate_below=REPLICATE_BELOW)
        t0 = time.perf_counter()
        cur_words = (int(m.layout.fixed_words) + int(m.layout.permutations)) if sc.e.local_distinct() else 0
        while True:
            d = sc.step()
            loc = d["local"]
            if record and loc["frontier"]:
                S["launches"] += 1
                S["alg_bytes"] += 8.0 * cur_words + 8.0 * loc["generated"] + 8.0 * loc["n_new"] + 8.0 * loc["record_words"]
            cur_words = loc["record_words"]
            if d["n_new"] == 0 or sc.violation is not None:
                break
        if sc.violation is not None:                          # counter-example reconstructed = found
            fps = sc.trace_fps(sc.violation["level"], sc.violation["fp"])
            if rank == 0:
                tr = sharded.replay_fps(m, fps, device=local_rank)
                assert len(tr) == sc.violation["level"]
        dt = time.perf_counter() - t0
        assert sc.distinct == EXPECT["distinct"] and sc.level == EXPECT["depth"], (sc.distinct, sc.level)
        assert sc.violation and (EXPECT["viol_fp"] is None or sc.violation["fp"] == EXPECT["viol_fp"])
        moved[0] = sc.moved
        sent[0] += sc.bytes_sent if native else 0
        if native:
            sc.close()
        if record:
            S["distinct"] += sc.distinct
            S["ttfv"].append(dt)
            S["expand_ms"] = S.get("expand_ms", 0.0) + eng.kernel_ms["expand"]
            S["mat_ms"] = S.get("mat_ms", 0.0) + eng.kernel_ms["materialize"]

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
        avg_launch_s = S["expand_ms"] / 1e3 / max(1, S["launches"])
        achieved = S["alg_bytes"] / max(1, S["launches"]) / max(avg_launch_s, 1e-12) / 1e9
        print(json.dumps({
            "metric": "distinct states/sec (whole node), VSR 3-replica", "value": round(value, 1), "unit": "distinct states/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "VSR.tla BFS, ReplicaCount=3 ClientCount=1 Values={v1,v2} StartViewOnTimerLimit=2 "
                                   "(BASELINE configs[1] = shipped VSR.cfg), VIEW+SYMMETRY, to first violation: 28 levels, "
                                   "319228361 distinct states", "parallelism": "seen-set sharded by fingerprint over %d ranks, "
                                   "all-to-all of (fp, key) candidates per level (RCCL), levels below %d new states replicated"
                                   % (world, REPLICATE_BELOW), "table_slots_log2_per_rank": table_log2,
                       "level_loop": ("native C++ (csrc/vsr_shard_loop.hpp), " + ("direct RCCL: grouped ncclSend/ncclRecv" if backend == "nccl" else "gloo callbacks, host-staged"))
                                     if native else "Python over torch.distributed (vsr_tlaplus_amd/sharded.py)"},
            "time_to_first_violation_s": round(sum(S["ttfv"]) / len(S["ttfv"]), 4),
            "xgmi_bytes_sent_rank0_per_step": int((sent[0] if native else x.bytes_sent) / max(1, args.steps + args.warmup)),
            "records_moved_by_rebalancing_rank0": moved[0],
            "roofline": {"bound": "hbm", "kernel": "k_expand (rank 0)", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                         "avg_launch_ms": round(1e3 * avg_launch_s, 4), "launches": S["launches"],
                         "kernel_ms_per_step": {"k_expand": round(S["expand_ms"] / args.steps, 3),
                                                "k_apply_verdict": round(S["mat_ms"] / args.steps, 3)}},
        }))
    dist.barrier()
    dist.destroy_process_group()

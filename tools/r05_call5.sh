#!/bin/bash
# round-5 GPU call 5: after the winner-set fixes (world 1 needs none; kept across resets) — the sharded tests, the deep-search tests, the world-1 RCCL leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_sharded_gloo.py tests/test_deep_search.py -q -m gpu 2>&1 | tail -n 6 > gpurun_out/r05_t7.log
tail -n 3 gpurun_out/r05_t7.log
VSR_BENCH_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline \
  > gpurun_out/r05_sharded_world1_rccl_bench.json 2> gpurun_out/r05_sharded_world1.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_sharded_world1_rccl_bench.json").read().strip().splitlines()[-1])
print("world1 rccl ms/step", d["ms_per_step"], d["roofline"]["kernel_ms_per_step"], d["roofline"]["launches"], [(p["level"], p["seconds"], p["launches"]) for p in d["deep_passes"]])
PY
python bench.py --workload readme --steps 3 --warmup 1 --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('unsharded ms/step', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
VSR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline \
  > gpurun_out/r05_sharded_world2_gloo_bench.json 2> gpurun_out/r05_sharded_world2.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05_sharded_world2_gloo_bench.json").read().strip().splitlines()[-1])
print("world2 gloo ms/step", d["ms_per_step"], d["exchange"]["xgmi_bytes_sent_rank0_per_step"], d["roofline"]["launches"])
PY

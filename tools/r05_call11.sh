#!/bin/bash
# round-5 GPU call 11 (last): the claim bitmap also through the level loop at world 1 — sharded + deep tests, the profile set again (final sources), world-1 RCCL leg
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_sharded_gloo.py tests/test_deep_search.py -q -m gpu 2>&1 | tail -n 4 > gpurun_out/r05_t11.log
tail -n 2 gpurun_out/r05_t11.log
python -m pytest tests/test_gpu_parity.py -q -m gpu -k "whole_workload or probe2 or probe3" 2>&1 | tail -n 2
VSR_BENCH_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline \
  > gpurun_out/r05_sharded_world1_rccl_bench.json 2> gpurun_out/r05_sharded_world1.err
python -c "
import json
d = json.loads(open('gpurun_out/r05_sharded_world1_rccl_bench.json').read().strip().splitlines()[-1])
print('world1 rccl ms/step', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches'])"
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
python -c "
import json
d = json.loads(open('gpurun_out/r05_bench.json').read().strip().splitlines()[-1])
print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config2']['value'], d['config4']['value'])"

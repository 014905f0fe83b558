#!/bin/bash
# round-6 final GPU call: the whole GPU suite, smoke, the round's profile set (tools/profile_round.sh r06), the sharded legs that one GPU allows, the layout and
# the claim-partition measurements, the analysis models to exhaustion (the kernels of all three models changed: MachineLICM off, cooperative copy)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -n 45 > gpurun_out/r06_gputests_final.log
tail -n 3 gpurun_out/r06_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1
tail -c 600 gpurun_out/r06_profile_round.log
VSR_BENCH_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline \
  > gpurun_out/r06_sharded_world1_rccl_bench.json 2> gpurun_out/r06_sharded_world1.err
tail -c 400 gpurun_out/r06_sharded_world1_rccl_bench.json
VSR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline \
  > gpurun_out/r06_sharded_world2_gloo_bench.json 2> gpurun_out/r06_sharded_world2.err
tail -c 400 gpurun_out/r06_sharded_world2_gloo_bench.json
timeout 600 python tools/bench_layout.py 24 20 > gpurun_out/r06_bench_layout.json 2>&1
timeout 600 python tools/bench_layout.py 27 20 >> gpurun_out/r06_bench_layout.json 2>&1
cat gpurun_out/r06_bench_layout.json
timeout 900 tools/bench_claim_partition.sh > gpurun_out/r06_claim_partition.log 2>&1
cat gpurun_out/r06_claim_partition.json
timeout 1200 python tools/run_models_deep.py --max-seconds 900 > gpurun_out/r06_models_exhausted_two_seeds.jsonl 2> gpurun_out/r06_models.err
cut -c1-420 gpurun_out/r06_models_exhausted_two_seeds.jsonl; tail -n 2 gpurun_out/r06_models.err

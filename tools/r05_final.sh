#!/bin/bash
# round-5 final GPU call: the whole GPU suite, the round's profile set (tools/profile_round.sh r05), the sharded legs that one GPU allows, the CLI with -audit
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -n 45 > gpurun_out/r05_gputests_final.log
tail -n 3 gpurun_out/r05_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
bash tools/profile_round.sh r05 > gpurun_out/r05_profile_round.log 2>&1
tail -c 600 gpurun_out/r05_profile_round.log
VSR_BENCH_SHARDED=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline \
  > gpurun_out/r05_sharded_world1_rccl_bench.json 2> gpurun_out/r05_sharded_world1.err
tail -c 400 gpurun_out/r05_sharded_world1_rccl_bench.json
VSR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 1 --warmup 0 --no-cpu-baseline \
  > gpurun_out/r05_sharded_world2_gloo_bench.json 2> gpurun_out/r05_sharded_world2.err
tail -c 400 gpurun_out/r05_sharded_world2_gloo_bench.json
python tools/make_cfg.py gpurun_out/README.cfg 3 1 "v1, v2, v3" 3
vsr_tlaplus_amd/vsrmc -config gpurun_out/README.cfg -noTLA -audit > gpurun_out/r05_cli_readme_audit.log 2>&1
grep -v "|->" gpurun_out/r05_cli_readme_audit.log | grep -v "^$" | head -60
timeout 1200 python tools/run_models_deep.py --max-seconds 900 > gpurun_out/r05_models_exhausted_two_seeds.jsonl 2> gpurun_out/r05_models.err
cut -c1-420 gpurun_out/r05_models_exhausted_two_seeds.jsonl; tail -n 2 gpurun_out/r05_models.err

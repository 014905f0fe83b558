#!/bin/bash
# round-5 GPU call 3: both analysis models to exhaustion under both fingerprint seeds (re-basing), config 4 to level 19 (both seeds)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 1500 python tools/run_models_deep.py --max-seconds 1200 > gpurun_out/r05_models_exhausted_two_seeds.jsonl 2> gpurun_out/r05_models.err
cut -c1-700 gpurun_out/r05_models_exhausted_two_seeds.jsonl; tail -n 3 gpurun_out/r05_models.err
python -m pytest tests/test_gpu_parity.py -q -k "whole_workload and config4" 2>&1 | tail -n 8 > gpurun_out/r05_t6.log
tail -n 4 gpurun_out/r05_t6.log

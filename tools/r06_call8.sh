#!/bin/bash
# round 6, GPU call 8: the five-block instantiation beside the four-block one (host picks per launch), LDS sweep of the five-block shape, overlapped
# exchange + sharded checkpoints through the GPU test suite
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call8.log
: > $L
README_VARIANTS="prod" timeout 2400 tools/ab_bench.sh prod prod:VSRMC_NO_OCC5=1 prod:VSRMC_LDS5=40000,VSRMC_CCAP5=640 prod:VSRMC_LDS5=40000,VSRMC_CCAP5=896 prod:VSRMC_LDS5=40000,VSRMC_CCAP5=1024 prod:VSRMC_LDS5=40000,VSRMC_CCAP5=512 >> $L 2>&1
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r06_gputests_call8.log 2>&1
tail -n 25 gpurun_out/r06_gputests_call8.log >> $L
grep -v amdgpu.ids $L | cut -c1-400 | tail -50

#!/bin/bash
# round 6, GPU call 7: why is the five-block shape slow in the product build?  + the layout benchmark + the sharded deep checkpoint tests
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call7.log
: > $L
timeout 600 python tools/bench_layout.py 24 20 >> $L 2>&1
timeout 600 python tools/bench_layout.py 27 20 >> $L 2>&1
timeout 1800 tools/ab_bench.sh prod prod:VSRMC_CCAP=768 prod:VSRMC_MAX_BPC=4 prod_noredo prod_noredo:VSRMC_CCAP=768 old5 >> $L 2>&1
timeout 1500 python -m pytest tests/test_sharded_gloo.py -x -q -m gpu -k "checkpointed_and_recovered or sharded_cli" > gpurun_out/r06_gputests_call7.log 2>&1
tail -n 12 gpurun_out/r06_gputests_call7.log >> $L
grep -v amdgpu.ids $L | cut -c1-700 | tail -40

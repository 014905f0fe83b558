#!/bin/bash
# round 6, GPU call 5: the whole GPU test suite on the new product build (MachineLICM off, cooperative copy, five blocks per CU for config 2's kernel,
# limit re-check of probe passes, trace to any violator), then the product build against itself without each of the two kernel changes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call5.log
: > $L
timeout 1500 python -m pytest tests -x -q -m gpu --durations=5 > gpurun_out/r06_gputests_call5.log 2>&1
tail -n 12 gpurun_out/r06_gputests_call5.log >> $L
README_VARIANTS="prod prod_nocoop" timeout 1800 tools/ab_bench.sh prod prod_occ4 prod_nocoop >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400 | tail -40

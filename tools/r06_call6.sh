#!/bin/bash
# round 6, GPU call 6: the GPU test suite on the product build (work-list overflow taken again in pieces, five blocks per CU for config 2's kernel),
# then the product build with / without: direct refs, the bottom barrier, five blocks
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call6.log
: > $L
timeout 1800 python -m pytest tests -q -m gpu --durations=5 > gpurun_out/r06_gputests_call6.log 2>&1
tail -n 15 gpurun_out/r06_gputests_call6.log >> $L
README_VARIANTS="prod" timeout 1800 tools/ab_bench.sh prod prod_occ4 p_drefs p_notail p_dn >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400 | tail -40

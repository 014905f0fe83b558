#!/bin/bash
# round 6, GPU call 10: the whole GPU suite on the product build; direct refs + no bottom barrier on both workloads
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call10.log
: > $L
timeout 2700 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r06_gputests_call10.log 2>&1
tail -n 20 gpurun_out/r06_gputests_call10.log >> $L
README_VARIANTS="prod p_dn" timeout 1500 tools/ab_bench.sh prod p_dn >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400 | tail -40

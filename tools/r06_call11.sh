#!/bin/bash
# round 6, GPU call 11: the whole GPU suite on the product build; direct refs + no bottom barrier on both workloads; the region-partitioned claim benchmark
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call11.log
: > $L
timeout 2700 python -m pytest tests -q -m gpu --durations=8 > gpurun_out/r06_gputests_call11.log 2>&1
tail -n 20 gpurun_out/r06_gputests_call11.log >> $L
README_VARIANTS="prod p_dn" timeout 1500 tools/ab_bench.sh prod p_dn >> $L 2>&1
timeout 900 tools/bench_claim_partition.sh > gpurun_out/r06_claim_partition.log 2>&1
tail -n 40 gpurun_out/r06_claim_partition.log >> $L
grep -v amdgpu.ids $L | cut -c1-400 | tail -70

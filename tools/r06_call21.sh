#!/bin/bash
# round 6, GPU call 21: instruction-scheduler switches of the compiler (relaxed-occupancy scheduling, no post-RA scheduler, no machine scheduler) on the final kernel
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call21.log
: > $L
README_VARIANTS="base relax nopost nomis" timeout 2400 tools/ab_bench.sh base relax nopost nomis >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -40

#!/bin/bash
# round 6, GPU call 4 (all on MachineLICM off, no phase clocks): wave-cooperative copy, 8-word copy, refs one tile ahead, five blocks per CU
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call4.log
: > $L
README_VARIANTS="n0 coop rahead cra occ5" timeout 3000 tools/ab_bench.sh n0 coop copy8 rahead cra occ5 cocc5 >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -40

#!/bin/bash
# round 6, GPU call 23: the cooperative copy with four records per trip instead of two
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call23.log
: > $L
README_VARIANTS="base c4" timeout 2400 tools/ab_bench.sh base c4 >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -40

#!/bin/bash
# round 6, GPU call 18: tiles of 128 records at three blocks per CU (work list 1024, overflow to the host's list) against the product's 64-record tiles at five / four
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call18.log
: > $L
README_VARIANTS="base t128" timeout 2400 tools/ab_bench.sh base t128 >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400 | tail -40

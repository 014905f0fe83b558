#!/bin/bash
# round 6, GPU call 17: with the cursor out of the way, again: a tile = as many records as fill ONE apply round (-DVSR_TAKE=232 / 248), cursor drawn 512 / 1024 records at a time
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call17.log
: > $L
README_VARIANTS="base take232 take248" timeout 2400 tools/ab_bench.sh base take232 take248 take232k >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-400 | tail -40

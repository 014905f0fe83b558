#!/bin/bash
# round 6, GPU call 24: the value bytes of a patched bag word permuted once for its old and its new hash term (-DVSR_HASH_PAIR=1)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call24.log
: > $L
README_VARIANTS="base hp" timeout 2400 tools/ab_bench.sh base hp >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -40

#!/bin/bash
# round 6, GPU call 20: 16-byte stores for the replica block and the view hashes of a successor; the five-block work list swept again now that the cursor is not the bound
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call20.log
: > $L
README_VARIANTS="base ps" timeout 2400 tools/ab_bench.sh base ps "base:VSRMC_CCAP5=768" "base:VSRMC_CCAP5=640" "base:VSRMC_CCAP5=1024,VSRMC_LDS5=32768" "base:VSRMC_NO_OCC5=1" >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -40

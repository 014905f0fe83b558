#!/bin/bash
# round 6, GPU call 2: which barriers wave 1 waits at; phase split of config 2; barrier / ref-load / round-mapping experiments
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call2.log
: > $L
for wl in config2 readme; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_diag3.so timeout 300 python tools/wave_diag.py $wl groups >> $L 2>&1
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_base.so timeout 300 python tools/phase_split.py $wl >> $L 2>&1
done
README_VARIANTS="base nd rrev" timeout 2400 tools/ab_bench.sh base rrev notail drefs nd >> $L 2>&1
grep -v amdgpu.ids $L | tail -40

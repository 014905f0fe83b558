#!/bin/bash
# round 6, GPU call 1: where do k_expand's waves wait, and the first three experiments (tile = as many records as fill the apply rounds;
# compare-and-swap in flight with the home-slot load)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call1.log
: > $L
for v in diag1 diag2; do
  for wl in config2 readme; do
    VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$v.so timeout 300 python tools/wave_diag.py $wl >> $L 2>&1
  done
done
README_VARIANTS="base take232 speccas" timeout 1500 tools/ab_bench.sh base take232 take208 speccas >> $L 2>&1
tail -60 $L

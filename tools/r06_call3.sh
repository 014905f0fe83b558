#!/bin/bash
# round 6, GPU call 3: MachineLICM off (no VGPR spills), with / without phase clocks, five blocks per CU, sink-to-avoid-spills
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call3.log
: > $L
README_VARIANTS="base nomlicm nomlicm_noclk sink" timeout 2700 tools/ab_bench.sh base nomlicm nomlicm_noclk nomlicm_occ5 sink >> $L 2>&1
for wl in config2 readme; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_base.so timeout 300 python tools/phase_split.py $wl >> $L 2>&1
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_nomlicm.so timeout 300 python tools/phase_split.py $wl >> $L 2>&1
done
grep -v amdgpu.ids $L | cut -c1-600 | tail -40

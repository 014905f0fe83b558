#!/bin/bash
# round 6, GPU call 19: the round's final kernel under the diagnostics of calls 1-2: barrier waits per wave and per barrier, phase split (builds with -DVSR_WAVE_DIAG / -DVSR_PHASE_CLOCKS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call19.log
: > $L
for w in config2 readme; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_diag1.so timeout 600 python tools/wave_diag.py $w >> $L 2>gpurun_out/c19.err || tail -3 gpurun_out/c19.err >> $L
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_diag3.so timeout 600 python tools/wave_diag.py $w groups >> $L 2>gpurun_out/c19.err || tail -3 gpurun_out/c19.err >> $L
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_pc.so timeout 600 python tools/phase_split.py $w >> $L 2>gpurun_out/c19.err || tail -3 gpurun_out/c19.err >> $L
done
cut -c1-1500 $L

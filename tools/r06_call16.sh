#!/bin/bash
# round 6, GPU call 16: how much of a launch is the drain of the per-block statistics (atomics of every block into one line of the control block)?  timing only
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call16.log
: > $L
for v in base noflush base noflush; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$v.so timeout 600 python tools/time_levels.py 3 >> $L 2>gpurun_out/c16.err || tail -3 gpurun_out/c16.err >> $L
done
cut -c1-600 $L

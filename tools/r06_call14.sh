#!/bin/bash
# round 6, GPU call 14: the staging micro-benchmark's pipelined variants; the whole GPU suite on the build with the probe-only instantiation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call14.log
: > $L
timeout 600 python tools/bench_layout.py 24 20 > gpurun_out/r06_bench_layout.json 2>gpurun_out/r06_bench_layout.err
timeout 600 python tools/bench_layout.py 27 20 >> gpurun_out/r06_bench_layout.json 2>>gpurun_out/r06_bench_layout.err
cat gpurun_out/r06_bench_layout.json >> $L; tail -n 3 gpurun_out/r06_bench_layout.err >> $L
timeout 2400 python -m pytest tests -q -m gpu --durations=8 -x > gpurun_out/r06_gputests_call14.log 2>&1
tail -n 25 gpurun_out/r06_gputests_call14.log >> $L
grep -v amdgpu.ids $L | cut -c1-1500 | tail -60

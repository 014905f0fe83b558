#!/bin/bash
# A/B timing of experimental builds (tools/ab_build.sh) on the GPU box: tools/ab_run.sh NAME [NAME ...] -> k_expand ms per run, each
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
for name in "$@"; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$name.so python bench.py --no-verify --no-config3 --no-cpu-baseline --steps 5 --warmup 1 \
    2> gpurun_out/ab_$name.err | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('$name', 'round $round', 'k_expand ms/run', d['roofline']['kernel_ms_per_step']['k_expand'], 'ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'])"
done
done

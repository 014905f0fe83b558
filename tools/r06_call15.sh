#!/bin/bash
# round 6, GPU call 15: tiles drawn from the cursor in batches (VSR_TILE_BATCH 1 / 4 / 8 / 16): the staging micro-benchmark, config 2 and the README configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call15.log
: > $L
timeout 600 python tools/bench_layout.py 24 20 > gpurun_out/r06_bench_layout.json 2>gpurun_out/r06_bench_layout.err
timeout 600 python tools/bench_layout.py 27 20 >> gpurun_out/r06_bench_layout.json 2>>gpurun_out/r06_bench_layout.err
cat gpurun_out/r06_bench_layout.json >> $L; tail -n 3 gpurun_out/r06_bench_layout.err >> $L
README_VARIANTS="b1 b4 b8 b16" timeout 2400 tools/ab_bench.sh b1 b4 b8 b16 >> $L 2>&1
for v in b1 b4 b8 b16; do
  python - $v >> $L <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open('gpurun_out/ab_readme_%s.out' % v).readline())
    print(v, 'README per pass:', {k: x['kernel_ms'] for k, x in d['roofline']['per_pass'].items()})
except Exception as e:
    print(v, 'no JSON', e)
PY
done
grep -v amdgpu.ids $L | cut -c1-2500 | tail -60

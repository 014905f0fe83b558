#!/bin/bash
# A/B of the README configuration's whole run: tools/r03_readme_ab.sh <lib name> ...  (two rounds each)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
for round in 1 2; do
for v in "$@"; do
  VSRMC_LIB=$R/vsr_tlaplus_amd/ab/libvsrmc_$v.so timeout 400 python bench.py --workload readme --no-config2 --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/readme_$v.err | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); print('readme $v round $round', 'ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'], d['probe3'], d['roofline']['kernel_ms_per_step'])
except Exception as e:
    print('readme $v FAILED', l[:300])"
done
done

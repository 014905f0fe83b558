#!/bin/bash
# A/B of one build under two environments on the README configuration: tools/r03_env_ab.sh LABEL_A "ENV=.." LABEL_B "ENV=.."
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R; mkdir -p gpurun_out
for round in 1 2; do
  set -- "${@}"
  for ((i = 1; i <= $#; i += 2)); do
    v=${!i}; j=$((i + 1)); E=${!j}
    env $E timeout 400 python bench.py --workload readme --no-config2 --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/readme_$v.err | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); print('readme $v round $round', 'ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'], d['probe3'], d['roofline']['kernel_ms_per_step'])
except Exception as e:
    print('readme $v FAILED', l[:300])"
    tail -2 gpurun_out/readme_$v.err | grep -v amdgpu.ids | cut -c1-300
  done
done

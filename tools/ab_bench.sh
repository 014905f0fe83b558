#!/bin/bash
# A/B runs on the GPU box: tools/ab_bench.sh "<lib name>[:ENV=VAL,...]" ...  -> per variant k_expand ms per run on config 2 (two rounds)
# and, for names listed in README_VARIANTS, one README-configuration leg (bench.py asserts every level figure against the oracle fixture)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$name.so python bench.py --no-verify --no-config3 --no-cpu-baseline --steps 5 --warmup 1 \
    2> gpurun_out/ab_$name.err | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l)
    print('$spec', 'round $round', 'config2 k_expand ms/run', d['roofline']['kernel_ms_per_step']['k_expand'], 'ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'])
except Exception as e:
    print('$spec', 'FAILED', l[:200])"
  tail -2 gpurun_out/ab_$name.err | cut -c1-300
done
done
for name in $README_VARIANTS; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$name.so python bench.py --no-verify --workload readme --no-cpu-baseline --steps 3 --warmup 1 \
    2> gpurun_out/ab_readme_$name.err | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l)
    print('$name', 'README ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_per_step'], 'materialised', d.get('materialised'), 'deep_passes', d.get('deep_passes'))
except Exception as e:
    print('$name', 'README FAILED', l[:200])"
  tail -2 gpurun_out/ab_readme_$name.err | cut -c1-300
done

#!/bin/bash
# A/B runs on the GPU box: tools/ab_bench.sh "<lib name>[:ENV=VAL,...]" ...  -> per variant k_expand ms per run on config 2 (two rounds)
# and, for names listed in README_VARIANTS, one README-configuration leg (bench.py asserts every level figure against the oracle fixture).
# A line says FAILED only when bench.py itself failed (non-zero exit: an assertion against the fixture, a device error); a line that ran but lacks a
# field prints what it has.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
show() {   # label, exit code of bench.py; the JSON line on stdin
  python -c "
import sys, json
label, rc = sys.argv[1], int(sys.argv[2])
l = sys.stdin.readline()
if rc != 0:
    print(label, 'FAILED (bench.py exit %d)' % rc, l[:200]); sys.exit(0)
try:
    d = json.loads(l)
except Exception as e:
    print(label, 'ran (exit 0) but printed no JSON line:', l[:200]); sys.exit(0)
r = d.get('roofline') or {}
print(label, 'ms_per_step', d.get('ms_per_step'), 'value %.4g' % d.get('value', 0), 'kernel_ms', r.get('kernel_ms_per_step'),
      'launches', r.get('launches_per_step'), 'materialised', d.get('materialised'), 'deep_passes', d.get('deep_passes'))" "$1" "$2"
}
for round in 1 2; do
for spec in "$@"; do
  name=${spec%%:*}; envs=""
  if [[ "$spec" == *:* ]]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
  env $envs VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$name.so python bench.py --no-verify --no-config3 --no-cpu-baseline --steps 5 --warmup 1 \
    > gpurun_out/ab_$name.out 2> gpurun_out/ab_$name.err
  show "$spec round $round config2" $? < gpurun_out/ab_$name.out
  tail -2 gpurun_out/ab_$name.err | cut -c1-300
done
done
for name in $README_VARIANTS; do
  VSRMC_LIB=$PWD/vsr_tlaplus_amd/ab/libvsrmc_$name.so python bench.py --no-verify --workload readme --no-cpu-baseline --steps 3 --warmup 1 \
    > gpurun_out/ab_readme_$name.out 2> gpurun_out/ab_readme_$name.err
  show "$name README" $? < gpurun_out/ab_readme_$name.out
  tail -2 gpurun_out/ab_readme_$name.err | cut -c1-300
done

#!/bin/bash
# round 6, GPU calls 12-13: the probe-only instantiation (k_expand<.., 6> + k_probe_resolve): the deep-search / probe tests, then the README configuration with and without it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call13.log
: > $L
timeout 1500 python -m pytest tests/test_probe_footprint.py tests/test_deep_search.py tests/test_config3_trace.py -q -m gpu -x --durations=5 > gpurun_out/r06_gputests_call13.log 2>&1
tail -n 15 gpurun_out/r06_gputests_call13.log >> $L
timeout 900 python -m pytest tests/test_sharded_gloo.py -q -m gpu -k "readme_configuration_on_two_ranks" > gpurun_out/r06_gputests_call13b.log 2>&1
grep -E "AssertionError|passed|failed" gpurun_out/r06_gputests_call13b.log | cut -c1-900 >> $L
for v in "" "VSRMC_NO_PROBE_KERNEL=1" "" "VSRMC_NO_PROBE_KERNEL=1"; do
  env $v python bench.py --no-verify --workload readme --no-cpu-baseline --no-config5 --steps 3 --warmup 1 > gpurun_out/c13.out 2> gpurun_out/c13.err
  echo "README [$v] rc=$?" >> $L
  python - >> $L <<'PY'
import json
try:
    d = json.loads(open('gpurun_out/c13.out').readline())
    r = d['roofline']
    print(' ms_per_step', d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_per_step'))
    for k, v in (r.get('per_pass') or {}).items(): print('   ', k, v)
except Exception as e:
    print(' no JSON:', e, open('gpurun_out/c13.err').read()[-600:])
PY
done
grep -v amdgpu.ids $L | cut -c1-400 | tail -70

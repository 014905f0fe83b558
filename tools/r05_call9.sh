#!/bin/bash
# round-5 GPU call 9: k_expand with ONE mode compiled in (regeneration by the claim bitmap, virtual level, probe) against the run-time-switched instantiation
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_deep_search.py tests/test_gpu_parity.py -q -m gpu -k "deep or probe2 or probe3 or rebas or checkpoint or overflow or whole_workload" 2>&1 | tail -n 6 > gpurun_out/r05_t9.log
tail -n 3 gpurun_out/r05_t9.log
for v in off on off on; do
  if [ $v = on ]; then unset VSRMC_NO_MODE_KERNELS; else export VSRMC_NO_MODE_KERNELS=1; fi
  python bench.py --workload readme --steps 3 --warmup 1 --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('one-mode kernels $v:', 'README ms_per_step', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], [(p['level'], p['k_expand_ms'], p['regenerate_ms'], p.get('probe_ms')) for p in d['deep_passes']])"
done
unset VSRMC_NO_MODE_KERNELS
python bench.py --workload config2 --steps 5 --warmup 1 --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config2 k_expand ms/run', d['roofline']['kernel_ms_per_step']['k_expand'], 'ms_per_step', d['ms_per_step'])"
timeout 300 python tools/run_config5.py | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config5', d['stop'][:30], d['depth'], d['distinct'], d['seconds'], d['probed'])"

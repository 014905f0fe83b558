#!/bin/bash
# round-5 GPU call 6: pooled scratch buffers of the deep search — tests, README bench, model 2 to exhaustion (one seed), config 5
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_deep_search.py tests/test_sharded_gloo.py -q -m gpu -k "deep or readme or rebas or checkpoint or overflow or probe" 2>&1 | tail -n 6 > gpurun_out/r05_t8.log
tail -n 3 gpurun_out/r05_t8.log
python bench.py --workload readme --steps 3 --warmup 1 --no-cpu-baseline 2>gpurun_out/r05_b8.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('README ms/step', d['ms_per_step'], d['roofline']['kernel_ms_per_step'], d['roofline']['launches'], d['deep_passes'])"
tail -n 2 gpurun_out/r05_b8.err | cut -c1-300
timeout 600 python tools/run_models_deep.py --max-seconds 500 --models model2 --one-seed 2>&1 | cut -c1-400 | head -3
timeout 300 python tools/run_config5.py | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('config5', d['stop'][:30], d['depth'], d['distinct'], d['seconds'], d['probed'])"

#!/bin/bash
# round-5 GPU call 2: the sharded paths with the generator-side winner set (exchange-free regeneration), the fixed CLI tests, then the two analysis
# models to exhaustion under BOTH fingerprint seeds (the re-basing makes that affordable) and config 5's probed level 15 under two seeds
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_sharded_gloo.py -q -m gpu 2>&1 | tail -40 > gpurun_out/r05_t4.log
python -m pytest tests/test_gpu_parity.py -q -k "cli_" 2>&1 | tail -15 > gpurun_out/r05_t5.log
tail -3 gpurun_out/r05_t4.log gpurun_out/r05_t5.log
timeout 1500 python tools/run_models_deep.py --max-seconds 1200 > gpurun_out/r05_models_exhausted_two_seeds.jsonl 2> gpurun_out/r05_models.err
tail -c 1500 gpurun_out/r05_models_exhausted_two_seeds.jsonl; tail -3 gpurun_out/r05_models.err
timeout 300 python tools/run_config5.py > gpurun_out/r05_config5_seed0.json 2> gpurun_out/r05_config5.err
timeout 300 python tools/run_config5.py --seed 0x5EED5EED5EED5EED > gpurun_out/r05_config5_seed1.json 2>> gpurun_out/r05_config5.err
python - <<'PY'
import json
for f in ("gpurun_out/r05_config5_seed0.json", "gpurun_out/r05_config5_seed1.json"):
    try:
        d = json.load(open(f))
        print(f, d["fp_seed"], d["stop"][:40], d["depth"], d["distinct"], d["seconds"], d["probed"])
    except Exception as e:
        print(f, "FAILED", e)
PY

#!/bin/bash
# round-5 GPU call 1: the new tests, then the record-size sensitivity A/B (base vs. 6 dead words per record)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -m pytest tests/test_deep_search.py -q 2>&1 | tail -40 > gpurun_out/r05_t1.log
python -m pytest tests/test_gpu_parity.py -q -k "whole_workload or cli_ or checkpoint or small_record or probe_after or seen_set_and_frontier" 2>&1 | tail -40 > gpurun_out/r05_t2.log
python -m pytest tests/test_sharded_gloo.py -q -m gpu -k "config4 or violation_of_any_mask" 2>&1 | tail -40 > gpurun_out/r05_t3.log
README_VARIANTS="base pad6" bash tools/ab_bench.sh base pad6 > gpurun_out/r05_ab_pad.log 2>&1
tail -5 gpurun_out/r05_t1.log gpurun_out/r05_t2.log gpurun_out/r05_t3.log; cat gpurun_out/r05_ab_pad.log

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
mkdir -p $OUT
timeout 600 python -m pytest tests/test_tlc_fp64.py -m gpu -q 2>&1 | tail -15
timeout 300 python tools/tlc_fp_rate.py 22 > $OUT/r03_tlc_fp_rate.json 2> $OUT/r03_tlc_fp_rate.err; cat $OUT/r03_tlc_fp_rate.json; tail -3 $OUT/r03_tlc_fp_rate.err

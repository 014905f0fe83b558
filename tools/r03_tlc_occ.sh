#!/bin/bash
# round 3, second session: the TLC-fingerprint mode on the GPU (tests + rate) and the occupancy-5 A/B of k_expand (96 VGPRs, 768-entry work list)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
mkdir -p $OUT
timeout 600 python -m pytest tests/test_tlc_fp64.py -m gpu -x -q 2>&1 | tail -15
timeout 300 python tools/tlc_fp_rate.py 22 > $OUT/r03_tlc_fp_rate.json 2> $OUT/r03_tlc_fp_rate.err; cat $OUT/r03_tlc_fp_rate.json; tail -3 $OUT/r03_tlc_fp_rate.err
tools/r03_ab.sh base occ5 cc768 2>&1 | grep -v amdgpu.ids
for v in base occ5; do
  VSRMC_LIB=$R/vsr_tlaplus_amd/ab/libvsrmc_$v.so timeout 400 python bench.py --workload readme --no-config2 --no-cpu-baseline --steps 3 --warmup 1 2> $OUT/readme_$v.err | python -c "
import sys, json
l = sys.stdin.readline()
try:
    d = json.loads(l); print('readme $v', 'ms_per_step', d['ms_per_step'], 'value %.4g' % d['value'], d['roofline']['kernel_ms_per_step'])
except Exception as e:
    print('readme $v FAILED', l[:300])"
  tail -2 $OUT/readme_$v.err | cut -c1-300
done

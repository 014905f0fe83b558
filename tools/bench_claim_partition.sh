#!/bin/bash
# builds tools/bench_claim_partition.hip on the GPU box and runs it at config 2's and README's final table loads; then the same binary under rocprofv3 for
# the fabric traffic of each kernel (separate --pmc passes).  -> gpurun_out/r06_claim_partition.json, r06_claim_partition_{rd,wr}.md
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bench_claim_partition $R/tools/bench_claim_partition.hip || exit 1
: > $OUT/r06_claim_partition.json
/tmp/bench_claim_partition 31 0.15 28 >> $OUT/r06_claim_partition.json
/tmp/bench_claim_partition 32 0.42 28 >> $OUT/r06_claim_partition.json
cat $OUT/r06_claim_partition.json
cd /tmp && export TMPDIR=/tmp
for pass in rd wr; do
  if [ $pass = rd ]; then PMC="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; else PMC="TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"; fi
  rm -rf $OUT/prof_cp_$pass
  timeout 600 rocprofv3 --kernel-trace --pmc $PMC -d $OUT/prof_cp_$pass -o cp_$pass -- /tmp/bench_claim_partition 31 0.15 28 > $OUT/prof_cp_$pass.log 2>&1
  DB=$(find $OUT/prof_cp_$pass -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/prof_summary.py "$DB" "r06 claim_partition $pass: rocprofv3 --kernel-trace --pmc $PMC -- bench_claim_partition 31 0.15 28" > $OUT/r06_claim_partition_$pass.md; fi
  rm -rf $OUT/prof_cp_$pass
  head -30 $OUT/r06_claim_partition_$pass.md
done

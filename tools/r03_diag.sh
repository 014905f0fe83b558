#!/bin/bash
# round-3 diagnostics on the GPU box: occupancy sweep, marginal-cost builds, TCC / TCP counter passes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
echo "== occupancy sweep (resident blocks per CU)"
for b in 1 2 3 4; do
  VSRMC_MAX_BPC=$b VSRMC_LIB=$R/vsr_tlaplus_amd/ab/libvsrmc_e2.so python bench.py --no-verify --no-config3 --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); print('bpc $b k_expand ms/run', d['roofline']['kernel_ms_per_step']['k_expand'])"
done
echo "== marginal costs"
tools/r03_ab.sh e2 dblprobe dblwrite 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/r03_counters_list.txt 2>&1
PROF="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config3 --no-verify"
export VSRMC_LIB=$R/vsr_tlaplus_amd/ab/libvsrmc_e2.so
one() {
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/prof_$name -o $name -- $PROF > $OUT/prof_$name.log 2>&1
  local db=$(find $OUT/prof_$name -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/prof_summary.py "$db" "r03 diag $name: --pmc $*" | grep -E "k_expand|^#" | head -12 > $OUT/r03_diag_$name.md; else tail -5 $OUT/prof_$name.log > $OUT/r03_diag_$name.md; fi
  rm -rf $OUT/prof_$name
  cat $OUT/r03_diag_$name.md
}
one tcc1 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum
one tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
one tcc3 TCC_BUSY_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_ATOMIC_sum
one tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum
one tcp2 TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum
one ta1 TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum
one sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM
one sq2 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE

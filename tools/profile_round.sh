#!/bin/bash
# The round's measurement set, run on the GPU box through gpurun:  tools/profile_round.sh r02
# -> gpurun_out/<tag>_bench.json (driver-style bench line), <tag>_kernel_stats.md (rocprofv3 --kernel-trace --stats),
#    <tag>_pmc_fetch.md / _pmc_write.md (separate --pmc passes), <tag>_sq_counters_{a,b}.md, <tag>_config3_bench.json,
#    <tag>_config3_kernel_stats.md.  Copy what is to be judged into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PROF="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-config3 --no-verify"
python $R/bench.py --steps 10 --warmup 2 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 600 $OUT/${TAG}_bench.json
one() {   # name, rocprofv3 flags...
  local name=$1; shift
  rm -rf $OUT/prof_$name
  timeout 300 rocprofv3 "$@" -d $OUT/prof_$name -o $name -- $PROF > $OUT/prof_$name.log 2>&1
  local db=$(find $OUT/prof_$name -name "*_results.db" | head -1)
  if [ -n "$db" ]; then python $R/tools/prof_summary.py "$db" "$TAG $name: rocprofv3 $* -- $PROF" > $OUT/${TAG}_$name.md; fi
  rm -rf $OUT/prof_$name
  head -8 $OUT/${TAG}_$name.md
}
# the README defect configuration to its depth-24 violation (levels 1-21 stored, 22 virtual, 23 streamed, 24 probed), alone and traced
python $R/bench.py --workload config3 > $OUT/${TAG}_config3_bench.json 2> $OUT/${TAG}_config3_bench.err
tail -c 400 $OUT/${TAG}_config3_bench.json
PROF_SAVE=$PROF
PROF="python $R/bench.py --workload config3"
one config3_kernel_stats --kernel-trace --stats
PROF=$PROF_SAVE
one kernel_stats --kernel-trace --stats
one pmc_fetch --kernel-trace --pmc FETCH_SIZE
one pmc_write --kernel-trace --pmc WRITE_SIZE
one sq_counters_a --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
one sq_counters_b --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU

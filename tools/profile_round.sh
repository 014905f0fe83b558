#!/bin/bash
# The round's measurement set, run on the GPU box through gpurun:  tools/profile_round.sh r03
# -> gpurun_out/<tag>_bench.json          the driver-style bench line (README configuration = headline, config2 object, cpu_baseline)
#    <tag>_{config2,readme}_kernel_stats.md  rocprofv3 --kernel-trace --stats of one run of each workload
#    <tag>_{config2,readme}_pmc_{rd,wr}.md   TCC fabric requests by size (separate --pmc passes) and <tag>_*_traffic.json from them
#    <tag>_sq_counters_{a,b}.md              SQ counters of the config-2 run
# Copy what is to be judged into profiles/.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
one() {   # name, workload flags, rocprofv3 flags...
  local name=$1 wl=$2; shift 2
  local PROF="python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify $wl"
  rm -rf $OUT/prof_$name
  timeout 600 rocprofv3 "$@" -d $OUT/prof_$name -o $name -- $PROF > $OUT/prof_$name.log 2>&1
  DB=$(find $OUT/prof_$name -name "*_results.db" | head -1)
  if [ -n "$DB" ]; then python $R/tools/prof_summary.py "$DB" "$TAG $name: rocprofv3 $* -- $PROF" > $OUT/${TAG}_$name.md; cp "$DB" $OUT/${TAG}_$name.db; fi
  rm -rf $OUT/prof_$name
  head -7 $OUT/${TAG}_$name.md
}
for W in config2 readme; do
  FL="--workload $W"
  one ${W}_kernel_stats "$FL" --kernel-trace --stats
  one ${W}_pmc_rd "$FL" --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum
  one ${W}_pmc_wr "$FL" --kernel-trace --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum
  python $R/tools/make_traffic.py $TAG $W $OUT/${TAG}_${W}_pmc_rd.db $OUT/${TAG}_${W}_pmc_wr.db "bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify $FL" > $OUT/${TAG}_${W}_traffic.json
  cat $OUT/${TAG}_${W}_traffic.json
  cp $OUT/${TAG}_${W}_traffic.json $R/profiles/${TAG}_${W}_traffic.json     # on the box: the bench line below reads the traffic of THIS build (same source hash)
done
rm -f $OUT/${TAG}_*.db
one sq_counters_a "--workload config2" --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
one sq_counters_b "--workload config2" --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
# the same two passes over the README run: one row per INSTANTIATION of k_expand (<313,1> stored levels and the level inserted beyond the buffers, <313,4> the
# virtual level, <313,3> its regeneration from the claim bitmap, <313,6> the probe) — 45 % of that run is not the plain instantiation
one sq_counters_readme_a "--workload readme" --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
one sq_counters_readme_b "--workload readme" --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
# registers, spills, scratch and code size of every instantiation in the shipped library (from the code object's notes)
$R/tools/kernel_resources.sh $R/vsr_tlaplus_amd/libvsrmc.so k_expandILb1ELi31 > $OUT/${TAG}_kernel_resources.txt 2>&1
rm -f $OUT/${TAG}_*.db
# the driver-style line last: its roofline.traffic comes from the request counters measured above on the same sources
python $R/bench.py --steps 5 --warmup 1 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
tail -c 1200 $OUT/${TAG}_bench.json

#!/bin/bash
# round-3 closing measurements on the GPU box: sharded leg at world 1 (native loop with / without issuing the collectives, Python loop),
# two more A/B rows (wave-level compaction of the outputs, the duplicate filter with its fabric traffic), then the round's profile set
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
cd $R
shard1() {   # label, env...
  local label=$1; shift
  env "$@" VSR_BENCH_SHARDED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29711 \
    bench.py --gpus 1 --steps 5 --warmup 1 2>$OUT/r03_shard1_$label.err | python -c "
import sys, json
l = [x for x in sys.stdin if x.startswith('{')]
d = json.loads(l[-1]) if l else {}
print('world-1 sharded leg, $label:', 'ms_per_step', d.get('ms_per_step'), d.get('roofline', {}).get('kernel_ms_per_step'), '|', d.get('config', {}).get('level_loop'))"
}
shard1 native_no_collectives A=1
shard1 native_rccl_collectives VSRMC_COMM_ALWAYS_CALL=1
shard1 python_loop VSR_BENCH_PYLOOP=1
cp vsr_tlaplus_amd/libvsrmc.so vsr_tlaplus_amd/ab/libvsrmc_fin.so
tools/r03_ab.sh fin wc dd1 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
for v in fin dd1; do
  for pass in "rd TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "wr TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"; do
    set -- $pass; name=$1; shift
    rm -rf $OUT/prof_x
    VSRMC_LIB=$R/vsr_tlaplus_amd/ab/libvsrmc_$v.so timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $OUT/prof_x -o x -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify --workload config2 > $OUT/prof_x.log 2>&1
    DB=$(find $OUT/prof_x -name "*_results.db" | head -1)
    [ -n "$DB" ] && cp "$DB" $OUT/r03_dedup_${v}_$name.db
    rm -rf $OUT/prof_x
  done
  python $R/tools/make_traffic.py r03 config2-$v $OUT/r03_dedup_${v}_rd.db $OUT/r03_dedup_${v}_wr.db "bench.py --workload config2 (VSRMC_LIB=libvsrmc_$v.so)" > $OUT/r03_dedup_${v}_traffic.json
  python -c "
import json; d = json.load(open('$OUT/r03_dedup_${v}_traffic.json')); print('$v', 'read GB', d['read_bytes'] / 1e9, 'write GB', d['write_bytes'] / 1e9, 'atomics', d['atomics_to_fabric'], 'rd128', d['rdreq']['n128'], 'rd64', d['rdreq']['n64'])"
done
rm -f $OUT/r03_dedup_*.db
$R/tools/profile_round.sh r03

#!/bin/bash
# round 6, GPU call 9: redo over a list, the overlapped exchange (small forced cases; the README configuration on two ranks with and without it), A/B of two barriers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
L=gpurun_out/r06_call9.log
: > $L
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_gloo.py -q -m gpu -k "overflows_the_work_list or overlapped_exchange or sharded_hip_engine_on_one_gpu or violation_of_any_mask or checkpointed_and_recovered or sharded_cli" > gpurun_out/r06_gputests_call9.log 2>&1
tail -n 15 gpurun_out/r06_gputests_call9.log >> $L
for ov in 0 1; do
  echo "== README configuration, two ranks on this GPU, VSRMC_OVERLAP=$ov" >> $L
  ( time VSRMC_OVERLAP=$ov SHARD_VERBOSE=1 OMP_NUM_THREADS=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2979$ov \
      tests/shard_deep_worker.py 3 1 3 3 1 23 gpurun_out/r06_readme_w2_ov$ov 0 0 ) 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" | tail -n 40 >> $L
done
timeout 900 tools/ab_bench.sh prod p_dn >> $L 2>&1
grep -v amdgpu.ids $L | cut -c1-330 | tail -90

#!/bin/bash
# debugging aid: one sharded worker run with the checkpoint + probe legs, full log kept
cd /root/repo
mkdir -p gpurun_out
SHARD_INV_MASK=2 SHARD_CHECKPOINT_AT=9 SHARD_PROBE_AT=19 OMP_NUM_THREADS=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29699 tests/shard_worker.py hip 3 1 2 1 18 /tmp/shdbg 0 > gpurun_out/shdbg.log 2>&1
grep -n "Error\|error\|File \|line " gpurun_out/shdbg.log | head -40

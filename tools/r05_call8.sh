#!/bin/bash
# round-5 GPU call 8: the seen-set in uncached / fine-grained memory (VSRMC_TABLE_MEM) against the default, config 2 and the README configuration
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for mem in default uncached finegrained default uncached; do
  if [ $mem = default ]; then unset VSRMC_TABLE_MEM; else export VSRMC_TABLE_MEM=$mem; fi
  python bench.py --workload config2 --steps 5 --warmup 1 --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mem', 'config2 k_expand ms/run', d['roofline']['kernel_ms_per_step']['k_expand'], 'ms_per_step', d['ms_per_step'])"
done
for mem in default uncached finegrained; do
  if [ $mem = default ]; then unset VSRMC_TABLE_MEM; else export VSRMC_TABLE_MEM=$mem; fi
  python bench.py --workload readme --steps 2 --warmup 1 --no-verify --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$mem', 'README ms_per_step', d['ms_per_step'], d['roofline']['kernel_ms_per_step'])"
done

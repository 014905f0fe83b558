#!/bin/bash
# Experimental builds of libvsrmc.so for A/B timing on the GPU box: tools/ab_build.sh NAME "-DFLAG=.. -DFLAG2=.." [NAME2 "..."] ...
# -> vsr_tlaplus_amd/ab/libvsrmc_NAME.so (git-ignored like every .so; travels with gpurun).  Select with VSRMC_LIB=<path>.
set -e
cd "$(dirname "$0")/.."
mkdir -p vsr_tlaplus_amd/ab
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared $flags -o vsr_tlaplus_amd/ab/libvsrmc_$name.so vsr_tlaplus_amd/csrc/vsrmc.hip &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
ls -la vsr_tlaplus_amd/ab/

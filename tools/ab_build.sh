#!/bin/bash
# Experimental builds of libvsrmc.so for A/B timing on the GPU box: tools/ab_build.sh NAME "-DFLAG=.. -DFLAG2=.." [NAME2 "..."] ...
# (the product's flags, vsr_tlaplus_amd/build.py, plus the variant's)
# -> vsr_tlaplus_amd/ab/libvsrmc_NAME.so (git-ignored like every .so; travels with gpurun).  Select with VSRMC_LIB=<path>.
# The sources are SNAPSHOT first (csrc + include into a temporary directory): hipcc maps the files it compiles, and an edit of a header while a
# six-minute compile runs ends it with a bus error — or, worse, builds a mixture.
set -e
cd "$(dirname "$0")/.."
mkdir -p vsr_tlaplus_amd/ab
SNAP=$(mktemp -d /tmp/vsrmc_ab.XXXXXX)
mkdir -p $SNAP/vsr_tlaplus_amd $SNAP/include
cp -r vsr_tlaplus_amd/csrc $SNAP/vsr_tlaplus_amd/csrc
cp include/vsrmc.h $SNAP/include/
pids=()
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -shared $flags -o vsr_tlaplus_amd/ab/libvsrmc_$name.so $SNAP/vsr_tlaplus_amd/csrc/vsrmc.hip &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=1; done
rm -rf $SNAP
ls -la vsr_tlaplus_amd/ab/
exit $rc

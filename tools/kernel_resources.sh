#!/bin/bash
# tools/kernel_resources.sh LIB.so [filter] -> VGPRs, spills, scratch, LDS and code bytes of every kernel in the library (from the code object's notes)
set -e
LIB=$(readlink -f "$1"); FILTER=${2:-k_expand}
T=$(mktemp -d); cd $T
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=<(objcopy -O binary --only-section=.hip_fatbin "$LIB" /dev/stdout) --output=co.o --unbundle 2>/dev/null || {
  objcopy -O binary --only-section=.hip_fatbin "$LIB" fat.bin
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=fat.bin --output=co.o --unbundle
}
/opt/rocm/lib/llvm/bin/llvm-readelf --notes co.o | python3 -c "
import sys, re
txt = sys.stdin.read()
blocks = re.split(r'\n\s+- ', txt)
for b in blocks:
    m = re.search(r'\.name:\s+(\S+)', b)
    if not m or '$FILTER' not in m.group(1): continue
    g = lambda k: (re.search(r'\.' + k + r':\s+(\d+)', b) or [None, '?'])[1]
    print(m.group(1)[:110], 'vgpr', g('vgpr_count'), 'agpr', g('agpr_count'), 'sgpr', g('sgpr_count'), 'spill_v', g('vgpr_spill_count'), 'spill_s', g('sgpr_spill_count'), 'scratch', g('private_segment_fixed_size'), 'lds', g('group_segment_fixed_size'))
"
/opt/rocm/lib/llvm/bin/llvm-readelf -s co.o | awk -v f="$FILTER" '$4=="FUNC" && index($8,f) {print $3, $8}' | sort -k2 | cut -c1-130
rm -rf $T

#!/bin/bash
# tools/kernel_resources.sh <object.o> [name-substring]: VGPRs / spills / scratch / LDS of the gfx950 kernels of one object (from its metadata notes)
set -euo pipefail
O=$(realpath $1); K=${2:-}; T=$(mktemp -d)
cp $O $T/x.o; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null)
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g=lambda k: (re.search(r'\.'+k+r':\s*(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if '$K' in name: print(f\"{name[:110]:110s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} sgpr {g('sgpr_count'):>4s}\")
"
rm -rf $T

#!/bin/bash
# tools/isa_hash.sh <object.o> <kernel-name-substring>: md5 of the gfx950 instruction stream of one kernel (addresses stripped)
# -- used to check that a refactoring leaves a tuned instance's code untouched
set -euo pipefail
O=$(realpath $1); K=$2; T=$(mktemp -d)
cp $O $T/x.o; (cd $T && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading x.o > /dev/null)
/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/x.o.0.hipv4-amdgcn-amd-amdhsa--gfx950 | awk -v k="$K" '/^[0-9a-f]+ <.*>:/{p=index($0,k)>0} p' | sed -E 's/\/\/ [0-9A-F]+:.*$//; s/^[0-9a-f]+ //' > $T/k.s
wc -l < $T/k.s; md5sum < $T/k.s; rm -rf $T

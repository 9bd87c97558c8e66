#!/bin/bash
# Experiment builds: tools/build_variant.sh <name> <source-stem> "<extra flags>"  ->  egc_amd/lib/var_<name>.so
# (the named source recompiled with the flags, every other object as built by egc_amd/csrc/build.sh); run a script
# against it with EGC_HIP_LIB=egc_amd/lib/var_<name>.so.  The variants are git-ignored.
set -euo pipefail
ROOT="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
H=$ROOT/egc_amd/csrc
N=$1; S=$2; F=${3:-}
mkdir -p $H/obj/var_$N
extra=""
{ [ "$S" = egc_gemm_f16x2 ] || [ "$S" = egc_gemm_f16x2k ]; } && extra="-fno-slp-vectorize"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$H -Wall -Wno-unused-function -Wno-pass-failed $extra $F -c $H/$S.hip -o $H/obj/var_$N/$S.o
objs=""
for o in $H/obj/*.o; do b=$(basename $o .o); if [ "$b" = "$S" ]; then objs="$objs $H/obj/var_$N/$S.o"; else objs="$objs $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/egc_amd/lib/var_$N.so $objs
echo built egc_amd/lib/var_$N.so

#!/bin/bash
# Build libegc_hip.so (gfx950 only) in-tree: egc_amd/lib/libegc_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/egc_amd/lib"
mkdir -p "$OUT" "$HERE/obj"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$HERE -Wall -Wno-unused-function -Wno-pass-failed ${EGC_EXTRA_FLAGS:-}"
pids=()
for src in egc_graph egc_gemm egc_gemm_bf16x3 egc_gemm_f16x2 egc_gemm_f16x2k egc_gemm_f16x2w egc_gemm_xt egc_aggregate egc_aggregate_fast egc_aggregate_tile egc_fused_tile egc_aggregate_fusedw egc_backward egc_tail; do
  if [ ! -f "$HERE/obj/$src.o" ] || [ "$HERE/$src.hip" -nt "$HERE/obj/$src.o" ] || [ "$HERE/egc_common.h" -nt "$HERE/obj/$src.o" ] || [ "$HERE/egc_gemm_split.h" -nt "$HERE/obj/$src.o" ] || [ "$HERE/egc_aggregate_dev.h" -nt "$HERE/obj/$src.o" ] || [ "$HERE/egc_aggregate_fast_dev.h" -nt "$HERE/obj/$src.o" ] || [ "$ROOT/include/egc_hip.h" -nt "$HERE/obj/$src.o" ]; then
    extra=""
    # packed-f32 VALU next to MFMAs costs more issue cycles than two scalar operations (egc_gemm_f16x2.hip header)
    { [ "$src" = egc_gemm_f16x2 ] || [ "$src" = egc_gemm_f16x2k ] || [ "$src" = egc_gemm_f16x2w ]; } && extra="-fno-slp-vectorize"
    $HIPCC $FLAGS $extra -c "$HERE/$src.hip" -o "$HERE/obj/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libegc_hip.so" "$HERE/obj/egc_graph.o" "$HERE/obj/egc_gemm.o" "$HERE/obj/egc_gemm_bf16x3.o" "$HERE/obj/egc_gemm_f16x2.o" "$HERE/obj/egc_gemm_f16x2k.o" "$HERE/obj/egc_gemm_f16x2w.o" "$HERE/obj/egc_gemm_xt.o" "$HERE/obj/egc_aggregate.o" "$HERE/obj/egc_aggregate_fast.o" "$HERE/obj/egc_aggregate_tile.o" "$HERE/obj/egc_fused_tile.o" "$HERE/obj/egc_aggregate_fusedw.o" "$HERE/obj/egc_backward.o" "$HERE/obj/egc_tail.o"
echo "built $OUT/libegc_hip.so"

#!/bin/bash
# Build libegc_hip.so (gfx950 only) in-tree: egc_amd/lib/libegc_hip.so
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/egc_amd/lib"
OBJ="$HERE/obj"
[ -n "${EGC_EXTRA_FLAGS:-}" ] && OBJ="$OBJ/dbg_$(echo "$EGC_EXTRA_FLAGS" | tr -c "A-Za-z0-9\n" "_")"   # a diagnostic build never shares objects with the plain one
mkdir -p "$OUT" "$OBJ"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -I$ROOT/include -I$HERE -Wall -Wno-unused-function -Wno-pass-failed ${EGC_EXTRA_FLAGS:-}"
pids=()
for src in egc_graph egc_gemm egc_gemm_bf16x3 egc_gemm_f16x2 egc_gemm_f16x2k egc_gemm_xt egc_aggregate egc_aggregate_fast egc_aggregate_tile egc_fused_tile egc_fused_tile_wide1 egc_fused_tile_wide2 egc_fused_tile_wide3 egc_backward egc_tail; do
  if [ ! -f "$OBJ/$src.o" ] || [ "$HERE/$src.hip" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_common.h" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_gemm_split.h" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_aggregate_dev.h" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_aggregate_fast_dev.h" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_fused_tile_dev.h" -nt "$OBJ/$src.o" ] || [ "$HERE/egc_fused_tile_wide.inc" -nt "$OBJ/$src.o" ] || [ "$ROOT/include/egc_hip.h" -nt "$OBJ/$src.o" ]; then
    extra=""
    # packed-f32 VALU next to MFMAs costs more issue cycles than two scalar operations (egc_gemm_f16x2.hip header)
    { [ "$src" = egc_gemm_f16x2 ] || [ "$src" = egc_gemm_f16x2k ]; } && extra="-fno-slp-vectorize"
    $HIPCC $FLAGS $extra -c "$HERE/$src.hip" -o "$OBJ/$src.o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT/libegc_hip.so" "$OBJ/egc_graph.o" "$OBJ/egc_gemm.o" "$OBJ/egc_gemm_bf16x3.o" "$OBJ/egc_gemm_f16x2.o" "$OBJ/egc_gemm_f16x2k.o" "$OBJ/egc_gemm_xt.o" "$OBJ/egc_aggregate.o" "$OBJ/egc_aggregate_fast.o" "$OBJ/egc_aggregate_tile.o" "$OBJ/egc_fused_tile.o" "$OBJ/egc_fused_tile_wide1.o" "$OBJ/egc_fused_tile_wide2.o" "$OBJ/egc_fused_tile_wide3.o" "$OBJ/egc_backward.o" "$OBJ/egc_tail.o"
echo "built $OUT/libegc_hip.so"

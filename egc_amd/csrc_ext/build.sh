#!/bin/bash
# Build egc_amd/lib/libegc_torch_ext.so: the TORCH_LIBRARY binding over libegc_hip.so (host C++ only, no device code).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
OUT="$ROOT/egc_amd/lib"
SRC="$HERE/egc_torch_ext.cpp"
if [ -f "$OUT/libegc_torch_ext.so" ] && [ "$OUT/libegc_torch_ext.so" -nt "$SRC" ] && [ "$OUT/libegc_torch_ext.so" -nt "$ROOT/include/egc_hip.h" ]; then
  echo "up to date: $OUT/libegc_torch_ext.so"; exit 0
fi
INCS="$(python3 -c 'from torch.utils.cpp_extension import include_paths; print(" ".join("-I" + p for p in include_paths()))')"
ABI="$(python3 -c 'import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))')"
TLIB="$(python3 -c 'import torch, os; print(os.path.join(os.path.dirname(torch.__file__), "lib"))')"
g++ -O2 -fPIC -shared -std=c++17 -D_GLIBCXX_USE_CXX11_ABI=$ABI $INCS -I"$ROOT/include" "$SRC" \
    -L"$TLIB" -ltorch -ltorch_cpu -lc10 -L"$OUT" -legc_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TLIB" -o "$OUT/libegc_torch_ext.so"
echo "built $OUT/libegc_torch_ext.so"

#!/bin/bash
# Build the stand-alone C++ check of the C ABI against the in-tree libegc_hip.so (host code only).
set -euo pipefail
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
mkdir -p "$HERE/_build"
"${HIPCC:-/opt/rocm/bin/hipcc}" -O2 -std=c++17 -I"$ROOT/include" -o "$HERE/_build/c_abi_check" "$HERE/c_abi_check.cpp" \
  -L"$ROOT/egc_amd/lib" -legc_hip -Wl,-rpath,'$ORIGIN/../../../egc_amd/lib'
echo "built $HERE/_build/c_abi_check"

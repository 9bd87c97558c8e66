"""The C ABI used from a plain C++ host (tests/c_abi/c_abi_check.cpp): hipMalloc'ed buffers, a caller-created
stream, no Python and no torch in the process.  CPU: the program compiles and links against libegc_hip.so.
GPU: it runs one EGC layer forward (COO -> CSR -> plan -> GEMM -> fused aggregate/combine) in both weight layouts
and both GEMM forms and checks it against its own double-precision scalar restatement (<= 1e-5)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "c_abi", "_build", "c_abi_check")


def _build():
    subprocess.run(["bash", os.path.join(ROOT, "tests", "c_abi", "build.sh")], check=True, capture_output=True)


def test_c_host_compiles_and_links_against_the_library():
    from egc_amd import _C
    _C.load()                       # the library itself must be there first
    _build()
    assert os.access(BIN, os.X_OK)
    needed = subprocess.run(["readelf", "-d", BIN], check=True, capture_output=True, text=True).stdout
    assert "libegc_hip.so" in needed and "libtorch" not in needed and "libpython" not in needed


@pytest.mark.gpu
def test_c_host_runs_layer_forward_on_the_gpu():
    if not os.access(BIN, os.X_OK):
        _build()
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "c_abi_check: OK" in r.stdout and r.stdout.count("max |diff|") == 5, r.stdout   # four forward forms + the backward's two paths

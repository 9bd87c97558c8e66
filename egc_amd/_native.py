"""The compiled PyTorch binding (egc_amd/csrc_ext/egc_torch_ext.cpp -> egc_amd/lib/libegc_torch_ext.so): TORCH_LIBRARY
operators ``torch.ops.egc_amd_native.*`` over the same C ABI the ctypes table of _C.py binds, registered for HIP devices
only.  The layer modules' inference forward goes through it when the library has been built (``__graft_entry__.build()``
builds it); without it the ctypes path serves the same calls -- same kernels, more host time per call (DESIGN.md
section 5: host time per eval-mode layer call)."""
from __future__ import annotations

import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "lib", "libegc_torch_ext.so")
_OPS = None
_TRIED = False


def lib_path() -> str:
    return _PATH


def ops():
    """torch.ops.egc_amd_native, or None when the extension is not built / disabled (EGC_NO_NATIVE_EXT=1)."""
    global _OPS, _TRIED
    if not _TRIED:
        _TRIED = True
        if os.environ.get("EGC_NO_NATIVE_EXT", "0") in ("", "0") and os.path.exists(_PATH):
            from . import _C
            _C.load()                      # libegc_hip.so first: the extension links against it
            torch.ops.load_library(_PATH)
            _OPS = torch.ops.egc_amd_native
    return _OPS

#!/usr/bin/env python3
"""Componentwise error of the basis-transform GEMMs against float64: max over outputs of |got - ref| / (|x| @ |w| + |b|) for
the split-precision kernels (fp16x2 at the north-star shape, its long-k form, the 24-bit-operand form) and the plain
fp32-MFMA kernel, on ordinary inputs and on rows / columns spread over 2^+-100 -- the numbers behind the tolerances of
tests/test_gemm_gpu.py and the `dtype` / `gemm` fields of bench.py."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egc_amd import _C  # noqa: E402

DEV = "cuda:0"


def run(x, wcat, bcat, f_g, w_cols, mode):
    lib = _C.load()
    n, f_in = x.shape
    ldb = (f_g + 3) & ~3
    bases = torch.empty((n, ldb), device=DEV)
    wt = torch.empty((n, w_cols), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    if mode == "exact":
        _C.check(lib.egc_basis_transform_f32(x.data_ptr(), wcat.data_ptr(), bcat.data_ptr(), n, f_in, f_g, w_cols, bases.data_ptr(),
                                             ldb, wt.data_ptr(), st), "f32")
    else:
        flags = 1 if mode == "24bit" else 0   # EGC_GEMM_24BIT (include/egc_hip.h)
        nb = lib.egc_basis_pack_bytes(f_in, f_g, w_cols)
        planes = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _C.check(lib.egc_basis_pack_ex(wcat.data_ptr(), f_in, f_g, w_cols, flags, planes.data_ptr(), nb, st), "pack")
        _C.check(lib.egc_basis_transform_packed_ex(x.data_ptr(), planes.data_ptr(), bcat.data_ptr(), n, f_in, f_g, w_cols, flags,
                                                   bases.data_ptr(), ldb, wt.data_ptr(), st), "packed")
    torch.cuda.synchronize()
    return bases, wt


def err(x, wcat, bcat, f_g, bases, wt, keep=None):
    ref = x.double() @ wcat.double()
    bud = x.double().abs() @ wcat.double().abs()
    tiny = torch.finfo(torch.float32).tiny
    eb = (bases[:, :f_g].double() - ref[:, :f_g]).abs() / (bud[:, :f_g] + tiny)
    ew = (wt.double() - (ref[:, f_g:] + bcat.double())).abs() / (bud[:, f_g:] + bcat.double().abs() + tiny)
    if keep is not None:
        eb, ew = eb[keep], ew[keep]
    return max(float(eb.max()), float(ew.max())), float(torch.cat([eb.flatten(), ew.flatten()]).mean())


def main():
    g = torch.Generator().manual_seed(0)
    for label, n, f_in, f_g, w_cols, spread in (("north star 128 -> 64 + 128", 16384, 128, 64, 128, False),
                                                ("north star, rows / columns over 2^+-100", 4099, 128, 64, 128, True),
                                                ("ogbn-mag 352 -> 176 + 32", 8192, 352, 176, 32, False),
                                                ("ogbn-mag, spread", 4099, 352, 176, 32, True)):
        x = torch.randn(n, f_in, generator=g)
        wcat = torch.randn(f_in, f_g + w_cols, generator=g) * 0.2
        bcat = torch.randn(w_cols, generator=g)
        if spread:
            x = x * torch.exp2(torch.randint(-100, 100, (n, 1), generator=g).float())
            x[:, ::7] *= 1e-4
            wcat = wcat * torch.exp2(torch.randint(-20, 20, (1, f_g + w_cols), generator=g).float())
            bcat = torch.zeros(w_cols)
        x, wcat, bcat = x.to(DEV), wcat.to(DEV), bcat.to(DEV)
        out = []
        for mode in ("split", "24bit", "exact"):
            try:
                b, w = run(x, wcat, bcat, f_g, w_cols, mode)
                mx, mean = err(x, wcat, bcat, f_g, b, w)
                out.append(f"{mode}: max {mx:.2e} mean {mean:.2e}")
            except Exception as ex:  # noqa: BLE001
                out.append(f"{mode}: {ex!r}"[:80])
        print(f"{label}: " + " | ".join(out))


if __name__ == "__main__":
    main()

"""The basis-transform GEMM through the C ABI (egc_basis_pack / egc_basis_transform_packed and the exact
fp32 form) against float64: fp32-level accuracy on every shape path -- the fp16x2 register-stationary kernel
(96 < F_in <= 128, F_g % 32 == 0, 192 padded columns), its long-k form (128 < F_in <= 384, egc_gemm_f16x2k.hip),
the bf16x3 kernels (everything else) -- including rows
and columns of wildly different magnitude, zero rows, ragged sizes and non-finite inputs.

Tolerance: componentwise |got - ref| <= tol * (|x| @ |w| + |b|): 5e-7 for the fp16x2 kernels (measured 1.3e-7 on ordinary
inputs and on rows / columns spread over 2^+-100, tools/gemm_error.py: 2^-22 operand rounding + fp32 accumulation; the plain
fp32-MFMA kernel measures 3.2e-7, the 24-bit bf16x3 form 2.9e-7), 4e-6 for the bf16x3 kernels of the remaining shapes
(north_star's 1e-5 bound is on the output scale; these are stricter).  The fp16x2 kernel is also asserted to be NO LESS
accurate than the exact-fp32 kernel on the same inputs: bench.py calls the layer `f32` on that ground."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _transform(x, wcat, bcat, f_g, w_cols, exact=False):
    import ctypes as C
    from egc_amd import _C
    lib = _C.load()
    n, f_in = x.shape
    ldb = (f_g + 3) & ~3
    bases = torch.full((n, ldb), float("nan"), device=DEV)
    wt = torch.full((n, w_cols), float("nan"), device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    if exact:
        _C.check(lib.egc_basis_transform_f32(x.data_ptr(), wcat.data_ptr(), bcat.data_ptr(), n, f_in, f_g, w_cols,
                                             bases.data_ptr(), ldb, wt.data_ptr(), st), "f32")
    else:
        nb = lib.egc_basis_pack_bytes(f_in, f_g, w_cols)
        planes = torch.empty(nb, dtype=torch.uint8, device=DEV)
        _C.check(lib.egc_basis_pack(wcat.data_ptr(), f_in, f_g, w_cols, planes.data_ptr(), nb, st), "pack")
        _C.check(lib.egc_basis_transform_packed(x.data_ptr(), planes.data_ptr(), bcat.data_ptr(), n, f_in, f_g, w_cols,
                                                bases.data_ptr(), ldb, wt.data_ptr(), st), "packed")
    torch.cuda.synchronize()
    return bases, wt


def _err(x, wcat, bcat, f_g, bases, wt):
    ref = x.double() @ wcat.double()
    budget = x.double().abs() @ wcat.double().abs()
    tiny = torch.finfo(torch.float32).tiny
    eb = ((bases[:, :f_g].double() - ref[:, :f_g]).abs() / (budget[:, :f_g] + tiny)).max() if bases.numel() else 0.0
    ew = ((wt.double() - (ref[:, f_g:] + bcat.double())).abs() / (budget[:, f_g:] + bcat.double().abs() + tiny)).max() if wt.numel() else 0.0
    return max(float(eb), float(ew))


F16X2_TOL = 5e-7      # the fp16x2 register-stationary kernels (north-star shapes, long k)


def _check(x, wcat, bcat, f_g, w_cols, bases, wt, tol=4e-6):
    ref = x.double() @ wcat.double()
    budget = x.double().abs() @ wcat.double().abs()
    rb, rw = ref[:, :f_g], ref[:, f_g:] + bcat.double()
    bb, bw = budget[:, :f_g], budget[:, f_g:] + bcat.double().abs()
    tiny = torch.finfo(torch.float32).tiny
    eb = ((bases[:, :f_g].double() - rb).abs() / (bb + tiny)).max() if bases.numel() else 0.0
    ew = ((wt.double() - rw).abs() / (bw + tiny)).max() if wt.numel() else 0.0
    assert float(eb) <= tol and float(ew) <= tol, (float(eb), float(ew))
    if bases.size(1) > f_g:
        assert bool((bases[:, f_g:] == 0).all())  # pad columns are written as zeros


@pytest.mark.parametrize("n", [1, 63, 64, 65, 1000, 16391])
@pytest.mark.parametrize("f_in,f_g,w_cols", [
    (128, 64, 128),   # north star: fp16x2 kernel
    (100, 64, 126),   # fp16x2 kernel, F_in < 128 (zero-filled ring), W not a multiple of 4
    (128, 32, 160),   # fp16x2 kernel, 1 bases tile + 5 weightings tiles
    (124, 124, 48),   # bf16x3 (F_g % 32 != 0)
    (168, 84, 32),    # long-k fp16x2 kernel (128 < F_in <= 384, <= 16 column tiles of 16)
    (352, 176, 32),   # long-k fp16x2: the ogbn-mag layer (13 column tiles, 11 k-steps)
    (224, 224, 48),   # long-k fp16x2: molhiv 224/H4/B4 (17 column tiles: ONE launch, two tiles per multiplier wavefront, the last one idle)
    (200, 150, 30),   # two tiles per wavefront (12 tiles): padded bases (ldb 152), ragged weightings width (dword stores)
    (160, 296, 0),    # 19 tiles, no weightings: the d x GEMM of 296 / H8 / B4
    (224, 200, 100),  # 13 + 7 = 20 tiles: the most one launch takes; a wavefront whose two tiles straddle bases | weightings
    (192, 320, 16),   # 21 tiles: two launches
    (128, 168, 0),    # fp16x2 kernel without weightings, ldb % 32 != 0: the d x GEMM of the 168-wide nets (round 6)
    (116, 184, 0),    # the same, F_in < 128: arxiv EGC-S's d x GEMM shape class
    (384, 64, 128),   # long-k fp16x2: the longest k it takes
    (132, 20, 7),     # long-k fp16x2: ragged weightings width (dword stores), partial column tiles
    (300, 300, 48),   # bf16x3, LDS-staged general kernel (too many column tiles for the long-k kernel)
    (7, 5, 3),        # tiny ragged
])
def test_packed_gemm_matches_float64(n, f_in, f_g, w_cols):
    g = torch.Generator(device="cpu").manual_seed(1000 * n + f_in)
    x = torch.randn(n, f_in, generator=g).to(DEV)
    wcat = (torch.randn(f_in, f_g + w_cols, generator=g) * 0.2).to(DEV)
    bcat = torch.randn(w_cols, generator=g).to(DEV)
    # the shapes the fp16x2 kernels serve (egc_gemm_split.h: f16x2_shape / f16x2k_shape) are held to 5e-7
    f16x2 = (f_in, f_g, w_cols) in {(128, 64, 128), (100, 64, 126), (128, 32, 160), (168, 84, 32), (352, 176, 32), (384, 64, 128),
                                    (132, 20, 7), (224, 224, 48), (200, 150, 30), (160, 296, 0), (224, 200, 100), (192, 320, 16), (128, 168, 0), (116, 184, 0)}
    _check(x, wcat, bcat, f_g, w_cols, *_transform(x, wcat, bcat, f_g, w_cols), tol=F16X2_TOL if f16x2 else 4e-6)


def test_rows_and_columns_of_wildly_different_magnitude():
    g = torch.Generator(device="cpu").manual_seed(7)
    n, f_in, f_g, w_cols = 4099, 128, 64, 128
    x = torch.randn(n, f_in, generator=g) * torch.exp2(torch.randint(-100, 100, (n, 1), generator=g).float())
    x[:, ::7] *= 1e-4                      # small elements next to large ones inside a row
    x[5] = 0.0                             # an all-zero row
    x[6] = 1e-42                           # a denormal row
    wcat = torch.randn(f_in, f_g + w_cols, generator=g) * torch.exp2(torch.randint(-20, 20, (1, f_g + w_cols), generator=g).float())
    wcat[:, 3] = 0.0                       # an all-zero column
    bcat = torch.zeros(w_cols)
    x, wcat, bcat = x.to(DEV), wcat.to(DEV), bcat.to(DEV)
    bases, wt = _transform(x, wcat, bcat, f_g, w_cols)
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[6] = False
    _check(x[keep], wcat, bcat, f_g, w_cols, bases[keep], wt[keep], tol=F16X2_TOL)
    # ... and no less accurate than the plain fp32-MFMA kernel on the same inputs (rows spread over 2^+-100 included)
    be, we = _transform(x, wcat, bcat, f_g, w_cols, exact=True)
    err_split = _err(x[keep], wcat, bcat, f_g, bases[keep], wt[keep])
    err_exact = _err(x[keep], wcat, bcat, f_g, be[keep], we[keep])
    assert err_split <= err_exact, (err_split, err_exact)
    assert bool((bases[5] == 0).all()) and bool((wt[5] == 0).all())
    assert bool((bases[:, 3] == 0).all())
    # a row whose largest magnitude is below 2^-113 is scaled by 2^114 only: it keeps its order of magnitude
    # but not 24 bits (the fp32 GEMMs of the matrix cores flush such inputs altogether)
    ref6 = (x[6].double() @ wcat.double())[:f_g]
    assert float((bases[6, :f_g].double() - ref6).abs().max()) <= 1e-3 * float((x[6].double().abs() @ wcat.double().abs()).max())


def test_non_finite_inputs_stay_in_their_rows():
    g = torch.Generator(device="cpu").manual_seed(9)
    n, f_in, f_g, w_cols = 300, 128, 64, 128
    x = torch.randn(n, f_in, generator=g)
    x[17, 5] = float("inf")
    x[130, 77] = float("nan")
    wcat = torch.randn(f_in, f_g + w_cols, generator=g)
    bcat = torch.randn(w_cols, generator=g)
    x, wcat, bcat = x.to(DEV), wcat.to(DEV), bcat.to(DEV)
    bases, wt = _transform(x, wcat, bcat, f_g, w_cols)
    bad = torch.zeros(n, dtype=torch.bool, device=DEV)
    bad[17] = bad[130] = True
    assert not bool(torch.isfinite(bases[17]).all()) and not bool(torch.isfinite(wt[130]).all())
    keep = ~bad
    _check(x[keep], wcat, bcat, f_g, w_cols, bases[keep], wt[keep])


def test_exact_fp32_form_agrees():
    g = torch.Generator(device="cpu").manual_seed(11)
    n, f_in, f_g, w_cols = 2000, 128, 64, 128
    x, wcat, bcat = (torch.randn(n, f_in, generator=g).to(DEV), torch.randn(f_in, f_g + w_cols, generator=g).to(DEV),
                     torch.randn(w_cols, generator=g).to(DEV))
    _check(x, wcat, bcat, f_g, w_cols, *_transform(x, wcat, bcat, f_g, w_cols, exact=True))


def test_row_ranges_launched_separately(monkeypatch):
    """Arrays past the 2 GiB buffer-offset range are processed as consecutive row ranges: force tiny ranges."""
    monkeypatch.setenv("EGC_GEMM_MAX_ROWS", "192")
    g = torch.Generator(device="cpu").manual_seed(13)
    n, f_in, f_g, w_cols = 1000, 128, 64, 128
    x, wcat, bcat = (torch.randn(n, f_in, generator=g).to(DEV), torch.randn(f_in, f_g + w_cols, generator=g).to(DEV),
                     torch.randn(w_cols, generator=g).to(DEV))
    _check(x, wcat, bcat, f_g, w_cols, *_transform(x, wcat, bcat, f_g, w_cols))


def test_long_k_kernel_rows_of_wildly_different_magnitude():
    """The per-row power-of-two scaling of the long-k fp16x2 kernel (row maximum over the WHOLE row before the split)."""
    g = torch.Generator(device="cpu").manual_seed(21)
    n, f_in, f_g, w_cols = 3001, 352, 176, 32
    x = torch.randn(n, f_in, generator=g) * torch.exp2(torch.randint(-100, 100, (n, 1), generator=g).float())
    x[:, ::5] *= 1e-4
    x[7] = 0.0
    wcat = torch.randn(f_in, f_g + w_cols, generator=g) * torch.exp2(torch.randint(-20, 20, (1, f_g + w_cols), generator=g).float())
    wcat[:, 9] = 0.0
    bcat = torch.randn(w_cols, generator=g)
    x, wcat, bcat = x.to(DEV), wcat.to(DEV), bcat.to(DEV)
    bases, wt = _transform(x, wcat, bcat, f_g, w_cols)
    _check(x, wcat, bcat, f_g, w_cols, bases, wt)
    assert bool((bases[7] == 0).all()) and bool((bases[:, 9] == 0).all())


def test_long_k_kernel_row_ranges_launched_separately(monkeypatch):
    monkeypatch.setenv("EGC_GEMM_MAX_ROWS", "96")
    g = torch.Generator(device="cpu").manual_seed(23)
    n, f_in, f_g, w_cols = 500, 352, 176, 32
    x, wcat, bcat = (torch.randn(n, f_in, generator=g).to(DEV), torch.randn(f_in, f_g + w_cols, generator=g).to(DEV),
                     torch.randn(w_cols, generator=g).to(DEV))
    _check(x, wcat, bcat, f_g, w_cols, *_transform(x, wcat, bcat, f_g, w_cols))


# --- bases = x W + addend (egc_basis_transform_packed_add: the residual branch's gradient joining d x in the d x GEMM) ------------

@pytest.mark.parametrize("n", [1, 16, 1000, 52771])
@pytest.mark.parametrize("k,f", [(272, 224), (192, 296), (368, 304), (208, 352), (184, 136), (132, 100)])
def test_gemm_addend_equals_gemm_then_add(n, k, f):
    """out = x W + addend in the long-k kernels' store (both forms of the kernel, one and two launches over the column tiles) against
    the plain call followed by a float32 add: the same bits (one rounding of the same two numbers); rows past the last full tile and
    the padding columns of the last column tile included."""
    from egc_amd import _C
    lib = _C.load()
    g = torch.Generator(device="cpu").manual_seed(n + k + f)
    x = torch.randn(n, k, generator=g).to(DEV)
    wt = torch.randn(f, k, generator=g).to(DEV)            # the transposed operand, as the d x GEMM has it
    add = torch.randn(n, f, generator=g).to(DEV)
    pb = int(lib.egc_basis_pack_bytes(k, f, 0))
    packed = torch.empty(pb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _C.check(lib.egc_basis_pack_transposed(wt.data_ptr(), k, k, f, 0, packed.data_ptr(), pb, st), "egc_basis_pack_transposed")
    plain = torch.full((n, f), float("nan"), device=DEV)
    _C.check(lib.egc_basis_transform_packed(x.data_ptr(), packed.data_ptr(), None, n, k, f, 0, plain.data_ptr(), f, None, st),
             "egc_basis_transform_packed")
    fused = torch.full((n, f), float("nan"), device=DEV)
    rc = lib.egc_basis_transform_packed_add(x.data_ptr(), packed.data_ptr(), None, n, k, f, 0, 0, add.data_ptr(), fused.data_ptr(), f, None, st)
    torch.cuda.synchronize()
    assert rc == 0
    assert torch.equal(fused, plain + add)
    assert float((plain.double() - x.double() @ wt.double().t()).abs().max()) <= 1e-5 * float((x.double().abs() @ wt.double().abs().t()).max())


def test_gemm_addend_outside_the_long_k_kernels_is_refused():
    from egc_amd import _C
    lib = _C.load()
    n, k, f = 100, 128, 168                               # the 168-wide nets' d x GEMM: k = 96 + 32, not a long-k shape
    x, add, out = torch.randn(n, k, device=DEV), torch.randn(n, f, device=DEV), torch.empty(n, f, device=DEV)
    pb = int(lib.egc_basis_pack_bytes(k, f, 0))
    packed = torch.empty(pb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _C.check(lib.egc_basis_pack_transposed(torch.randn(f, k, device=DEV).data_ptr(), k, k, f, 0, packed.data_ptr(), pb, st), "pack")
    assert lib.egc_basis_transform_packed_add(x.data_ptr(), packed.data_ptr(), None, n, k, f, 0, 0, add.data_ptr(), out.data_ptr(), f, None, st) == 4
    assert lib.egc_basis_transform_packed_add(x.data_ptr(), packed.data_ptr(), None, n, k, f, 0, 0, None, out.data_ptr(), f, None, st) == 0


# --- weight gradient x^T @ d + column sums (egc_weight_grad_f32, egc_gemm_xt.hip) ---------------------------------

def _weight_grad(x, d, sums=True):
    from egc_amd import _C
    lib = _C.load()
    n, f = x.shape
    k = d.size(1)
    out = torch.full((f, k), float("nan"), device=DEV)
    cs = torch.full((k,), float("nan"), device=DEV) if sums else None
    nb = int(lib.egc_weight_grad_workspace_bytes(n, f, k))
    ws = torch.full((max(nb, 16) // 4,), float("nan"), device=DEV)   # contents irrelevant on entry
    _C.check(lib.egc_weight_grad_f32(x.data_ptr(), x.stride(0), d.data_ptr(), d.stride(0), n, f, k, out.data_ptr(),
                                     cs.data_ptr() if sums else None, ws.data_ptr(), ws.numel() * 4,
                                     torch.cuda.current_stream().cuda_stream), "egc_weight_grad_f32")
    torch.cuda.synchronize()
    return out, cs


@pytest.mark.parametrize("n", [0, 1, 31, 32, 33, 1000, 16391, 70001])
@pytest.mark.parametrize("f,k", [(128, 192), (100, 132), (4, 4), (32, 64), (64, 128), (352, 208), (36, 260), (132, 68)])
def test_weight_grad_matches_float64(n, f, k):
    g = torch.Generator(device="cpu").manual_seed(n * 131 + f * 7 + k)
    x = torch.randn(n, f, generator=g).to(DEV)
    d = (torch.randn(n, k, generator=g) * torch.logspace(-6, 2, k)).to(DEV)   # columns of very different size
    out, cs = _weight_grad(x, d)
    ref = x.double().t() @ d.double()
    budget = x.double().abs().t() @ d.double().abs()
    tiny = torch.finfo(torch.float32).tiny
    # fp32 products, fp32 running sums over <= a few hundred rows per range, then <= 256 partial sums
    tol = 2e-6 * max(1.0, (n ** 0.5) / 8)
    assert float(((out.double() - ref).abs() / (budget + tiny)).max() if n else out.abs().max()) <= tol
    rs, bs = d.double().sum(0), d.double().abs().sum(0)
    assert float(((cs.double() - rs).abs() / (bs + tiny)).max() if n else cs.abs().max()) <= tol


XT_TILES = [(1, 1, 4), (1, 2, 4), (1, 3, 4), (2, 1, 4), (2, 2, 4), (2, 3, 4), (4, 1, 4), (4, 2, 4), (4, 3, 4), (5, 2, 4), (5, 3, 4),
            (7, 2, 4), (7, 3, 3), (7, 3, 4), (7, 5, 4), (7, 3, 6)]


@pytest.mark.parametrize("tile", XT_TILES, ids=lambda t: "%dx%d" % (32 * t[0], 16 * t[2] * t[1]))
@pytest.mark.parametrize("n,f,k", [(4133, 224, 272), (1501, 296, 180), (777, 132, 196)])
def test_weight_grad_every_compiled_tile(tile, n, f, k, monkeypatch):
    """Every tile shape of the exact-fp32 kernel (round 6: 160 / 224-row tiles for the reference's 168 - 296-wide nets), forced
    through EGC_XT_TILE at the output shapes of those nets (ragged against every tile), against float64; the plan query names the
    tile that ran."""
    import ctypes as C
    from egc_amd import _C
    monkeypatch.setenv("EGC_XT_TILE", "%d,%d,%d" % tile)
    plan = (C.c_int32 * 8)()
    _C.check(_C.load().egc_weight_grad_plan(n, f, k, C.cast(plan, C.c_void_p)), "egc_weight_grad_plan")
    assert list(plan)[:3] == [0, 32 * tile[0], 16 * tile[2] * tile[1]] and plan[7] == 128 * tile[2]
    assert plan[3] * plan[1] >= f and plan[4] * plan[2] >= k and plan[5] * plan[6] >= n
    g = torch.Generator(device="cpu").manual_seed(n + f + k)
    x = torch.randn(n, f, generator=g).to(DEV)
    d = (torch.randn(n, k, generator=g) * torch.logspace(-4, 2, k)).to(DEV)
    out, cs = _weight_grad(x, d)
    ref = x.double().t() @ d.double()
    budget = x.double().abs().t() @ d.double().abs()
    assert float(((out.double() - ref).abs() / budget).max()) <= 2e-6
    assert float(((cs.double() - d.double().sum(0)).abs() / d.double().abs().sum(0)).max()) <= 2e-6


def test_weight_grad_strided_operands_and_no_sums():
    g = torch.Generator(device="cpu").manual_seed(9)
    xs = torch.randn(5000, 160, generator=g).to(DEV)
    ds = torch.randn(5000, 256, generator=g).to(DEV)
    x, d = xs[:, 16:144], ds[:, 64:256]            # column blocks: row strides 160 / 256
    out, cs = _weight_grad(x, d, sums=False)
    ref = x.double().t() @ d.double()
    assert cs is None and float((out.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_weight_grad_rejects_unaligned_shapes():
    from egc_amd import _C
    lib = _C.load()
    x = torch.zeros(8, 6, device=DEV)
    d = torch.zeros(8, 8, device=DEV)
    o = torch.zeros(6, 8, device=DEV)
    ws = torch.zeros(1 << 16, device=DEV)
    rc = lib.egc_weight_grad_f32(x.data_ptr(), 6, d.data_ptr(), 8, 8, 6, 8, o.data_ptr(), None, ws.data_ptr(),
                                 ws.numel() * 4, torch.cuda.current_stream().cuda_stream)
    assert rc == 4   # EGC_ERR_UNSUPPORTED


@pytest.mark.parametrize("n", [0, 5, 1000, 40001])
@pytest.mark.parametrize("f,k,e", [(128, 192, 128), (100, 132, 36), (32, 64, 4)])
def test_weight_grad_with_the_column_sums_of_a_third_array(n, f, k, e):
    """egc_weight_grad_ex_f32: x^T d, the column sums of d and the column sums of a third array (grad_out) in one pass."""
    from egc_amd import functional as F
    g = torch.Generator(device="cpu").manual_seed(n + 3 * f + 5 * k + 7 * e)
    x = torch.randn(n, f, generator=g).to(DEV)
    d = torch.randn(n, k, generator=g).to(DEV)
    ex = (torch.randn(n, e, generator=g) * 3 + 0.5).to(DEV)
    w, cs, es = F._weight_grads(x, d, col_sums=True, extra=ex)
    tol = 2e-6 * max(1.0, (n ** 0.5) / 8)
    tiny = torch.finfo(torch.float32).tiny
    ref, bud = x.double().t() @ d.double(), x.double().abs().t() @ d.double().abs()
    assert float(((w.double() - ref).abs() / (bud + tiny)).max() if n else w.abs().max()) <= tol
    for got, src in ((cs, d), (es, ex)):
        r, b = src.double().sum(0), src.double().abs().sum(0)
        assert got.shape == r.shape
        assert float(((got.double() - r).abs() / (b + tiny)).max() if n else got.abs().max()) <= tol


@pytest.mark.parametrize("f_in,f_g,w_cols", [(128, 64, 128), (192, 128, 0), (352, 176, 32), (124, 124, 48)])
def test_pack_transposed_equals_pack(f_in, f_g, w_cols):
    """egc_basis_pack_transposed(W^T stored row-major) produces the very planes egc_basis_pack(W) does (all three
    split forms: fp16x2, long-k fp16x2, bf16x3)."""
    from egc_amd import _C
    lib = _C.load()
    g = torch.Generator(device="cpu").manual_seed(f_in + f_g)
    w = (torch.randn(f_in, f_g + w_cols, generator=g) * torch.logspace(-3, 3, f_g + w_cols)).to(DEV)
    wt = torch.zeros(f_g + w_cols, f_in + 4, device=DEV)      # row stride > f_in
    wt[:, :f_in] = w.t()
    nb = lib.egc_basis_pack_bytes(f_in, f_g, w_cols)
    a = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    b = torch.zeros(nb, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _C.check(lib.egc_basis_pack(w.data_ptr(), f_in, f_g, w_cols, a.data_ptr(), nb, st), "pack")
    _C.check(lib.egc_basis_pack_transposed(wt.data_ptr(), wt.stride(0), f_in, f_g, w_cols, b.data_ptr(), nb, st), "pack_t")
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("n", [7, 3000, 52771])
@pytest.mark.parametrize("f_in,H,A,B,L,Ls,n_parts", [(224, 4, 3, 4, 56, 56, 4), (296, 8, 1, 4, 37, 40, 4), (168, 8, 1, 4, 21, 24, 4), (304, 8, 1, 8, 38, 40, 1)])
def test_weight_grad_params_at_the_wide_nets_shapes(n, f_in, H, A, B, L, Ls, n_parts):
    """The same entry point beyond one accumulator tile (the exact-fp32 tile grid; round 6: the compiled training nodes write the
    reference's wide nets' parameter gradients through it): == egc_weight_grad_ex_f32 + the gradient unpack, bit for bit, no third
    array (that rides only in the one-tile kernel: EGC_ERR_UNSUPPORTED)."""
    import ctypes as C
    from egc_amd import _C
    lib = _C.load()
    torch.manual_seed(n + f_in)
    W, f_g = H * B * A, B * Ls
    k = f_g + W
    x = torch.randn(n, f_in, device=DEV)
    d = torch.randn(n, k, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.egc_weight_grad_ex_workspace_bytes(n, f_in, k, 0)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    part_shape = (f_in, B * L) if n_parts == 1 else (f_in, L)

    def fresh():
        return ([torch.full(part_shape, float("nan"), device=DEV) for _ in range(n_parts)],
                torch.full((W, f_in), float("nan"), device=DEV), torch.full((W,), float("nan"), device=DEV))
    dwcat, cs = torch.empty(f_in, k, device=DEV), torch.empty(k, device=DEV)
    _C.check(lib.egc_weight_grad_ex_f32(x.data_ptr(), f_in, d.data_ptr(), k, n, f_in, k, dwcat.data_ptr(), cs.data_ptr(), None, 0, 0, None,
                                        ws.data_ptr(), ws.numel(), st), "ex")
    parts_a, cw_a, cb_a = fresh()
    ptrs = (C.c_void_p * n_parts)(*[p.data_ptr() for p in parts_a])
    dbcat = cs[f_g:].contiguous()
    _C.check(lib.egc_weights_pack_f32(ptrs, n_parts, cw_a.data_ptr(), None, f_in, H, A, B, L, Ls, 0, dwcat.data_ptr(), None, 1, st), "unpack")
    parts_b, cw_b, cb_b = fresh()
    ptrs_b = (C.c_void_p * n_parts)(*[p.data_ptr() for p in parts_b])
    _C.check(lib.egc_weight_grad_params_f32(x.data_ptr(), f_in, d.data_ptr(), k, n, f_in, H, A, B, L, Ls, 0, ptrs_b, n_parts, cw_b.data_ptr(), None,
                                            cb_b.data_ptr(), None, 0, 0, None, ws.data_ptr(), ws.numel(), st), "params")
    torch.cuda.synchronize()
    for a, b in zip(parts_a + [cw_a, dbcat], parts_b + [cw_b, cb_b]):
        assert torch.equal(a, b) and not torch.isnan(b).any()
    e = torch.randn(n, 128, device=DEV)
    es = torch.empty(128, device=DEV)
    ws2 = torch.empty(max(lib.egc_weight_grad_ex_workspace_bytes(n, f_in, k, 128), 16), dtype=torch.uint8, device=DEV)
    assert lib.egc_weight_grad_params_f32(x.data_ptr(), f_in, d.data_ptr(), k, n, f_in, H, A, B, L, Ls, 0, ptrs_b, n_parts, cw_b.data_ptr(), None,
                                          cb_b.data_ptr(), e.data_ptr(), 128, 128, es.data_ptr(), ws2.data_ptr(), ws2.numel(), st) == 4


@pytest.mark.parametrize("n", [5, 1000, 40001])
@pytest.mark.parametrize("f_in,H,A,B,L,Ls,permute,n_parts", [
    (128, 8, 4, 4, 16, 16, True, 1),      # EGConv north star: one [F_in, B L] basis matrix, Linear rows [h][a][b]
    (48, 4, 3, 4, 8, 8, False, 4),        # EfficientGraphConv: B basis matrices, rows [h][b][a]
    (64, 2, 1, 2, 21, 24, False, 2),      # padded bases (L = 21 -> Ls = 24): padding columns have no parameter behind them
])
def test_weight_grad_params_equals_the_two_step_form(n, f_in, H, A, B, L, Ls, permute, n_parts):
    """egc_weight_grad_params_f32 (x^T d and the column sums written straight into the parameters' gradients through the
    pack's index map) == egc_weight_grad_ex_f32 followed by the gradient unpack of egc_weights_pack_f32, bit for bit: same
    partial products, same reduction order, one launch less."""
    import ctypes as C
    from egc_amd import _C
    lib = _C.load()
    torch.manual_seed(n + f_in)
    W, f_g = H * B * A, B * Ls
    k, f_out = f_g + W, H * L
    e_cols = (f_out + 3) // 4 * 4
    x = torch.randn(n, f_in, device=DEV)
    d = torch.randn(n, k, device=DEV)
    e = torch.randn(n, e_cols, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    nb = lib.egc_weight_grad_ex_workspace_bytes(n, f_in, k, e_cols)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device=DEV)
    part_shape = (f_in, B * L) if n_parts == 1 else (f_in, L)

    def fresh():
        return ([torch.full(part_shape, float("nan"), device=DEV) for _ in range(n_parts)],
                torch.full((W, f_in), float("nan"), device=DEV), torch.full((W,), float("nan"), device=DEV))
    # two steps
    dwcat, cs, es = torch.empty(f_in, k, device=DEV), torch.empty(k, device=DEV), torch.empty(e_cols, device=DEV)
    _C.check(lib.egc_weight_grad_ex_f32(x.data_ptr(), f_in, d.data_ptr(), k, n, f_in, k, dwcat.data_ptr(), cs.data_ptr(), e.data_ptr(),
                                        e_cols, e_cols, es.data_ptr(), ws.data_ptr(), ws.numel(), st), "ex")
    parts_a, cw_a, cb_a = fresh()
    ptrs = (C.c_void_p * n_parts)(*[p.data_ptr() for p in parts_a])
    dbcat = cs[f_g:].contiguous()
    _C.check(lib.egc_weights_pack_f32(ptrs, n_parts, cw_a.data_ptr(), cb_a.data_ptr(), f_in, H, A, B, L, Ls, int(permute),
                                      dwcat.data_ptr(), dbcat.data_ptr(), 1, st), "unpack")
    # one step
    parts_b, cw_b, cb_b = fresh()
    es_b = torch.empty(e_cols, device=DEV)
    ptrs_b = (C.c_void_p * n_parts)(*[p.data_ptr() for p in parts_b])
    _C.check(lib.egc_weight_grad_params_f32(x.data_ptr(), f_in, d.data_ptr(), k, n, f_in, H, A, B, L, Ls, int(permute), ptrs_b, n_parts,
                                            cw_b.data_ptr(), cb_b.data_ptr(), None, e.data_ptr(), e_cols, e_cols, es_b.data_ptr(),
                                            ws.data_ptr(), ws.numel(), st), "params")
    torch.cuda.synchronize()
    for a, b in zip(parts_a + [cw_a, cb_a, es], parts_b + [cw_b, cb_b, es_b]):
        assert torch.equal(a, b) and not torch.isnan(b).any()
    # and against float64
    ref = x.double().t() @ d.double()
    assert float((dwcat.double() - ref).abs().max() / ref.abs().max()) <= 1e-5


@pytest.mark.parametrize("f_in,f_g,w_cols", [(128, 64, 128), (352, 176, 32)])
def test_fp16x2_gemm_is_no_less_accurate_than_the_exact_fp32_kernel(f_in, f_g, w_cols):
    """VERDICT r3 next #4a: bench.py reports the layer as `f32` although its GEMM runs on 22-bit operands.  The ground: on the
    same inputs its componentwise error against float64 does not exceed that of the plain fp32-MFMA kernel (fewer, shorter
    fp32 accumulation chains outweigh the 2^-22 operand rounding) -- north-star shape and the ogbn-mag shape."""
    g = torch.Generator(device="cpu").manual_seed(77 + f_in)
    n = 16384
    x = torch.randn(n, f_in, generator=g).to(DEV)
    wcat = (torch.randn(f_in, f_g + w_cols, generator=g) * 0.2).to(DEV)
    bcat = torch.randn(w_cols, generator=g).to(DEV)
    err_split = _err(x, wcat, bcat, f_g, *_transform(x, wcat, bcat, f_g, w_cols))
    err_exact = _err(x, wcat, bcat, f_g, *_transform(x, wcat, bcat, f_g, w_cols, exact=True))
    assert err_split <= F16X2_TOL, err_split
    assert err_split <= err_exact, (err_split, err_exact)

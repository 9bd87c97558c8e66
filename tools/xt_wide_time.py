"""x^T d + column sums (egc_weight_grad_ex_f32) at the output shapes of the reference's wide nets, where the exact-fp32 kernel
xt_gemm_kernel runs: HIP-event time per call, the error against float64 and, for comparison, the library's split GEMM
(functional._xt_library).  EGC_XT_SHAPES="n,f,k;..." overrides the list.  Development aid; parity: tests/test_backward_gpu.py."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egc_amd import functional as F, _C
dev = torch.device("cuda:0")
lib = _C.load()
shapes = [(52771, 224, 272), (52771, 296, 180), (241600, 168, 116), (2998, 168, 116), (16000, 304, 368), (169343, 136, 184),
          (169343, 128, 184), (169343, 256, 320), (6500, 224, 272), (30000, 136, 184), (700, 168, 116)]
if os.environ.get("EGC_XT_SHAPES"):
    shapes = [tuple(int(v) for v in s.split(",")) for s in os.environ["EGC_XT_SHAPES"].split(";")]


def direct(x, d):
    n, f = x.shape
    k = d.size(1)
    out = torch.empty((f, k), dtype=torch.float32, device=dev)
    cs = torch.empty(k, dtype=torch.float32, device=dev)
    nbytes = int(lib.egc_weight_grad_ex_workspace_bytes(n, f, k, 0))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=dev)
    _C.check(lib.egc_weight_grad_ex_f32(x.data_ptr(), x.stride(0), d.data_ptr(), d.stride(0), n, f, k, out.data_ptr(), cs.data_ptr(),
                                        None, 0, 0, None, ws.data_ptr(), ws.numel(), F._stream_ptr(dev)), "egc_weight_grad_ex_f32")
    return out, cs


def timed(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


TILES = [(1, 1, 4), (1, 2, 4), (1, 3, 4), (2, 1, 4), (2, 2, 4), (2, 3, 4), (4, 1, 4), (4, 2, 4), (4, 3, 4), (5, 2, 4), (5, 3, 4), (7, 2, 4),
         (7, 3, 3), (7, 3, 4), (7, 5, 4), (7, 3, 6)]


def plan(n, f, k):
    import ctypes as C
    buf = (C.c_int32 * 8)()
    _C.check(lib.egc_weight_grad_plan(n, f, k, C.cast(buf, C.c_void_p)), "egc_weight_grad_plan")
    return list(buf)


if os.environ.get("EGC_XT_SWEEP"):   # every compiled tile at every shape: what the host's model (xt_plan) is held against
    for n, f, k in shapes:
        x = torch.randn(n, f, device=dev)
        d = torch.randn(n, k, device=dev)
        os.environ.pop("EGC_XT_TILE", None)
        chosen = plan(n, f, k)
        row = []
        for mt, nt, wn in TILES:
            os.environ["EGC_XT_TILE"] = f"{mt},{nt},{wn}"
            pl = plan(n, f, k)
            if pl[0]:
                continue
            row.append((timed(lambda: direct(x, d), 50), pl))
        os.environ.pop("EGC_XT_TILE", None)
        print(f"N={n} {f} x {k}: model picks {chosen[1]} x {chosen[2]} ({chosen[3]} x {chosen[4]} tiles, {chosen[5]} ranges)")
        for t, pl in sorted(row, key=lambda r: r[0]):
            print(f"    {t:7.1f} us  tile {pl[1]:3d} x {pl[2]:3d}  grid {pl[3]} x {pl[4]} x {pl[5]} ranges of {pl[6]} rows" + ("   <- picked" if pl[1:5] == chosen[1:5] else ""), flush=True)
    sys.exit(0)

for n, f, k in shapes:
    torch.manual_seed(0)
    x = torch.randn(n, f, device=dev)
    d = torch.randn(n, k, device=dev) * torch.rand(n, 1, device=dev)
    out, cs = direct(x, d)
    ref = x.double().t() @ d.double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    errs = float((cs.double() - d.double().sum(0)).abs().max() / d.double().sum(0).abs().max())
    t = timed(lambda: direct(x, d))
    tl = timed(lambda: (F._xt_library(x, d), F._column_sums(d)))
    mb = (x.numel() + d.numel()) * 4 / 1e6
    fl = 2.0 * n * f * k / 1e6
    print(f"N={n} {f} x {k}: kernel + reduction {t:.1f} us ({mb:.0f} MB, {fl / t / 1e6:.1f} TFLOP/s of 157 fp32-MFMA); library split GEMM + sums {tl:.1f} us; "
          f"max err / max |out| {err:.1e}, sums {errs:.1e}", flush=True)

"""The weight-gradient launch x^T d (egc_weight_grad_ex_f32: xt_gemm_bf16x3_kernel + its reduction) alone, at the config-2 and the molhiv
row counts, HIP-event time per call and the error against float64 (development aid; parity: tests/test_backward_gpu.py, test_gemm_gpu.py)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egc_amd import functional as F
dev = torch.device("cuda:0")
for n in (169343, 52771, 2998):
    torch.manual_seed(0)
    x = torch.randn(n, 128, device=dev)
    d = torch.randn(n, 192, device=dev) * torch.rand(n, 1, device=dev)
    g = torch.randn(n, 128, device=dev)
    out = F._weight_grads(x, d, col_sums=True, extra=g)
    ref = x.double().t() @ d.double()
    err = float((out[0].double() - ref).abs().max() / ref.abs().max())
    errs = float((out[1].double() - d.double().sum(0)).abs().max() / d.double().sum(0).abs().max())
    for _ in range(5): F._weight_grads(x, d, col_sums=True, extra=g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 200
    e0.record()
    for _ in range(reps): F._weight_grads(x, d, col_sums=True, extra=g)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e3
    mb = (x.numel() + d.numel() + g.numel()) * 4 / 1e6
    print(f"N={n}: x^T d + column sums {t:.1f} us per call ({mb:.1f} MB: {mb / t / 8e3 * 1e3:.2f} of 8 TB/s); max err / max |out| {err:.1e}, sums {errs:.1e}", flush=True)

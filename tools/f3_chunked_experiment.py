#!/usr/bin/env python3
"""SURVEY.md 8(f) rank 3 / VERDICT r2 next #5: the `weightings` round trip without a fused kernel -- existing entry points
only.  A bases-only GEMM over all rows, then K row chunks of { GEMM of the chunk's weightings into ONE small buffer that
stays in the L2s / Infinity Cache -> egc_aggregate_combine_rows_f32 over the chunk }.  The 87 MB written and 87 MB read
back by the two-launch layer never reach HBM; the price is a second read of x (87 MB) and 2 K + 1 launches instead of 2.
Prints the layer time for several K next to the default path (config 2 of BASELINE.json)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import egc_amd  # noqa: E402
from egc_amd import _C  # noqa: E402
from egc_amd.functional import pack_weights  # noqa: E402
from egc_amd.workloads import arxiv_like  # noqa: E402
from types import SimpleNamespace  # noqa: E402


def timed(fn, iters=50, reps=5):
    out = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(iters):
            fn()
        e.record()
        e.synchronize()
        out.append(s.elapsed_time(e) / iters)
    return sorted(out)[reps // 2]


def main():
    dev = torch.device("cuda:0")
    lib = _C.load()
    torch.manual_seed(0)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4, cached=True).to(dev).eval()
    with torch.no_grad():
        conv.bias.normal_()
    spec = conv._spec_coo
    ei, n = arxiv_like(seed=0)
    graph = egc_amd.CSRGraph.from_edge_index(ei.to(dev), n, build="sort").trim_launches()
    x = torch.randn(n, 128, device=dev)
    wcat, bcat = conv._packed_weights()
    planes = pack_weights(spec, wcat)
    bias = conv.bias.detach()
    ldb, W, fg = spec.ldb, spec.w_cols, spec.f_g
    bases = torch.empty((n, ldb), device=dev)
    weightings = torch.empty((n, W), device=dev)
    out = torch.empty((n, 128), device=dev)
    ref = torch.empty((n, 128), device=dev)
    g = graph.c_struct()
    ws = graph.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)))
    stream = torch.cuda.current_stream(dev).cuda_stream

    def default():
        _C.check(lib.egc_layer_forward_packed(C.byref(g), C.byref(spec.c), x.data_ptr(), planes.data_ptr(), bcat.data_ptr(),
                                              bias.data_ptr(), bases.data_ptr(), ldb, weightings.data_ptr(), ref.data_ptr(),
                                              ws.data_ptr(), ws.numel(), stream), "layer")
    default()
    t_default = timed(default)
    # bases-only planes (w_cols = 0) and weightings planes: the GEMM of [32 dummy basis columns | comb weights] -- the kernels
    # write `bases` and `weightings` blocks, the weightings block carries the bias; 32 columns is the narrowest bases block
    wb = wcat[:, :fg].contiguous()
    sb = SimpleNamespace(f_in=128, f_g=fg, w_cols=0, ldb=ldb)
    planes_b = pack_weights(sb, wb)
    ww = torch.cat([wcat[:, :32], wcat[:, fg:]], dim=1).contiguous()
    sw = SimpleNamespace(f_in=128, f_g=32, w_cols=W, ldb=32)
    planes_w = pack_weights(sw, ww)
    print(f"config 2, default two-launch layer: {t_default * 1e3:.1f} us")
    for K in (2, 4, 8, 16, 32):
        rows = -(-n // K)
        rows = (rows + 63) & ~63
        wbuf = torch.empty((rows, W), device=dev)
        junk = torch.empty((rows, 32), device=dev)

        def chunked():
            _C.check(lib.egc_basis_transform_packed(x.data_ptr(), planes_b.data_ptr(), None, n, 128, fg, 0, bases.data_ptr(), ldb,
                                                    None, stream), "gemm bases")
            for k in range(K):
                lo, hi = k * rows, min(n, (k + 1) * rows)
                if lo >= hi:
                    break
                _C.check(lib.egc_basis_transform_packed(x.data_ptr() + lo * 128 * 4, planes_w.data_ptr(), bcat.data_ptr(), hi - lo,
                                                        128, 32, W, junk.data_ptr(), 32, wbuf.data_ptr(), stream), "gemm w")
                # the rows kernel indexes weightings by the GLOBAL row: hand it the buffer's base moved back by `lo` rows
                _C.check(lib.egc_aggregate_combine_rows_f32(C.byref(g), C.byref(spec.c), bases.data_ptr(), ldb,
                                                            wbuf.data_ptr() - lo * W * 4, bias.data_ptr(), out.data_ptr(), lo, hi,
                                                            ws.data_ptr(), ws.numel(), stream), "agg rows")
        chunked()
        torch.cuda.synchronize()
        err = float((out - ref).abs().max() / ref.abs().max().clamp(min=1))
        t = timed(chunked, iters=20)
        print(f"  K = {K:2d} chunks of {rows} rows ({rows * W * 4 / 1e6:.1f} MB of weightings in flight): {t * 1e3:.1f} us"
              f"   rel err vs default {err:.1e}")


if __name__ == "__main__":
    main()

"""Repeat the one-launch backward on the same batch and report any call whose gradients leave the CSR path's by more than
rounding (development aid: looks for races; the parity tests are tests/test_fused_bwd_gpu.py)."""
import os, sys
import torch
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import egc_amd
from test_batch_tile_gpu import _messy_batch
dev = torch.device("cuda:0")
reps = int(os.environ.get("EGC_REPS", "200"))
kinds = {
    "ns": lambda: egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4),
    "lay": lambda: egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=["symadd", "max", "mean"]),
    "d64": lambda: egc_amd.EGConv(64, 64, aggrs=["sum", "max"], num_heads=4, num_bases=4),
}
for name, mk in kinds.items():
    for seed in (34, 7):
        ei, n, ptr = _messy_batch(seed, max_size=80)
        torch.manual_seed(0)
        conv = mk().to(dev).train()
        fin = conv.in_channels
        x0 = torch.randn(n, fin, device=dev); go = torch.randn(n, conv.out_channels, device=dev)
        def once(g):
            conv.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_(True)
            out = conv(x, g) if isinstance(conv, egc_amd.EGConv) else conv(x=x, edge_index=g)
            out.backward(go)
            return [x.grad.detach().clone()] + [p.grad.detach().clone() for p in conv.parameters()]
        ref = once(ei.to(dev))
        bad = 0; worst = 0.0
        for i in range(reps):
            g = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=80, num_nodes=n)
            got = once(g); g.check()
            e = max(float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)) for a, b in zip(got, ref))
            worst = max(worst, e)
            if e > 2e-5:
                bad += 1
                if bad <= 3:
                    d = (got[0] - ref[0]).abs().max(dim=1).values / ref[0].abs().max()
                    rows = (d > 2e-5).nonzero().flatten().cpu().numpy()
                    print(f"  {name} seed {seed} call {i}: err {e:.2e}; dx rows off: {len(rows)} {rows[:12]}; per-param " +
                          " ".join(f"{float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.1e}" for a, b in zip(got[1:], ref[1:])), flush=True)
        print(f"{name} seed {seed} N={n}: {reps} calls, {bad} off, worst {worst:.2e}", flush=True)

"""Worker of tests/test_partition_gpu.py: two ranks (gloo, both on cuda:0 -- a functional arrangement, the product
path is RCCL) or ONE rank over RCCL itself (EGC_TEST_BACKEND=nccl: every collective of the setup and of the step goes
through the library the multi-GPU runs use) train one EGConv layer on a vertex-partitioned graph; forward output, input gradients and the
all-reduced parameter gradients must equal the single-device run."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import egc_amd  # noqa: E402
from egc_amd import partition as P  # noqa: E402
from egc_amd.workloads import heavy_tailed_graph  # noqa: E402


def main():
    backend = os.environ.get("EGC_TEST_BACKEND", "gloo")     # "nccl" (= RCCL): world size 1 on a one-GPU box
    dev = torch.device("cuda:0")
    if backend == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    rank, world = dist.get_rank(), dist.get_world_size()
    n = 3000
    ei = heavy_tailed_graph(n, 20000, seed=4)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(64, 64, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev)
    with torch.no_grad():
        conv.bias.normal_()
    x_all = torch.randn(n, 64)
    g_all = torch.randn(n, 64)

    bounds = P.vertex_ranges(n, world)
    lo, hi = bounds[rank], bounds[rank + 1]
    ei_l, plan = P.build_distributed(P.local_edges(ei, lo, hi).to(dev), n, interior_first=(rank == 1))
    graph = egc_amd.CSRGraph.from_partition(ei_l, plan, global_max_index=n - 1)
    if backend == "nccl":
        # the layer's collectives, called explicitly (with one rank there is no halo and the layer skips them):
        # blocking and overlapped forms of the all-to-all-v and its reverse, on RCCL, with empty peer lists
        table = torch.randn(plan.n_local + plan.n_halo, 8, device=dev)
        keep = table.clone()
        plan.exchange(table)
        plan.exchange_finish(plan.exchange_start(table))
        back = plan.exchange_reverse(table)
        torch.cuda.synchronize()
        assert torch.equal(table, keep) and back.shape[0] == plan.send_idx.numel()
    order = plan.order if plan.order is not None else torch.arange(hi - lo, device=dev)
    x = x_all[lo:hi].to(dev)[order].requires_grad_(True)
    out = conv(x, graph)
    (out * g_all[lo:hi].to(dev)[order]).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in conv.named_parameters()}
    for k, v in grads.items():
        if backend == "nccl":
            dist.all_reduce(v)                 # replicated weights: the ranks' contributions add up
            grads[k] = v.cpu()
        else:
            c = v.cpu()
            dist.all_reduce(c)
            grads[k] = c
    out_nat = torch.empty_like(out); out_nat[order] = out.detach()
    gx_nat = torch.empty_like(x.grad); gx_nat[order] = x.grad

    # single-device reference on this rank
    conv.zero_grad()
    xr = x_all.to(dev).requires_grad_(True)
    ref = conv(xr, ei.to(dev))
    (ref * g_all.to(dev)).sum().backward()

    def rel(a, b):
        return float((a - b).abs().max()) / max(1e-12, float(b.abs().max()))

    errs = {"out": rel(out_nat, ref.detach()[lo:hi]), "dx": rel(gx_nat, xr.grad[lo:hi])}
    for k, p in conv.named_parameters():
        errs[k] = rel(grads[k], p.grad.detach().cpu())
    bad = {k: v for k, v in errs.items() if v > 2e-5}   # two fp32 evaluations of the same sums (float atomics in the source kernel)
    print(f"rank {rank}: halo {plan.n_halo} rows, errors {errs}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

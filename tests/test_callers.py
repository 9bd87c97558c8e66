"""Caller-level tests (BASELINE config 1): a ZINC-style net built on the drop-in layer.
CPU part: the net's parameter counts reproduce the reference's known answers (output/pretrained.txt).
GPU part: forward and gradients of the whole net on the gfx950 layer == the same net on the CPU restatement."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import egc_amd
from callers import ZincStyleNet
from egc_amd.workloads import zinc_like_batch
from oracle import egc_torch_ref as tref


def _make(hidden, H, B, aggrs):
    return lambda d: egc_amd.EfficientGraphConv(d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs)


@pytest.mark.parametrize("hidden,H,B,aggrs,total", [
    (168, 8, 4, ["symadd"], 102861),            # EgcZincNet EGC-S, output/pretrained.txt:41
    (124, 4, 4, ["add", "std", "max"], 100385),  # EgcZincNet EGC-M, output/pretrained.txt:129
])
def test_zinc_net_parameter_counts_match_reference(hidden, H, B, aggrs, total):
    net = ZincStyleNet(hidden, _make(hidden, H, B, aggrs))
    assert sum(p.numel() for p in net.parameters()) == total


class _RefLayer(nn.Module):
    """The CPU restatement wearing the drop-in's parameters (shared by reference)."""

    def __init__(self, layer):
        super().__init__()
        self.layer = layer

    def forward(self, x, edge_index):
        m = self.layer
        return tref.efficient_graph_conv_forward(
            x, edge_index.numpy(), list(m.bases_weight), m.comb_weights.weight, m.comb_weights.bias, m.bias,
            m.num_heads, [a.aggr_fun for a in m.aggs], softmax=m.softmax_weights, sigmoid=m.sigmoid_weights,
            hardtanh=m.hardtanh_weights, add_self_loops=m.add_self_loops)


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs", [
    (128, 1, 1, ["add"]),                 # BASELINE config 1 as written (plumbing shape)
    (168, 8, 4, ["symadd"]),              # the reference's actual EGC-S ZINC net
    (124, 4, 4, ["add", "std", "max"]),   # the reference's EGC-M ZINC net
])
def test_zinc_net_forward_and_gradients_match_cpu_restatement(hidden, H, B, aggrs):
    import copy
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(128, seed=3)
    n_graphs = int(batch.max()) + 1
    torch.manual_seed(0)
    net = ZincStyleNet(hidden, _make(hidden, H, B, aggrs))
    with torch.no_grad():
        for conv in net.convs:
            conv.bias.normal_(std=0.1)
    ref = copy.deepcopy(net).double()
    ref.convs = nn.ModuleList([_RefLayer(c) for c in ref.convs])
    ref32 = copy.deepcopy(net)   # the same restatement in fp32: calibrates how much fp32 rounding a 4-layer
    ref32.convs = nn.ModuleList([_RefLayer(c) for c in ref32.convs])  # net with 1/(2 std) gradients carries
    target = torch.randn(n_graphs, 1)
    # training-mode step (BatchNorm batch statistics, as in zinc/configs.py:53-72): L1 loss, backward
    net = net.to(dev).train()
    ref = ref.train()
    out = net(atom.to(dev), ei.to(dev), batch.to(dev), n_graphs)
    loss = (out - target.to(dev)).abs().mean()
    loss.backward()
    out_ref = ref(atom, ei, batch, n_graphs)
    loss_ref = (out_ref - target.double()).abs().mean()
    loss_ref.backward()
    ref32.train()
    (ref32(atom, ei, batch, n_graphs) - target).abs().mean().backward()

    def rel(a, b):
        return float((a.detach().cpu().double() - b.detach().double()).abs().max()) / max(1e-6, float(b.abs().max()))

    scale = max(1.0, float(out_ref.detach().abs().max()))
    assert float((out.detach().cpu().double() - out_ref.detach()).abs().max()) / scale <= 5e-5
    for got, want, cal in [
        (net.embedding.weight.grad, ref.embedding.weight.grad, ref32.embedding.weight.grad),
        (net.convs[0].comb_weights.weight.grad, ref.convs[0].layer.comb_weights.weight.grad,
         ref32.convs[0].layer.comb_weights.weight.grad),
        (net.convs[3].bases_weight[0].grad, ref.convs[3].layer.bases_weight[0].grad,
         ref32.convs[3].layer.bases_weight[0].grad),
    ]:
        # the HIP path may not be further from float64 than a few times the fp32 CPU restatement is.
        # Atoms of one type share an embedding row, so neighbourhoods hold exact duplicates: relu(var) sits
        # at its kink and max has near-ties, and WHICH side fp32 rounding lands on differs between any two
        # fp32 evaluations (the fp32 CPU restatement itself is 2.4e-2 from float64 on a single such layer)
        floor = 2e-3 if set(aggrs) & {"std", "max", "min"} else 2e-4   # (fixture-calibrated bounds: tests/test_nets_golden.py)
        assert rel(got, want) <= max(floor, 4.0 * rel(cal, want)), (rel(got, want), rel(cal, want))


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs", [(128, 8, 4, ["symadd", "max", "mean"]), (168, 8, 4, ["symadd"]),
                                              (42, 6, 3, ["add", "std"])])   # register / padded register / generic kernels
def test_fused_block_equals_conv_bn_relu_residual(hidden, H, B, aggrs):
    """FusedEGCBlock (eval): the BatchNorm1d-eval affine map, ReLU and the residual add inside the kernel's store
    == the reference nets' separate conv -> bn -> relu -> + identity (zinc/models.py:66-72)."""
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(64, seed=5)
    torch.manual_seed(1)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs).to(dev)
    bn = nn.BatchNorm1d(hidden).to(dev)
    with torch.no_grad():
        bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.normal_(); bn.bias.normal_()
        conv.bias.normal_()
    x = torch.randn(n, hidden, device=dev)
    ei = ei.to(dev)
    for relu, residual, use_bn in [(True, True, True), (False, True, True), (True, False, False), (False, False, True)]:
        block = egc_amd.FusedEGCBlock(conv, bn if use_bn else None, relu=relu, residual=residual).eval()
        with torch.no_grad():
            got = block(x, ei)
            ref = block._plain(x, ei)
        scale = max(1.0, float(ref.abs().max()))
        assert float((got - ref).abs().max()) / scale <= 1e-5, (relu, residual, use_bn)
    # training mode: batch statistics (test_fused_block_training_tail_equals_batch_norm_relu_residual); differentiable
    block = egc_amd.FusedEGCBlock(conv, bn).train()
    xg = x.clone().requires_grad_(True)
    block(xg, ei).sum().backward()
    assert xg.grad is not None and bool(torch.isfinite(xg.grad).all())


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs", [(128, 8, 4, ["symadd", "max", "mean"]), (168, 8, 4, ["symadd"]),
                                              (42, 6, 3, ["add", "std"])])   # 42 % 4 != 0: the plain sequence
@pytest.mark.parametrize("relu,residual,affine,momentum", [(True, True, True, 0.1), (False, True, True, None),
                                                           (True, False, False, 0.1)])
def test_fused_block_training_tail_equals_batch_norm_relu_residual(hidden, H, B, aggrs, relu, residual, affine, momentum):
    """FusedEGCBlock (training): conv -> BatchNorm1d on batch statistics -> ReLU -> + identity in two streaming passes
    each way (egc_tail.hip) == the reference nets' separate operators (zinc/models.py:66-72): output, gradients
    w.r.t. the input and every parameter, running statistics."""
    import copy
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(64, seed=7)
    torch.manual_seed(3)
    conv = egc_amd.EfficientGraphConv(hidden, hidden, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs).to(dev)
    bn = nn.BatchNorm1d(hidden, affine=affine, momentum=momentum).to(dev)
    with torch.no_grad():
        if affine:
            bn.weight.normal_(); bn.bias.normal_()
        conv.bias.normal_()
    conv_r, bn_r = copy.deepcopy(conv), copy.deepcopy(bn)
    x = torch.randn(n, hidden, device=dev)
    gout = torch.randn(n, hidden, device=dev)
    ei = ei.to(dev)
    block = egc_amd.FusedEGCBlock(conv, bn, relu=relu, residual=residual).train()
    ref_block = egc_amd.FusedEGCBlock(conv_r, bn_r, relu=relu, residual=residual).train()
    for step in range(2):                       # two steps: the running statistics move twice
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        got = block(xa, ei)
        ref = ref_block._plain(xb, ei)
        (got * gout).sum().backward()
        (ref * gout).sum().backward()
        scale = max(1.0, float(ref.detach().abs().max()))
        assert float((got.detach() - ref.detach()).abs().max()) / scale <= 1e-5, step

        def close(a, b, what, tol=2e-5):
            assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), (what, step)
        close(xa.grad, xb.grad, "x.grad")
        for (name, pa), (_, pb) in zip(list(conv.named_parameters()) + list(bn.named_parameters()),
                                       list(conv_r.named_parameters()) + list(bn_r.named_parameters())):
            # PyTorch's own fp32 BatchNorm backward is the yardstick here: both sides sum 10^3 float32 products
            close(pa.grad, pb.grad, name, tol=1e-4)
            pa.grad = None; pb.grad = None
        close(bn.running_mean, bn_r.running_mean, "running_mean")
        close(bn.running_var, bn_r.running_var, "running_var")
        assert int(bn.num_batches_tracked) == int(bn_r.num_batches_tracked) == step + 1


@pytest.mark.gpu
@pytest.mark.parametrize("relu,p", [(True, 0.2), (False, 0.5)])
def test_fused_block_training_tail_with_the_arxiv_nets_dropout(relu, p):
    """The ogbn-arxiv net's block (arxiv/norm_models.py:34-40): conv -> bn -> relu -> F.dropout -> + identity.  The fused
    tail carries the dropout mask through its three passes; against torch's operators applied with the SAME mask
    (FusedEGCBlock.last_keep_mask): output, input and parameter gradients.  Eval mode: dropout is the identity."""
    import copy
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(64, seed=9)
    torch.manual_seed(5)
    hidden = 128
    conv = egc_amd.EGConv(hidden, hidden, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4).to(dev)
    bn = nn.BatchNorm1d(hidden).to(dev)
    with torch.no_grad():
        bn.weight.normal_(); bn.bias.normal_(); conv.bias.normal_()
    conv_r, bn_r = copy.deepcopy(conv), copy.deepcopy(bn)
    x = torch.randn(n, hidden, device=dev)
    gout = torch.randn(n, hidden, device=dev)
    ei = ei.to(dev)
    block = egc_amd.FusedEGCBlock(conv, bn, relu=relu, residual=True, dropout=p).train()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    got = block(xa, ei)
    keep = block.last_keep_mask
    assert keep is not None and keep.dtype == torch.uint8 and keep.shape == got.shape
    frac = float(keep.float().mean())
    assert abs(frac - (1.0 - p)) < 0.02, frac
    h = bn_r(conv_r(xb, ei))
    h = torch.relu(h) if relu else h
    ref = h * keep.float() / (1.0 - p) + xb
    (got * gout).sum().backward()
    (ref * gout).sum().backward()
    scale = max(1.0, float(ref.detach().abs().max()))
    assert float((got.detach() - ref.detach()).abs().max()) / scale <= 1e-5

    def close(a, b, what, tol):
        assert float((a - b).abs().max()) <= tol * max(1.0, float(b.abs().max())), what
    close(xa.grad, xb.grad, "x.grad", 2e-5)
    for (name, pa), (_, pb) in zip(list(conv.named_parameters()) + list(bn.named_parameters()),
                                   list(conv_r.named_parameters()) + list(bn_r.named_parameters())):
        close(pa.grad, pb.grad, name, 1e-4)
    close(bn.running_mean, bn_r.running_mean, "running_mean", 2e-5)
    # a second forward draws a new mask; the same seed reproduces it
    torch.manual_seed(11); block(x, ei); k1 = block.last_keep_mask.clone()
    torch.manual_seed(11); block(x, ei); k2 = block.last_keep_mask
    assert torch.equal(k1, k2) and not torch.equal(k1, keep)
    block.eval(); bn_r.eval()
    with torch.no_grad():
        e_got = block(x, ei)
        e_ref = x + (torch.relu(bn_r(conv_r(x, ei))) if relu else bn_r(conv_r(x, ei)))
    # (the two BatchNorms saw a different number of training batches above: compare with the block's own statistics)
    bn_r.load_state_dict(bn.state_dict())
    with torch.no_grad():
        e_ref = x + (torch.relu(bn_r(conv_r(x, ei))) if relu else bn_r(conv_r(x, ei)))
    assert float((e_got - e_ref).abs().max()) <= 1e-5 * max(1.0, float(e_ref.abs().max()))


@pytest.mark.gpu
def test_fused_block_with_a_separate_identity_as_in_the_cifar_net():
    """cifar/models.py:64-71 drops out the layer's INPUT and adds the undropped activations back:
    block(drop(x), edge_index, identity=x) in training (two-pass tail) and in eval mode (tail in the kernel's store)."""
    import copy
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(48, seed=4)
    torch.manual_seed(2)
    hidden = 64
    conv = egc_amd.EfficientGraphConv(hidden, hidden, num_heads=4, num_bases=4, softmax_weights=False, aggrs=["symadd", "max"]).to(dev)
    bn = nn.BatchNorm1d(hidden).to(dev)
    conv_r, bn_r = copy.deepcopy(conv), copy.deepcopy(bn)
    ei = ei.to(dev)
    x = torch.randn(n, hidden, device=dev)
    xd = torch.nn.functional.dropout(x, 0.3, True)
    gout = torch.randn(n, hidden, device=dev)
    block = egc_amd.FusedEGCBlock(conv, bn).train()
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    mask = (xd != 0).float() / 0.7
    got = block(xa * mask, ei, identity=xa)
    ref = torch.relu(bn_r(conv_r(x=xb * mask, edge_index=ei))) + xb
    (got * gout).sum().backward(); (ref * gout).sum().backward()
    tol = lambda b: 2e-5 * max(1.0, float(b.abs().max()))
    assert float((got.detach() - ref.detach()).abs().max()) <= tol(ref.detach())
    assert float((xa.grad - xb.grad).abs().max()) <= tol(xb.grad)
    block.eval(); bn_r.eval()
    with torch.no_grad():
        got = block(xd, ei, identity=x)
        ref = torch.relu(bn_r(conv_r(x=xd, edge_index=ei))) + x
    assert float((got - ref).abs().max()) <= tol(ref)


@pytest.mark.gpu
def test_fused_block_eval_affine_follows_the_batch_norm_state():
    """The eval-mode affine pair is cached per state of the BatchNorm module: in-place updates of its buffers or
    parameters (a training step in between, load_state_dict) must show in the next eval forward.  (Keyed on the tensors'
    version counters, like the layers' packed weights: writes through ``.data`` bypass those by design.)"""
    dev = torch.device("cuda:0")
    atom, ei, n, batch = zinc_like_batch(16, seed=1)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(32, 32, aggrs=["sum", "max"], num_heads=4, num_bases=2).to(dev)
    bn = nn.BatchNorm1d(32).to(dev)
    block = egc_amd.FusedEGCBlock(conv, bn).eval()
    x, ei = torch.randn(n, 32, device=dev), ei.to(dev)

    def ref():
        with torch.no_grad():
            return x + torch.relu(bn(conv(x, ei)))
    with torch.no_grad():
        for change in (lambda: None, lambda: bn.running_mean.normal_(), lambda: bn.weight.mul_(1.7),
                       lambda: bn.load_state_dict({k: v + 0.25 if v.dtype.is_floating_point else v for k, v in bn.state_dict().items()})):
            change()
            got, want = block(x, ei), ref()
            assert float((got - want).abs().max()) <= 1e-5 * max(1.0, float(want.abs().max()))


@pytest.mark.gpu
def test_global_mean_pool_matches_index_add():
    dev = torch.device("cuda:0")
    _, _, n, batch = zinc_like_batch(200, seed=2)
    x = torch.randn(n, 77, device=dev)
    batch = batch.to(dev)
    n_graphs = int(batch.max()) + 1
    got = egc_amd.global_mean_pool(x, batch)
    ref = torch.zeros(n_graphs, 77, device=dev).index_add_(0, batch, x) / torch.bincount(batch, minlength=n_graphs).view(-1, 1)
    assert float((got - ref).abs().max()) <= 1e-5
    assert egc_amd.global_mean_pool(x, batch, size=n_graphs + 3).shape == (n_graphs + 3, 77)   # trailing empty graphs -> 0
    empty = egc_amd.global_mean_pool(x[:0], batch[:0], size=2)                                 # no nodes at all
    assert empty.shape == (2, 77) and not empty.any()
    # differentiable: d x[r] = d out[batch[r]] / count of its graph (trailing empty graphs contribute nothing)
    xg = x.clone().requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    go = torch.randn(n_graphs + 3, 77, device=dev)
    egc_amd.global_mean_pool(xg, batch, size=n_graphs + 3).backward(go)
    cnt = torch.bincount(batch, minlength=n_graphs + 3).clamp(min=1).view(-1, 1)
    (torch.zeros(n_graphs + 3, 77, device=dev).index_add(0, batch, xr) / cnt).backward(go)
    assert float((xg.grad - xr.grad).abs().max()) <= 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("momentum", [0.1, None])
@pytest.mark.parametrize("n", [300, 4000])        # 3 and 32 partial blocks: both inside the one-launch form's 64
def test_batch_norm_statistics_in_one_launch_equal_the_two_launches(n, momentum, monkeypatch):
    """The opt-in one-launch form of the BatchNorm statistics step (egc_bn_forward_stats_f32 / egc_bn_backward_stats_f32 with a
    sync word: the last block adds the partials in the finalize kernels' order) gives the SAME BITS as the two launches --
    outputs, every gradient, running statistics, num_batches_tracked -- over several steps (the sync word returns to zero)."""
    import copy
    import egc_amd
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(n)
    ei = torch.from_numpy(rng.integers(0, n, size=(2, 6 * n))).to(dev)
    torch.manual_seed(0)
    conv = egc_amd.EGConv(64, 64, aggrs=["sum", "max", "symnorm"], num_heads=4, num_bases=4)
    bn = torch.nn.BatchNorm1d(64, momentum=momentum)
    blk_a = egc_amd.FusedEGCBlock(conv, bn).to(dev).train()
    blk_b = copy.deepcopy(blk_a)
    x = torch.randn(n, 64, device=dev)
    gout = torch.randn(n, 64, device=dev)

    def steps(blk):
        outs = []
        for _ in range(3):
            xi = x.clone().requires_grad_(True)
            blk.zero_grad()
            out = blk(xi, ei)
            out.backward(gout)
            outs.append([out.detach().clone(), xi.grad.clone()] + [p.grad.clone() for p in blk.parameters()])
        return outs, [blk.bn.running_mean.clone(), blk.bn.running_var.clone(), blk.bn.num_batches_tracked.clone()]
    two = steps(blk_a)
    monkeypatch.setenv("EGC_BN_ONE_LAUNCH", "1")
    one = steps(blk_b)
    for sa, sb in zip(two[0], one[0]):
        for a, b in zip(sa, sb):
            assert torch.equal(a, b)
    for a, b in zip(two[1], one[1]):
        assert torch.equal(a, b)
    assert int(blk_b._bn_sync.item()) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("hidden,H,B,aggrs", [(304, 8, 8, ["symadd"]), (300, 4, 4, ["symadd", "min", "max"]), (128, 8, 4, ["symadd", "max", "mean"])])
def test_code_shaped_batch_against_the_oracle(hidden, H, B, aggrs):
    """ogbg-code2-shaped batches (egc_amd.workloads.code_like_batch: ASTs of ~125 nodes, up to 250, edges in the order the
    reference's augment_edge leaves them -- code/utils.py:74-135) through the reference's code nets' layers
    (run_pretrained.sh:47-48) handed over as a GraphBatch, full batch against the numpy oracle and every element against float64.
    Which kernel serves it is asserted: a 250-node graph's `bases` rows (1,216 - 1,280 bytes each at these widths) do not fit the
    LDS of a CU, so the 300 / 304-wide layers take the CSR path; the d = 128 layer's 160-row tiles do not hold a 250-node graph
    either and go through the two-launch tile path."""
    import egc_amd
    from egc_amd import workloads as wl
    from golden_util import elementwise_excess, float64_forward, oracle_forward, rel_err
    from oracle import egc_oracle as orc
    dev = torch.device("cuda:0")
    ei, n, batch = wl.code_like_batch(128, seed=3)
    sizes = torch.bincount(batch)
    ptr = torch.cat([torch.zeros(1, dtype=torch.int64), torch.cumsum(sizes, 0)])
    torch.manual_seed(hidden)
    layer = egc_amd.EfficientGraphConv(hidden, hidden, H, B, False, aggrs=aggrs)
    with torch.no_grad():
        layer.bias.normal_()
    x = torch.randn(n, hidden)
    meta = dict(kind="lay", fin=hidden, fout=hidden, H=H, B=B, aggrs=aggrs, softmax=False, sigmoid=False, hardtanh=False,
                add_self_loops=True, bias=True, sparse=False)
    g = dict(meta=meta, params={k: v.numpy() for k, v in layer.state_dict().items()}, x=x.numpy(), edge_index=ei.numpy())
    ref = oracle_forward(g, orc)
    layer = layer.to(dev).eval()
    gb = egc_amd.GraphBatch(ei.to(dev), ptr=ptr.to(dev), max_nodes=int(sizes.max()))
    with torch.no_grad():
        out = layer(x=x.to(dev), edge_index=gb).cpu().numpy()
    gb.check()
    ran = sorted({(k[-1] if isinstance(k[-1], str) else "tile") for k, v in gb._setups.items() if v})
    assert ran == ([] if hidden >= 300 else ["tile"]), ran          # [] = neither tile kernel: the CSR of the batch (gb.csr())
    assert (gb._csr is not None) == (hidden >= 300)
    assert rel_err(out, ref) <= 1e-5, rel_err(out, ref)
    assert elementwise_excess(out, float64_forward(g), 1e-5) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("n,c,one_launch", [(300, 64, False), (4000, 128, True), (52771, 224, False)])
def test_dh_column_sums_from_the_batch_norm_backward_step(n, c, one_launch, relu):
    """egc_bn_backward_stats_sums_f32: the column sums of the dh the elementwise pass writes -- the gradient of a conv bias in
    front of the BatchNorm -- from the sums the step already holds, against the float64 sum of the ACTUAL dh: within the rounding
    of the dh elements themselves (the float32 sum autograd forms is no closer; a wrong term would be off by n |coef|), and
    equal to the exact sum with the stored coefficients, which is zero but for those coefficients' rounding."""
    from egc_amd import _C, functional as Fn
    lib = _C.load()
    dev = torch.device("cuda:0")
    torch.manual_seed(n + c)
    h = (torch.randn(n, c, device=dev) * torch.logspace(-1, 1, c, device=dev) + torch.linspace(-3, 3, c, device=dev)).contiguous()
    dout = torch.randn(n, c, device=dev)
    gamma, beta = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
    n_parts = max(1, min(1024, (n + 127) // 128))
    parts = torch.empty((n_parts, 2, c), dtype=torch.float64, device=dev)
    stats = torch.empty((3, c), dtype=torch.float64, device=dev)
    affine = torch.empty((2, c), dtype=torch.float32, device=dev)
    stream = Fn._stream_ptr(dev)
    _C.check(lib.egc_bn_forward_stats_f32(h.data_ptr(), n, c, parts.data_ptr(), n_parts, None, None, gamma.data_ptr(), beta.data_ptr(), 1e-5,
                                          stats.data_ptr(), affine.data_ptr(), None, None, 0.1, None, None, stream), "egc_bn_forward_stats_f32")
    out5 = torch.empty((5, c), dtype=torch.float32, device=dev)
    sums = torch.full((c,), float("nan"), device=dev)
    sync = torch.zeros(1, dtype=torch.int32, device=dev) if one_launch else None
    _C.check(lib.egc_bn_backward_stats_sums_f32(dout.data_ptr(), h.data_ptr(), affine[0].data_ptr(), affine[1].data_ptr(), int(relu), None, 1.0, n, c,
                                                parts.data_ptr(), n_parts, None, stats.data_ptr(), gamma.data_ptr(), out5.data_ptr(),
                                                sums.data_ptr(), sync.data_ptr() if one_launch else None, stream), "egc_bn_backward_stats_sums_f32")
    dh = torch.empty_like(h)
    _C.check(lib.egc_affine_act_backward_f32(dout.data_ptr(), h.data_ptr(), affine[0].data_ptr(), affine[1].data_ptr(), int(relu), None, 1.0,
                                             out5[2].data_ptr(), out5[3].data_ptr(), out5[4].data_ptr(), n, c, dh.data_ptr(), None, stream),
             "egc_affine_act_backward_f32")
    torch.cuda.synchronize()
    ref = dh.double().sum(0)
    # every dh element carries up to ~2 ulp of its three terms' sizes: |cg g| + |ch h| + |c1|
    g = dout * ((h * affine[0] + affine[1]) > 0) if relu else dout
    terms = (out5[2].abs() * g.abs() + out5[3].abs() * h.abs() + out5[4].abs()).double()
    # (the roundings do not average out: adding coef_1 last is biased on structured data -- measured -6.6e-8 per element against an
    # rms of 1.2e-7 at 52,771 x 224 with the ReLU mask -- so the float32 dh sums drift by up to half an ulp per ROW from the exact sum)
    bound = 2.0 ** -24 * n * terms.max(0).values + 1e-30
    assert bool(((sums.double() - ref).abs() <= bound).all()), float(((sums.double() - ref).abs() / bound).max())
    # ... and the true value is zero: what is left is the rounding of the three float32 coefficients (coef_h h and coef_1 cancel
    # where a channel's mean is large against its spread), a few ulp of the terms' total
    assert bool((sums.double().abs() <= 2.0 ** -22 * terms.sum(0)).all()), float((sums.double().abs() / terms.sum(0)).max())
    exact = out5[2].double() * g.double().sum(0) + out5[3].double() * h.double().sum(0) + n * out5[4].double()
    assert bool(((sums.double() - exact).abs() <= 2.0 ** -40 * terms.sum(0) + 2.0 ** -23 * sums.double().abs()).all())
    out5b = torch.empty_like(out5)
    _C.check(lib.egc_bn_backward_stats_f32(dout.data_ptr(), h.data_ptr(), affine[0].data_ptr(), affine[1].data_ptr(), int(relu), None, 1.0, n, c,
                                           parts.data_ptr(), n_parts, None, stats.data_ptr(), gamma.data_ptr(), out5b.data_ptr(), None, stream),
             "egc_bn_backward_stats_f32")
    assert torch.equal(out5, out5b)

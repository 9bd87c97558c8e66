"""The reference's own NETS as fixtures (tests/golden/net_*.npz: EgcZincNet, EgcArxivNet, mag EGC, forward + backward,
float32 and float64, captured from /root/reference by make_golden_nets.py; relgrad_*.npz: REGConv's gradients).

CPU: the test-side counterparts of the nets (tests/callers.py, same module names: the fixtures' state dicts load with
strict=True) over the differentiable CPU restatement of the layer (oracle/egc_torch_ref.py) reproduce the float64
fixtures to 1e-9 -- callers and restatement are pinned to the reference's code, not to each other.
GPU: the same nets on the gfx950 layers -- FusedEGCBlock for conv -> BatchNorm -> ReLU -> + x, the segmented-mean
readout, REGConv's backward -- against the float64 fixtures, bounded by max(1e-5, 5 x the distance between the
reference's OWN float32 and float64 runs on that fixture) per parameter (MANIFEST_NETS.json), as test_backward_golden.py
does for single layers."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import egc_amd
from callers import ArxivNetLike, MagNetLike, ZincNetLike
from golden_util import GOLDEN_DIR
from oracle import egc_torch_ref as tref


def _names(prefix):
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


def _load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return z, json.loads(bytes(z["meta"]).decode())


def _build(meta):
    H, B, aggrs = meta["H"], meta["B"], meta["aggrs"]
    if meta["net"] == "EgcZincNet":
        return ZincNetLike(meta["hidden"], meta["layers"], lambda d: egc_amd.EfficientGraphConv(
            d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs), residual=meta["residual"])
    if meta["net"] == "EgcArxivNet":
        return ArxivNetLike(meta["hidden"], meta["layers"], lambda d: egc_amd.EfficientGraphConv(
            d, d, num_heads=H, num_bases=B, softmax_weights=False, aggrs=aggrs), residual=meta["residual"])
    return MagNetLike(meta["hidden"], meta["layers"], lambda a, b: egc_amd.EGConv(
        a, b, aggrs=aggrs, num_heads=H, num_bases=B, cached=True), out_true=meta["out_true"])


def _state(z):
    return {k[len("param:"):]: torch.from_numpy(z[k]) for k in z.files if k.startswith("param:")}


def _restated(conv, x, edge_index):
    """The CPU restatement wearing the module's parameters (whatever their dtype)."""
    if hasattr(conv, "aggs"):
        return tref.efficient_graph_conv_forward(
            x, edge_index.numpy(), list(conv.bases_weight), conv.comb_weights.weight, conv.comb_weights.bias, conv.bias,
            conv.num_heads, [a.aggr_fun for a in conv.aggs], softmax=conv.softmax_weights, sigmoid=conv.sigmoid_weights,
            hardtanh=conv.hardtanh_weights, add_self_loops=conv.add_self_loops)
    return tref.egconv_forward(x, edge_index.numpy(), conv.bases_weight, conv.comb_weight.weight, conv.comb_weight.bias,
                               conv.bias, conv.num_heads, conv.num_bases, list(conv.aggregators),
                               add_self_loops=conv.add_self_loops)


def _call(net, meta, z, dev, dtype, train, **kw):
    ei = torch.from_numpy(z["in:edge_index"])
    if meta["net"] == "EgcZincNet":
        graph = kw.pop("graph", None)
        out = net(torch.from_numpy(z["in:atom"]).to(dev), graph if graph is not None else ei.to(dev),
                  torch.from_numpy(z["in:batch"]).to(dev), meta["n_graphs"], **kw)
        return out, None
    x = torch.from_numpy(z["in:x"]).to(dev, dtype).requires_grad_(train)
    if meta["net"] == "mag EGC":
        n = meta["n"]
        if kw.get("conv_fn") is _restated:
            # adj_t semantics for the COO restatement: fill_diag gives EVERY row a loop (optimized_layers.py:168-175);
            # a loop on the last node makes add_remaining_self_loops infer the same N (loops are replaced, not kept twice)
            arg = torch.cat([ei, torch.tensor([[n - 1], [n - 1]])], dim=1)
        else:
            arg = egc_amd.SparseTensor(row=ei[1].to(dev), col=ei[0].to(dev), sparse_sizes=(n, n))
        return net(x, arg, **{k: v for k, v in kw.items() if k == "conv_fn"}), x
    return net(x, ei.to(dev) if dev.type == "cuda" else ei, **kw), x


@pytest.mark.parametrize("name", _names("net_"))
def test_callers_and_restatement_reproduce_the_reference_nets_in_float64(name):
    z, meta = _load(name)
    net = _build(meta)
    net.load_state_dict(_state(z), strict=True)          # the reference's state dict, key for key
    net = net.double()
    cpu = torch.device("cpu")
    net.eval()
    with torch.no_grad():
        out_eval, _ = _call(net, meta, z, cpu, torch.float64, False, conv_fn=_restated)
    net.train()
    out, leaf = _call(net, meta, z, cpu, torch.float64, True, conv_fn=_restated)
    out.backward(torch.from_numpy(z["gout"]).double())

    def rel(a, b):
        b = torch.from_numpy(b)
        return float((a.detach() - b).abs().max() / max(1.0, float(b.abs().max())))
    # 1e-9, except where gcn_norm's FLOAT32 constants enter (PyG forms deg^-1/2 and the edge weight in float32 whatever
    # the features' dtype; two float32 evaluation orders of dis[i] * dis[j] differ by one ulp = 6e-8): 1e-7 there
    tol = 1e-7 if any(a in ("symnorm", "symadd") for a in meta["aggrs"]) else 1e-9
    assert rel(out_eval, z["out_eval64"]) <= tol and rel(out, z["out_train64"]) <= tol
    gscale = meta["grad_scale"]
    for k, p in net.named_parameters():
        want = z[f"grad64:{k}"]
        assert float((p.grad - torch.from_numpy(want)).abs().max()) <= tol * max(gscale, 1.0), k
    if leaf is not None:
        assert rel(leaf.grad, z["grad_x64"]) <= tol


def _fuse(conv, bn, residual):
    return egc_amd.FusedEGCBlock(conv, bn, relu=True, residual=residual)


@pytest.mark.gpu
@pytest.mark.parametrize("fused", [False, True, "batch"])
@pytest.mark.parametrize("name", _names("net_"))
def test_nets_on_the_hip_layers_match_the_reference_float64(name, fused):
    """fused = "batch": the ZINC nets with the PyG batch handed over as an egc_amd.GraphBatch (node offsets from the batch
    vector) -- the one-launch forward of the batch path, and for net_zinc_b64 (four bases of 16 channels) the one-launch
    BACKWARD with the residual gradient joined in the launch (egc_layer_backward_batch_fused_f32)."""
    z, meta = _load(name)
    if fused and meta["net"] == "mag EGC":
        pytest.skip("the mag net has no BatchNorm / residual tail to fuse")
    if fused == "batch" and meta["net"] != "EgcZincNet":
        pytest.skip("a batch of whole graphs: the ZINC nets")
    dev = torch.device("cuda:0")
    net = _build(meta)
    net.load_state_dict(_state(z), strict=True)
    net = net.to(dev)
    kw = {}
    if fused:
        blocks = {}

        def fuse(conv, bn, residual):          # one block per (conv, bn) pair, sharing the net's modules
            key = id(conv)
            if key not in blocks:
                blocks[key] = _fuse(conv, bn, residual)
            blocks[key].train(bn.training)
            return blocks[key]
        kw["fuse"] = fuse
        if meta["net"] == "EgcZincNet":
            kw["pool"] = lambda x, batch, n_graphs: egc_amd.global_mean_pool(x, batch, n_graphs)
    gb = None
    if fused == "batch":
        sizes = np.bincount(z["in:batch"], minlength=meta["n_graphs"])
        gb = egc_amd.GraphBatch(torch.from_numpy(z["in:edge_index"]).to(dev), batch=torch.from_numpy(z["in:batch"]).to(dev),
                                num_graphs=meta["n_graphs"], max_nodes=int(sizes.max()))
        kw["graph"] = gb
    net.eval()
    with torch.no_grad():
        out_eval, _ = _call(net, meta, z, dev, torch.float32, False, **dict(kw))
    net.train()
    out, leaf = _call(net, meta, z, dev, torch.float32, True, **dict(kw))
    out.backward(torch.from_numpy(z["gout"]).to(dev))
    if gb is not None:
        gb.check()
        ran = {k[-1] for k, v in gb._setups.items() if isinstance(k, tuple) and isinstance(k[-1], str) and v}
        if name == "net_zinc_b64":
            assert "fused_bwd" in ran, ran          # the one-launch backward, not a fallback

    def rel(a, b):
        b = torch.from_numpy(b).double()
        return float((a.detach().cpu().double() - b).abs().max() / max(1.0, float(b.abs().max())))
    # outputs: the north star's 1e-5 (std / var nets: the reference's own float32 is the yardstick, see the manifest)
    d_out = max(1e-5, 5.0 * meta["f32_vs_f64_out_train"])
    assert rel(out_eval, z["out_eval64"]) <= d_out, rel(out_eval, z["out_eval64"])
    assert rel(out, z["out_train64"]) <= d_out, rel(out, z["out_train64"])
    gscale = meta["grad_scale"]
    for k, p in net.named_parameters():
        want = torch.from_numpy(z[f"grad64:{k}"]).double()
        bound = max(1e-5, 5.0 * meta["f32_vs_f64_grad"][k])
        wmax = float(want.abs().max())
        # a bias in front of a BatchNorm on batch statistics has an analytically ZERO gradient (1e-17 in float64): what any
        # float32 evaluation leaves there is cancellation noise of terms of the net's gradient scale -- held to that scale
        denom = gscale if wmax < 1e-6 * gscale else max(1e-2 * gscale, wmax)
        err = float((p.grad.cpu().double() - want).abs().max() / denom)
        assert err <= bound, (k, err, bound)       # (every mode, the batch path included: its launches are deterministic since round 6)
    if leaf is not None:
        assert rel(leaf.grad, z["grad_x64"]) <= max(1e-5, 5.0 * meta["f32_vs_f64_grad_max"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", _names("relgrad_"))
def test_regconv_backward_matches_the_reference_float64(name):
    """REGConv's gradients (every node type's features, every parameter) against the reference's own REGConv run in
    float64 on the inputs of the forward fixture (VERDICT r2 missing #4)."""
    from golden_util import load_rel_golden
    z, meta = _load(name)
    g = load_rel_golden(meta["forward_fixture"])
    m = g["meta"]
    dev = torch.device("cuda:0")
    conv = egc_amd.REGConv(m["fin"], m["fout"], m["H"], m["B"])
    conv.load_state_dict({k: torch.from_numpy(v) for k, v in g["params"].items()}, strict=True)
    conv = conv.to(dev).train()
    sizes = {k: g["x"][k].shape[0] for k in m["node_types"]}
    adj = {k: egc_amd.SparseTensor(row=torch.from_numpy(ei[1]).to(dev), col=torch.from_numpy(ei[0]).to(dev),
                                   sparse_sizes=(sizes[k[2]], sizes[k[0]])) for k, ei in g["ei"].items()}
    xs = {k: torch.from_numpy(g["x"][k]).to(dev).requires_grad_(True) for k in sizes}
    out = conv(xs, adj)
    sum((out[k] * torch.from_numpy(z[f"gout_{k}"]).to(dev)).sum() for k in sizes).backward()
    bound = max(1e-5, 5.0 * meta["f32_vs_f64_grad_max"])
    for k in sizes:
        want = torch.from_numpy(z[f"out64_{k}"])
        assert float((out[k].detach().cpu().double() - want).abs().max() / max(1.0, float(want.abs().max()))) <= 1e-5
        want = torch.from_numpy(z[f"grad_x64_{k}"])
        err = float((xs[k].grad.cpu().double() - want).abs().max() / max(1e-30, float(want.abs().max())))
        assert err <= bound, (k, err)
    for k, p in conv.named_parameters():
        want = torch.from_numpy(z[f"grad64:{k}"])
        err = float((p.grad.cpu().double() - want).abs().max() / max(1e-30, float(want.abs().max())))
        assert err <= bound, (k, err)

"""GPU parity of the fused-weightings launch (egc_aggregate_fusedw.hip): the combination
Linear computed inside the aggregate launch on the fp32 matrix cores, `weightings` never in memory.  Opt-in at BUILD time (EGC_WITH_FUSEDW=1 egc_amd/csrc/build.sh) and at run time
(EGC_FUSEDW=1); checked against the committed goldens that fall inside its envelope, the numpy oracle, and the
two-launch HIP path.

Tolerance (north_star): 1e-5, scale-relative (max |diff| / max(1, max |ref|))."""
import numpy as np
import pytest
import torch

from golden_util import golden_names, load_golden, oracle_forward, rel_err
from oracle import egc_oracle as orc
from test_parity_gpu import build_layer, run_layer

pytestmark = pytest.mark.gpu

TOL = 1e-5


@pytest.fixture(autouse=True)
def _needs_the_experimental_build():
    """The launch lives in the EGC_WITH_FUSEDW=1 build of the library only (round 4: it is slower than the two-launch path and
    left the default .so); in the default build egc_fused_supported() answers 0 and these tests have nothing to run."""
    import ctypes as C
    import egc_amd
    from egc_amd import _C
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4)
    if not _C.load().egc_fused_supported(C.byref(conv._spec_coo.c)):
        pytest.skip("libegc_hip.so built without EGC_WITH_FUSEDW=1")


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _spec_of(layer):
    return layer._spec_coo if hasattr(layer, "_spec_coo") else layer._spec


def test_goldens_inside_the_envelope_through_the_fused_launch(monkeypatch):
    from egc_amd.functional import fused_supported
    dev = _dev()
    monkeypatch.setenv("EGC_FUSEDW", "1")
    ran = 0
    for name in golden_names():
        g = load_golden(name)
        layer = build_layer(g["meta"], g["params"], dev)
        if not fused_supported(_spec_of(layer)):
            continue
        out = run_layer(layer, g, dev)
        assert rel_err(out, g["out"]) <= TOL, f"{name}: rel err {rel_err(out, g['out']):.3e}"
        ran += 1
    assert ran >= 2, "no golden exercised the fused launch"


def test_envelope():
    import egc_amd
    from egc_amd.functional import fused_supported
    ok = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"], num_heads=8, num_bases=4)
    assert fused_supported(ok._spec_coo)
    assert fused_supported(egc_amd.EGConv(100, 64, aggrs=["mean", "max"], num_heads=4, num_bases=4)._spec_coo)
    assert not fused_supported(egc_amd.EGConv(352, 352, aggrs=["symnorm"], num_heads=8, num_bases=4)._spec_coo)   # 44 slots
    assert not fused_supported(egc_amd.EGConv(128, 128, aggrs=["std"], num_heads=8, num_bases=4)._spec_coo)
    assert not fused_supported(egc_amd.EGConv(128, 128, aggrs=["sum"], num_heads=8, num_bases=2)._spec_coo)


CASES = [
    # n, e, fin, fout, H, aggrs, kind, long rows, act
    (50, 300, 128, 128, 8, ["sum", "mean", "max", "symnorm"], "opt", False, None),
    (1000, 9000, 128, 128, 8, ["sum", "mean", "max", "symnorm"], "opt", True, None),
    (1000, 9000, 128, 128, 8, ["symadd", "max", "mean"], "lay", True, None),
    (1000, 9000, 128, 64, 4, ["mean", "max"], "opt", True, None),
    (1003, 20000, 100, 128, 8, ["symnorm"], "opt", False, None),           # odd number of 16-wide k-steps, ragged last tile
    (777, 6000, 36, 128, 8, ["sum", "max"], "opt", True, "sigmoid"),
    (777, 6000, 64, 128, 8, ["add", "mean"], "lay", False, "hardtanh"),
    (5, 0, 128, 128, 8, ["sum", "mean", "max", "symnorm"], "opt", False, None),   # no edges at all
]


@pytest.mark.parametrize("n,e,fin,fout,H,aggrs,kind,long_rows,act", CASES)
def test_fused_launch_against_oracle_and_two_launch_path(n, e, fin, fout, H, aggrs, kind, long_rows, act, monkeypatch):
    import egc_amd
    dev = _dev()
    rng = np.random.default_rng(n + e)
    ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
    if long_rows and e > 900:
        ei[1, :700] = 3      # several chunks
        ei[1, 700:800] = 17  # a single chunk
    torch.manual_seed(1)
    if kind == "opt":
        layer = egc_amd.EGConv(fin, fout, aggrs=aggrs, num_heads=H, num_bases=4, sigmoid=(act == "sigmoid"))
    else:
        layer = egc_amd.EfficientGraphConv(fin, fout, H, 4, False, aggrs=aggrs, sigmoid_weights=(act == "sigmoid"),
                                           hardtanh_weights=(act == "hardtanh"))
    with torch.no_grad():
        layer.bias.normal_()
    x = rng.standard_normal((n, fin)).astype(np.float32)
    sd = {k: v.detach().numpy() for k, v in layer.state_dict().items()}
    if kind == "opt":
        ref = orc.egconv_forward(x, ei, sd["bases_weight"], sd["comb_weight.weight"], sd["comb_weight.bias"], sd["bias"],
                                 H, 4, aggrs, sigmoid=(act == "sigmoid"))
    else:
        ref = orc.efficient_graph_conv_forward(x, ei, [sd[f"bases_weight.{b}"] for b in range(4)], sd["comb_weights.weight"],
                                               sd["comb_weights.bias"], sd["bias"], H, aggrs,
                                               sigmoid_weights=(act == "sigmoid"), hardtanh_weights=(act == "hardtanh"))
    layer = layer.to(dev).eval()
    xd, eid = torch.from_numpy(x).to(dev), torch.from_numpy(ei).to(dev)
    call = (lambda: layer(xd, eid)) if kind == "opt" else (lambda: layer(x=xd, edge_index=eid))
    with torch.no_grad():
        two = call()
        monkeypatch.setenv("EGC_FUSEDW", "1")
        fused = call()
        again = call()
    torch.cuda.synchronize()
    assert rel_err(fused.cpu().numpy(), ref) <= TOL, rel_err(fused.cpu().numpy(), ref)
    assert rel_err(fused.cpu().numpy(), two.cpu().numpy()) <= TOL
    assert torch.equal(fused, again)          # deterministic, workspace left reusable


def test_fused_launch_with_the_callers_tail(monkeypatch):
    """BatchNorm(eval) -> ReLU -> + identity folded into the store (egc_post) on the fused launch."""
    import egc_amd
    from egc_amd.functional import PostOp, egc_layer_forward
    dev = _dev()
    n, e = 600, 5000
    torch.manual_seed(3)
    conv = egc_amd.EGConv(128, 128, aggrs=["sum", "mean", "max", "symnorm"]).to(dev).eval()
    x = torch.randn(n, 128, device=dev)
    ei = torch.randint(0, n, (2, e), device=dev)
    g = egc_amd.CSRGraph.from_edge_index(ei, n)
    wcat, bcat = conv._packed_weights()
    post = PostOp(scale=torch.rand(128, device=dev) + 0.5, shift=torch.randn(128, device=dev), residual=x, relu=True)
    with torch.no_grad():
        plain = egc_layer_forward(g, conv._spec_coo, x, wcat, bcat, conv.bias)
        want = torch.relu(plain * post.scale + post.shift) + x
        monkeypatch.setenv("EGC_FUSEDW", "1")
        got = egc_layer_forward(g, conv._spec_coo, x, wcat, bcat, conv.bias, post=post)
    assert rel_err(got.cpu().numpy(), want.cpu().numpy()) <= TOL

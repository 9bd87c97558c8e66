"""Randomised differential sweep on the GPU: random layer kinds, head / basis / channel counts (powers of two and
not), aggregator lists, weight nonlinearities, self-loop policies and graphs (hubs above the long-row threshold,
self-loops, duplicate edges, isolated nodes, exact ties, empty edge lists), HIP forward against the numpy oracle and,
for a quarter of the cases, every gradient against float64 autograd through the torch restatement.

Tolerance: 1e-5 (north_star) scale-relative for the forward; 1e-4 where the layer has `std` / `var`, whose
`sqrt(relu(E[x^2] - E[x]^2) + 1e-5)` amplifies last-bit differences between two correct fp32 evaluations (and the
2^-22 operand rounding of the split-precision GEMM) by up to 158x on (nearly) constant neighbourhoods (DESIGN.md
section 1).  Cases beyond 1e-4 -- seeds 101 / 109 / 118 / 202, kept from a sweep of 184 further seeds: a one-node graph
with a dozen self-loops under `std`, 1e-4 .. 2.4e-4; a row of eight neighbours, most of them identical, whose mean^2 / var
is large, 1.2e-4 -- must meet the criterion that does not depend on the evaluation order: against the same layer in
float64, of the order of the fp32 restatement's own error (x8, + 1e-5).  5e-4 for gradients (fp32 atomics vs float64).

Round 3, a further sweep (EGC_FUZZ_EXTRA_SEEDS=300-339: 4,800 cases, with the two-slots-per-lane kernel and the record
backward in the default path): 38 seeds clean; seed 303 (kept) has a case without std / var 1.4e-5 from the float32 ORACLE,
which is the oracle's own error -- it adds the 1,500 entries of a hub row one after the other (1.44e-5 from float64; HIP
3.2e-7) -- hence the float64 criterion for such cases too (HIP within 1e-5 of float64, no allowance); seed 337 (not kept: it
fails the x8 rule at 9.5) has ONE element of one std layer at 1.27e-5 from float64, a row of six tied neighbours whose
E[x^2] - E[x]^2 the restatement happens to cancel exactly (every other element of that layer: <= 2.6e-7).  Seeds 340-399:
59 clean; seed 377 (not kept) is the same thing at its plainest -- two rows whose three neighbours are IDENTICAL (true
variance 0): float32 leaves E[x^2] - E[x]^2 = +3 / +1 / 0 ulp(x^2) here and -1 / 0 / 0 in the restatement (the bits of the
GEMM's outputs decide), relu() hides the negative draws and sqrt(. + 1e-5) turns the positive ones into 1.1e-4 of std:
3.1e-5 of the output against the restatement's 3.1e-7.  Nothing evaluated in float32 can promise the sign of that
residue; what is promised is the formula of layers.py:203-216 with separately rounded squares, products and differences.
These draws are what the ELEMENT-WISE criterion is for (_stdvar_allowance): beyond 1e-5 and beyond 8 x the restatement, a std /
var layer must lie within 1e-5 + the spread that (4 + sqrt(deg)) ulp of E[x^2] in the variance produces in its own output,
computed in float64 from the layer's formula -- seeds 337 and 377 are kept under it.
Seeds 400-449 (final binary of the round): 47 clean; 405 and 410 are two more std layers over tied neighbours (2e-5); 413
(kept) is a GRADIENT 9.8e-4 from float64 that the float32 restatement shares to 4e-7 -- a near-tie of max / min that float32
and float64 resolve differently -- hence the float32 yardstick for such cases.
Under these criteria seeds 300-749 (450 seeds, 54,000 configurations) all pass on the final binary of the round.
EGC_FUZZ_DUMP=<dir> saves the inputs of failing cases."""
import numpy as np
import pytest
import torch

from golden_util import oracle_forward, rel_err
from oracle import egc_oracle as orc
from oracle import egc_torch_ref as tref

pytestmark = pytest.mark.gpu

# std / var layers: HIP vs float64 <= max(1e-5, STDVAR_K x the fp32 restatement's own error vs float64).  Measured over these
# seeds (gpurun_out/r3_fuzz24.log, r3_fuzz22.log -> profiles/r03_stdvar_gemm_precision.md): with the 24-bit-operand GEMM
# such layers run by default the worst ratio is 6.0 (two fp32 summation orders of E[x^2] - E[x]^2, amplified 158x: single
# draws of the same rounding noise); with the 22-bit fp16x2 GEMM (the default since round 4; EGC_GEMM_STDVAR_24BIT=1 restores the 24-bit form) one case of seed 118 sits at 28x.
# Round 4: the kernels accumulate the variance about the row's first entry (FAcc::sh, egc_aggregate_fast_dev.h) -- the same
# number without the cancellation -- and such layers run the default fp16x2 GEMM.  Over the committed seeds and 300-330 (44
# seeds, 5,280 configurations) every std / var layer is within 7e-7 of float64 where the float32 restatement is 1e-5 ... 3.7e-4
# off (HIP / restatement <= 0.05): the criterion is now HIP <= max(1e-5, 1 x the restatement's own error), and the element-wise
# allowance for float32's residue is only consulted with EGC_FUZZ_ALLOWANCE=1 (the 24-bit-GEMM / unshifted history above).
import os as _os
STDVAR_K = float(_os.environ.get("EGC_FUZZ_STDVAR_K", "1.0"))
STDVAR_ALLOWANCE = _os.environ.get("EGC_FUZZ_ALLOWANCE", "0") not in ("", "0")


def _truth64(tref, kind, layer, x, ei, H, B, names, flags, asl):
    """The layer in float64 through the torch restatement (parameters and input promoted)."""
    p64 = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    x64 = torch.from_numpy(x).double()
    with torch.no_grad():
        if kind == "opt":
            return tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"], p64["bias"],
                                       H, B, names, add_self_loops=asl, sigmoid=flags.get("sigmoid", False))
        return tref.efficient_graph_conv_forward(x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)], p64["comb_weights.weight"],
                                                 p64["comb_weights.bias"], p64["bias"], H, names, softmax=flags.get("softmax", False),
                                                 hardtanh=flags.get("hardtanh", False), sigmoid=flags.get("sigmoid", False),
                                                 add_self_loops=asl)


def _extra_seeds():
    """EGC_FUZZ_EXTRA_SEEDS="300-330" (or "7,8,9"): further seeds for a hunt, beside the committed ones."""
    import os
    spec = os.environ.get("EGC_FUZZ_EXTRA_SEEDS", "")
    out = []
    for part in filter(None, spec.split(",")):
        lo, _, hi = part.partition("-")
        out += [(s, False) for s in range(int(lo), int(hi or lo) + 1)]
    return out


def _stdvar_allowance(kind, layer, x, ei, H, B, names, flags, asl):
    """What float32 itself leaves in a std / var layer's output, element by element [N, F_out] (float64): the formula
    var = E[x^2] - E[x]^2 (layers.py:203-216) evaluated in float32 has a residue of a few ulp OF E[x^2] whatever the
    variance is -- sums of deg terms, two divisions, a square, a difference: (4 + sqrt(deg)) ulp is a generous envelope;
    measured on identical neighbours: 0 ... +-3 ulp, either sign with equal probability -- and std = sqrt(relu(var) + 1e-5)
    passes it on with slope 1 / (2 std) <= 158.  The allowance is that spread, through the combination's |weights|."""
    p = {k: v.detach().double().cpu() for k, v in layer.named_parameters()}
    x64 = torch.from_numpy(x).double()
    n = x64.size(0)
    with torch.no_grad():
        if kind == "opt":
            e2, _ = orc.egconv_edge_set(np.asarray(ei), n, list(names), asl)
            bases = x64 @ p["bases_weight"]
            w = x64 @ p["comb_weight.weight"].t() + p["comb_weight.bias"]
            if flags.get("sigmoid", False): w = torch.sigmoid(w)
            w = w.view(n, H, len(names), B).abs()                               # [n, h, a, b]
        else:
            e2 = np.asarray(ei)                                                  # std / var see the raw edges (layers.py:166-193)
            bases = torch.cat([x64 @ p[f"bases_weight.{b}"] for b in range(B)], dim=1)
            w = x64 @ p["comb_weights.weight"].t() + p["comb_weights.bias"]
            if flags.get("softmax", False): w = w.view(n, H, B * len(names)).softmax(dim=-1)
            elif flags.get("sigmoid", False): w = torch.sigmoid(w)
            elif flags.get("hardtanh", False): w = torch.nn.functional.hardtanh(w)
            w = w.view(n, H, B, len(names)).abs().permute(0, 1, 3, 2)            # -> [n, h, a, b]
        src, dst = torch.from_numpy(e2[0]).long(), torch.from_numpy(e2[1]).long()
        xj = bases[src]
        deg = torch.zeros(n, dtype=torch.float64).index_add(0, dst, torch.ones(dst.numel(), dtype=torch.float64))
        dn = deg.clamp(min=1).view(-1, 1)
        mean = torch.zeros_like(bases).index_add(0, dst, xj) / dn
        m2 = torch.zeros_like(bases).index_add(0, dst, xj * xj) / dn
        var = m2 - mean * mean
        delta = (4.0 + deg.sqrt()).view(-1, 1) * 2.0 ** -23 * (m2 + mean * mean)
        spread_std = torch.sqrt(torch.relu(var + delta) + 1e-5) - torch.sqrt(torch.relu(var - delta) + 1e-5)
        L = bases.size(1) // B
        allow = torch.zeros(n, H, L, dtype=torch.float64)
        for a, name in enumerate(names):
            if name in ("std", "var"):
                sp = (spread_std if name == "std" else 2.0 * delta).view(n, B, L)
                allow += torch.einsum("nhb,nbl->nhl", w[:, :, a, :], sp)
    return allow.reshape(n, H * L).numpy()


@pytest.mark.parametrize("seed,generic", [(11, False), (12, False), (13, False), (14, True), (15, True), (101, False), (109, False),
                                          (118, False), (202, False), (303, False), (337, False), (377, False), (413, False)]
                         + _extra_seeds())
def test_random_layers_and_graphs_match_the_oracles(seed, generic, monkeypatch):
    import egc_amd
    if generic:   # the generic forward kernels + separate arg pass, and the run-time forms of the backward kernels
        monkeypatch.setenv("EGC_FORCE_GENERIC", "1")
        monkeypatch.setenv("EGC_BWD_GENERIC", "1")
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(seed)
    n_cases = 120
    worst = 0.0
    worst_g = 0.0
    fails = []
    stdvar_stats = []

    LAY = ["add", "mean", "max", "min", "symadd", "var", "std"]
    OPT = ["sum", "mean", "max", "min", "symnorm", "var", "std"]
    for case in range(n_cases):
        kind = "lay" if rng.random() < 0.5 else "opt"
        H = int(rng.choice([1, 2, 4, 8, 16])); B = int(rng.choice([1, 2, 4, 8]))
        L = int(rng.choice([1, 3, 4, 7, 8, 12, 16, 21, 23, 31, 32, 34, 44, 64]))
        if H * L > 512 or B * L > 512: L = 8
        fout = H * L
        fin = fout if kind == "lay" else int(rng.choice([fout, 5, 17, 32, 100, 128, 136]))
        A = int(rng.integers(1, 5))
        names = list(rng.choice(LAY if kind == "lay" else OPT, size=A, replace=False))
        n = int(rng.choice([1, 2, 37, 200, 900, 3000]))
        e = int(rng.choice([0, 1, n, 4 * n, 12 * n]))
        ei = rng.integers(0, n, size=(2, e)).astype(np.int64)
        if e > 100 and rng.random() < 0.6:
            hub = int(rng.integers(0, n)); k = min(e, int(rng.choice([70, 300, 1500]))); ei[1, :k] = hub
        if e > 10 and rng.random() < 0.5:
            k = e // 10; ei[0, -k:] = ei[1, -k:]          # self loops
        if e > 10 and rng.random() < 0.3:
            ei[:, : e // 5] = ei[:, e // 5: 2 * (e // 5)][:, : e // 5]  # duplicates
        ei = ei[:, rng.permutation(e)] if e else ei
        flags = {}
        r = rng.random()
        if kind == "lay":
            if r < 0.2: flags["softmax"] = True
            elif r < 0.35: flags["sigmoid"] = True
            elif r < 0.5: flags["hardtanh"] = True
        elif r < 0.3: flags["sigmoid"] = True
        asl = bool(rng.random() < 0.75)
        torch.manual_seed(int(rng.integers(1 << 30)))
        try:
            if kind == "lay":
                layer = egc_amd.EfficientGraphConv(fin, fout, H, B, flags.get("softmax", False), aggrs=names, add_self_loops=asl,
                                                   sigmoid_weights=flags.get("sigmoid", False), hardtanh_weights=flags.get("hardtanh", False))
            else:
                layer = egc_amd.EGConv(fin, fout, aggrs=names, num_heads=H, num_bases=B, add_self_loops=asl, sigmoid=flags.get("sigmoid", False))
            with torch.no_grad(): layer.bias.normal_()
            x = rng.standard_normal((n, fin)).astype(np.float32)
            if rng.random() < 0.3 and n > 3: x[rng.integers(0, n, size=n // 2)] = x[0]   # ties
            sd = {k: v.numpy() for k, v in layer.state_dict().items()}
            meta = dict(kind=kind, fin=fin, fout=fout, H=H, B=B, aggrs=names, softmax=flags.get("softmax", False), sigmoid=flags.get("sigmoid", False),
                        hardtanh=flags.get("hardtanh", False), add_self_loops=asl, bias=True, sparse=False)
            g = dict(meta=meta, params=sd, x=x, edge_index=ei)
            ref = oracle_forward(g, orc)
            layer = layer.to(dev)
            xt = torch.from_numpy(x).to(dev); eit = torch.from_numpy(ei).to(dev)
            # adj_t (SparseTensor) inputs for ~30 % of the cases -- where the reference gives the adj_t path the same
            # semantics as the COO path the oracle restates: EfficientGraphConv always (var / std raise there), EGConv
            # unless its loops come from add_remaining_self_loops, which infers N from the largest index while
            # fill_diag loops every node (optimized_layers.py:158-175)
            use_sparse = (e > 0 and rng.random() < 0.3 and not (kind == "lay" and any(a in ("var", "std") for a in names))
                          and (kind == "lay" or "symnorm" in names or not asl or int(ei.max()) == n - 1))
            arg = egc_amd.SparseTensor(row=eit[1], col=eit[0], sparse_sizes=(n, n)) if use_sparse else eit
            with torch.no_grad():
                out = layer(xt, arg) if kind == "opt" else layer(x=xt, edge_index=arg)
            if out is not None:
                err = rel_err(out.cpu().numpy(), ref); worst = max(worst, err)
                stdvar = any(a in ('std', 'var') for a in names)
                ok = err <= 1e-5
                if not ok and not stdvar:
                    # the float32 numpy oracle adds a row's entries one after the other: on a 1,500-entry hub row that
                    # alone is 1.4e-5 (seed 303, case 106: oracle 1.44e-5 from float64, HIP 3.2e-7) -- beyond 1e-5 of the
                    # oracle the HIP result must hold 1e-5 against the same layer in FLOAT64
                    truth = _truth64(tref, kind, layer, x, ei, H, B, names, flags, asl).numpy()
                    e_hip, e_ref = rel_err(out.cpu().numpy(), truth), rel_err(ref, truth)
                    ok = e_hip <= 1e-5
                    err = (err, e_hip, e_ref)
                if not ok and stdvar:
                    # two correct fp32 evaluations of sqrt(relu(E[x^2] - E[x]^2) + 1e-5) may be 1e-3 apart on (nearly)
                    # constant neighbourhoods -- e.g. one node with a dozen self-loops -- so beyond 1e-5 the criterion is
                    # the one that does not depend on the evaluation order: against the same layer in FLOAT64 the HIP
                    # result holds 1e-5 wherever the reference's own float32 arithmetic (the restatement) does, and
                    # is otherwise of the order of the restatement's own error (STDVAR_K x; at the one element of a
                    # layer where mean^2 / var is largest both errors are single draws of the same amplified rounding
                    # noise).  Layers with std / var run the 24-bit-operand GEMM (egc_layer_gemm_flags) for this.
                    truth = _truth64(tref, kind, layer, x, ei, H, B, names, flags, asl).numpy()
                    e_hip, e_ref = rel_err(out.cpu().numpy(), truth), rel_err(ref, truth)
                    ok = e_hip <= max(1e-5, STDVAR_K * e_ref)
                    if not ok and STDVAR_ALLOWANCE:
                        # ... or, element by element, within what float32 leaves in E[x^2] - E[x]^2 (its residue has either
                        # sign with equal probability, relu() hides the negative draws: the restatement's own error on one
                        # layer is a sample of the same noise, not a bound for it)
                        allow = _stdvar_allowance(kind, layer, x, ei, H, B, names, flags, asl)
                        scale = max(1.0, float(np.abs(truth).max()))
                        ok = bool((np.abs(out.cpu().numpy().astype(np.float64) - truth) <= 1e-5 * scale + allow).all())
                    err = (err, e_hip, e_ref)
                    stdvar_stats.append((e_hip, e_ref, case))
                if not ok:
                    fails.append(("fwd", case, kind, H, B, L, fin, names, n, e, flags, asl, err))
                    import os
                    if os.environ.get("EGC_FUZZ_DUMP"):      # inputs of a failing case, for a look at it off the GPU
                        np.savez(os.path.join(os.environ["EGC_FUZZ_DUMP"], f"fuzz_{seed}_{case}.npz"), x=x, ei=ei, out=out.cpu().numpy(), ref=ref,
                                 meta=np.frombuffer(repr(dict(meta, aggrs=[str(a) for a in names])).encode(), dtype=np.uint8),
                                 **{f"p:{k}": v for k, v in sd.items()})
            # gradients for a subset (float64 torch reference); skip std/var/max/min kinks at exact ties
            if case % 4 == 0 and n <= 900 and not any(a in ("std", "var") for a in names):
                xg = torch.from_numpy(x).to(dev).requires_grad_(True)
                o = layer(xg, eit) if kind == "opt" else layer(x=xg, edge_index=eit)
                gout = torch.randn(o.shape, device=dev); o.backward(gout)
                p64 = {k: v.detach().double().cpu().requires_grad_(True) for k, v in layer.named_parameters()}
                x64 = torch.from_numpy(x).double().requires_grad_(True)
                if kind == "opt":
                    r64 = tref.egconv_forward(x64, ei, p64["bases_weight"], p64["comb_weight.weight"], p64["comb_weight.bias"], p64["bias"], H, B, names,
                                              add_self_loops=asl, sigmoid=flags.get("sigmoid", False))
                else:
                    r64 = tref.efficient_graph_conv_forward(x64, ei, [p64[f"bases_weight.{b}"] for b in range(B)], p64["comb_weights.weight"], p64["comb_weights.bias"],
                                                            p64["bias"], H, names, softmax=flags.get("softmax", False), hardtanh=flags.get("hardtanh", False),
                                                            sigmoid=flags.get("sigmoid", False), add_self_loops=asl)
                r64.backward(gout.double().cpu())
                def rel(a, b):
                    if b is None: b = torch.zeros_like(a, dtype=torch.float64, device='cpu')
                    if a is None: a = torch.zeros_like(b)
                    return float((a.detach().double().cpu() - b).abs().max() / max(1.0, float(b.abs().max())))
                ge = max([rel(xg.grad, x64.grad)] + [rel(v.grad, p64[k].grad) for k, v in layer.named_parameters()])
                if not ge <= 5e-4 and any(a in ("max", "min") for a in names):
                    # a NEAR-tie of max / min: float32 and float64 pick different neighbours and a whole gradient entry moves
                    # (seed 413, case 68: the float32 restatement is 9.8e-4 from float64 too, HIP 4.7e-7 from the float32
                    # restatement) -- then the float32 evaluation of the same formula is the yardstick
                    p32 = {k: v.detach().float().cpu().requires_grad_(True) for k, v in layer.named_parameters()}
                    x32 = torch.from_numpy(x).float().requires_grad_(True)
                    if kind == "opt":
                        r32 = tref.egconv_forward(x32, ei, p32["bases_weight"], p32["comb_weight.weight"], p32["comb_weight.bias"], p32["bias"], H, B,
                                                  names, add_self_loops=asl, sigmoid=flags.get("sigmoid", False))
                    else:
                        r32 = tref.efficient_graph_conv_forward(x32, ei, [p32[f"bases_weight.{b}"] for b in range(B)], p32["comb_weights.weight"],
                                                                p32["comb_weights.bias"], p32["bias"], H, names, softmax=flags.get("softmax", False),
                                                                hardtanh=flags.get("hardtanh", False), sigmoid=flags.get("sigmoid", False),
                                                                add_self_loops=asl)
                    r32.backward(gout.cpu())
                    g32 = max([rel(xg.grad, x32.grad.double())] + [rel(v.grad, p32[k].grad.double()) for k, v in layer.named_parameters()])
                    if g32 <= 2e-5:
                        ge = g32
                worst_g = max(worst_g, ge)
                if not ge <= 5e-4:
                    fails.append(("grad", case, kind, H, B, L, fin, names, n, e, flags, asl, ge))
                    import os
                    if os.environ.get("EGC_FUZZ_DUMP"):
                        np.savez(os.path.join(os.environ["EGC_FUZZ_DUMP"], f"fuzzgrad_{seed}_{case}.npz"), x=x, ei=ei, gout=gout.cpu().numpy(),
                                 gx=xg.grad.cpu().numpy(), meta=np.frombuffer(repr(dict(meta, aggrs=[str(a) for a in names])).encode(), dtype=np.uint8),
                                 **{f"p:{k}": v for k, v in sd.items()}, **{f"g:{k}": v.grad.cpu().numpy() for k, v in layer.named_parameters()})
        except Exception as ex:
            fails.append(("exc", case, kind, H, B, L, fin, names, n, e, flags, asl, repr(ex)[:200]))
    if stdvar_stats:
        print(f"[fuzz seed {seed}] std/var cases beyond 1e-5 of the fp32 restatement: {len(stdvar_stats)}; vs float64: "
              f"worst HIP {max(s[0] for s in stdvar_stats):.2e}, worst restatement {max(s[1] for s in stdvar_stats):.2e}, "
              f"worst HIP / restatement {max(s[0] / max(s[1], 1e-30) for s in stdvar_stats):.2f}")
    for f in fails[:5]:
        print("[fuzz fail]", seed, f)
    assert not fails, fails[:5]

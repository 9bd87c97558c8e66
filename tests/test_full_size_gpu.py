"""GPU parity at the FULL sizes of BASELINE.json's configs 3, 4 and 5 (config 2 at full size lives in
test_parity_gpu.py::test_config2_full_size_against_oracle): where the numpy oracle finishes in seconds the whole
output is compared with it; at ogbn-mag size sampled rows are recomputed in float64; on top, size-independent
properties (a sub-batch of a batched graph gives the same rows bit for bit; a cached graph is reused).

Tolerance (north_star): 1e-5, scale-relative (max |diff| / max(1, max |ref|))."""
import numpy as np
import pytest
import torch

from golden_util import rel_err
from oracle import egc_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _lay_oracle(conv, x, ei, H, aggrs):
    sd = {k: v.detach().cpu().numpy() for k, v in conv.state_dict().items()}
    B = conv.num_bases
    return orc.efficient_graph_conv_forward(
        x.numpy(), ei.numpy(), [sd[f"bases_weight.{b}"] for b in range(B)], sd["comb_weights.weight"],
        sd["comb_weights.bias"], sd["bias"], H, aggrs)


@pytest.mark.parametrize("workload,aggrs", [("molhiv", ["symadd", "max", "mean"]), ("cifar", ["symadd", "std", "max"])])
def test_batched_configs_full_size_against_oracle(workload, aggrs):
    """Config 3 (2048 molhiv-like molecules, ~52 k nodes) and config 4 (2048 CIFAR10-superpixel-like 8-NN graphs,
    ~241 k nodes / 1.93 M edges), EGC-M d=128 H=8 B=4 through EfficientGraphConv as the reference's nets build it
    (mol/pna_style_models.py:168-176, cifar/models.py:122-130), per-batch COO->CSR included."""
    import egc_amd
    from egc_amd.workloads import knn_superpixel_batch, molecule_batch
    dev = _dev()
    ei, n, batch = molecule_batch(2048, seed=0) if workload == "molhiv" else knn_superpixel_batch(2048, seed=0)
    torch.manual_seed(3)
    conv = egc_amd.EfficientGraphConv(128, 128, 8, 4, False, aggrs=aggrs)
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 128)
    ref = _lay_oracle(conv, x, ei, 8, aggrs)
    conv = conv.to(dev).eval()
    with torch.no_grad():
        out = conv(x=x.to(dev), edge_index=ei.to(dev))
        # property: graphs of a batch do not interact -- the first 100 graphs alone give the same rows, bit for bit
        n_sub = int((batch < 100).sum())
        keep = ei[1] < n_sub
        assert bool((ei[0][keep] < n_sub).all())
        sub = conv(x=x[:n_sub].to(dev), edge_index=ei[:, keep].to(dev))
    assert rel_err(out.cpu().numpy(), ref) <= TOL, rel_err(out.cpu().numpy(), ref)
    assert torch.equal(sub, out[:n_sub])


@pytest.mark.parametrize("aggrs", [["symnorm"], ["mean"]])
def test_config5_homogeneous_mag_size_sampled_rows(aggrs):
    """Config 5 as `main.py egc mag` runs it (mag/models.py:23-53, train_main_table.sh:53-54): N = 736,389,
    ~10.8 M symmetrised edges, EGConv(128 -> 352, H=8, B=4, cached=True) on an adj_t.  Sampled rows (hubs, the
    last row, random ones) recomputed in float64; the cached graph serves the second call."""
    import egc_amd
    from egc_amd.workloads import mag_like
    dev = _dev()
    ei, n = mag_like(seed=0)
    ei = ei.to(dev)
    torch.manual_seed(1)
    conv = egc_amd.EGConv(128, 352, aggrs=aggrs, num_heads=8, num_bases=4, cached=True).to(dev).eval()
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 128, device=dev)
    adj_t = egc_amd.SparseTensor(row=ei[1], col=ei[0], sparse_sizes=(n, n))
    H, B, L = 8, 4, 44
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        out = conv(x, adj_t)
        out2 = conv(2.0 * x, None)        # cached=True: the first graph is pinned (optimized_layers.py:138-139)
        rows = torch.cat([torch.tensor([0, 1, 2, n - 1]), torch.randint(0, n, (150,), generator=g)]).to(dev)
        nonself = torch.bincount(ei[1][ei[0] != ei[1]], minlength=n).double() + 1   # gcn_norm / fill_diag: one loop per node
        dis = nonself.pow(-0.5)
        sel = torch.isin(ei[1], rows) & (ei[0] != ei[1])
        s_sel, d_sel = ei[0][sel], ei[1][sel]
        Wb = conv.bases_weight.double()
        worst = 0.0
        for r in rows.tolist():
            nb = torch.cat([s_sel[d_sel == r], torch.tensor([r], device=dev)])
            bj = x[nb].double() @ Wb
            agg = (bj * (dis[nb] * dis[r])[:, None]).sum(0) if aggrs == ["symnorm"] else bj.mean(0)
            wt = (x[r].double() @ conv.comb_weight.weight.double().t() + conv.comb_weight.bias.double()).view(H, 1, B)
            ref = torch.einsum("hab,abl->hl", wt, agg.view(1, B, L)).reshape(-1) + conv.bias.double()
            worst = max(worst, float((out[r].double() - ref).abs().max() / ref.abs().max().clamp(min=1)))
    assert worst <= TOL, worst
    assert out2.shape == out.shape and bool(torch.isfinite(out2).all())


def _rel_adj(rel, nodes, dev):
    import egc_amd
    return {k: egc_amd.SparseTensor(row=ei[1].to(dev), col=ei[0].to(dev), sparse_sizes=(nodes[k[2]], nodes[k[0]]))
            for k, ei in rel.items()}


def test_config5_heterogeneous_scaled_against_oracle():
    """The relational path (REGConv, rmag/models.py:75-148) on a 2 % scale model of ogbn-mag's typed graph
    (same relation list, hub-heavy rectangular adjacencies): whole output against the numpy oracle."""
    import egc_amd
    from egc_amd.workloads import rmag_like
    dev = _dev()
    nodes, rel = rmag_like(seed=4, scale=0.02)
    torch.manual_seed(5)
    conv = egc_amd.REGConv(128, 64, 4, 4)
    x = {k: torch.randn(n, 128) for k, n in nodes.items()}
    sd = {k: v.detach().numpy() for k, v in conv.state_dict().items()}
    ref = orc.regconv_forward(
        {k: v.numpy() for k, v in x.items()}, {k: v.numpy() for k, v in rel.items()}, sd["bases_weight"],
        {f"{k[0]}_{k[1]}_{k[2]}": (sd[f"rel_combs.{k[0]}_{k[1]}_{k[2]}.weight"], sd[f"rel_combs.{k[0]}_{k[1]}_{k[2]}.bias"])
         for k in rel},
        {k: (sd[f"root_combs.{k}.weight"], sd[f"root_combs.{k}.bias"]) for k in nodes}, 4, 4)
    conv = conv.to(dev).eval()
    with torch.no_grad():
        out = conv({k: v.to(dev) for k, v in x.items()}, _rel_adj(rel, nodes, dev))
    for k in nodes:
        assert rel_err(out[k].cpu().numpy(), ref[k]) <= TOL, (k, rel_err(out[k].cpu().numpy(), ref[k]))


def test_config5_heterogeneous_full_size_sampled_rows():
    """ogbn-mag's full typed graph shape (1.94 M nodes, 21.1 M typed edges -> 42.1 M CSR entries over the seven
    relations): sampled `paper` and `field_of_study` rows (incl. the heaviest hubs) recomputed in float64."""
    import egc_amd
    from egc_amd.workloads import rmag_like
    dev = _dev()
    nodes, rel = rmag_like(seed=0)
    rel = {k: v.to(dev) for k, v in rel.items()}
    torch.manual_seed(6)
    H, B, L = 4, 4, 16
    conv = egc_amd.REGConv(128, H * L, H, B).to(dev).eval()
    x = {k: torch.randn(n, 128, device=dev) for k, n in nodes.items()}
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        out = conv(x, _rel_adj(rel, nodes, dev))
        Wb = conv.bases_weight.double()
        worst = 0.0
        for t in ("paper", "field_of_study"):
            rows = torch.cat([torch.tensor([0, 1, nodes[t] - 1]), torch.randint(0, nodes[t], (40,), generator=g)]).tolist()
            for r in rows:
                xr = x[t][r].double()
                lin = conv.root_combs[t]
                ref = torch.einsum("hb,bl->hl", (xr @ lin.weight.double().t() + lin.bias.double()).view(H, B),
                                   (xr @ Wb).view(B, L))
                for key, ei in rel.items():
                    if key[2] != t:
                        continue
                    nb = ei[0][ei[1] == r]
                    if nb.numel() == 0:
                        continue            # empty neighbourhood: mean and max are both 0
                    bj = x[key[0]][nb].double() @ Wb
                    agg = torch.stack([bj.mean(0), bj.max(0).values]).view(2 * B, L)
                    lin = conv.rel_combs[f"{key[0]}_{key[1]}_{key[2]}"]
                    w = (xr @ lin.weight.double().t() + lin.bias.double()).view(H, 2 * B)
                    ref = ref + w @ agg
                ref = ref.reshape(-1)
                worst = max(worst, float((out[t][r].double() - ref).abs().max() / ref.abs().max().clamp(min=1)))
    assert worst <= TOL, worst


@pytest.mark.parametrize("world", [8])
def test_config5_partitioned_at_mag_size_equals_single_gpu(world):
    """BASELINE config 5 as it is benchmarked at --gpus N: ONE ogbn-mag-sized graph, renumbered by
    partition.locality_partition, split into cost-balanced contiguous ranges, interior rows first; every rank's
    pieces run one after the other on this GPU (the all-to-all-v is simulated; the collective itself is covered by
    the gloo tests).  Un-permuted and concatenated, the per-rank outputs equal the single-GPU layer output:
    bit-exact is not required (a row's neighbour order changes with the renumbering), 1e-5 is."""
    import egc_amd
    from egc_amd import partition as P
    from egc_amd.functional import egc_aggregate_combine, egc_basis_transform
    from egc_amd.workloads import mag_like
    dev = _dev()
    ei, n = mag_like(seed=0)
    torch.manual_seed(5)
    conv = egc_amd.EGConv(352, 352, aggrs=["symnorm"], num_heads=8, num_bases=4).to(dev).eval()
    with torch.no_grad():
        conv.bias.normal_()
    x = torch.randn(n, 352, device=dev)
    ei_d = ei.to(dev)
    with torch.no_grad():
        ref = conv(x, ei_d)
        order, new_of_old, bounds = P.locality_partition(ei_d, n, world)
        q = P.partition_quality(new_of_old[ei_d], bounds)
        assert max(q["entries_per_rank"]) <= 1.2 * ei.size(1) / world
        ei2 = new_of_old[ei_d]
        x2 = x[order]                                   # features follow the renumbering
        wcat, bcat = conv._packed_weights()
        parts = P.build_local_simulation(ei2, n, world, interior_first=True, bounds=bounds)
        plans = [p for _, p in parts]
        graphs = [egc_amd.CSRGraph.from_partition(e, plan, global_max_index=n - 1, exchange_dis=False) for e, plan in parts]
        for key in ("dis_raw", "dis_looped"):
            P.simulate_exchange([getattr(g, key) for g in graphs], plans)
        for g in graphs:
            g.refresh_edge_dis()
        stage = [egc_basis_transform(g, conv._spec_coo, x2[pl.lo:pl.hi][pl.order], wcat, bcat) for g, pl in zip(graphs, plans)]
        outs = []
        for g, pl, (b, w) in zip(graphs, plans, stage):   # interior rows before the halo rows exist
            b[pl.n_local:] = float("nan")
            out = torch.empty((pl.n_local, 352), device=dev)
            egc_aggregate_combine(g, conv._spec_coo, b, w, conv.bias, rows=(0, pl.n_interior), out=out)
            outs.append(out)
        P.simulate_exchange([b for b, _ in stage], plans)
        got2 = torch.empty_like(ref)
        for g, pl, (b, w), out in zip(graphs, plans, stage, outs):
            egc_aggregate_combine(g, conv._spec_coo, b, w, conv.bias, rows=(pl.n_interior, pl.n_local), out=out)
            back = torch.empty_like(out)
            back[pl.order] = out
            got2[pl.lo:pl.hi] = back
        got = torch.empty_like(ref)
        got[order] = got2                               # back to the caller's vertex ids
    assert bool(torch.isfinite(got).all())
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) <= TOL
    # ... and against FLOAT64 recomputations of sampled rows (hubs, the last row, rows on both sides of every cut, random
    # ones) from the caller's un-renumbered graph: the partitioned path is held to the reference's arithmetic, not to
    # the single-GPU HIP result
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        cuts = torch.tensor([b for b in bounds[1:-1]] + [b - 1 for b in bounds[1:-1]])
        rows = torch.cat([torch.tensor([0, 1, 2, n - 1]), order.cpu()[cuts.clamp(0, n - 1)],
                          torch.randint(0, n, (120,), generator=g)]).to(dev)
        nonself = torch.bincount(ei_d[1][ei_d[0] != ei_d[1]], minlength=n).double() + 1
        dis = nonself.pow(-0.5)
        sel = torch.isin(ei_d[1], rows) & (ei_d[0] != ei_d[1])
        s_sel, d_sel = ei_d[0][sel], ei_d[1][sel]
        Wb = conv.bases_weight.double()
        H, B, L = 8, 4, 44
        worst = 0.0
        for r in rows.tolist():
            nb = torch.cat([s_sel[d_sel == r], torch.tensor([r], device=dev)])
            agg = ((x[nb].double() @ Wb) * (dis[nb] * dis[r])[:, None]).sum(0)
            wt = (x[r].double() @ conv.comb_weight.weight.double().t() + conv.comb_weight.bias.double()).view(H, 1, B)
            want = torch.einsum("hab,abl->hl", wt, agg.view(1, B, L)).reshape(-1) + conv.bias.double()
            worst = max(worst, float((got[r].double() - want).abs().max() / want.abs().max().clamp(min=1)))
    assert worst <= TOL, worst


def _regconv_partitioned(conv, x, nodes, rel, world, dev):
    """REGConv on `world` simulated ranks, one after the other on this GPU (typed partition, simulated all-to-all-v);
    returns {type: rows of all ranks concatenated} -- the ranks own consecutive ranges of every type."""
    from egc_amd import partition as P
    layout = P.typed_layout(nodes, rel, world, node_types=list(conv.node_types))
    parts = P.build_typed_local_simulation(rel, layout)
    ldb = conv._spec_root.ldb
    tables = [torch.full((max(p.n_table, 1), ldb), float("nan"), device=dev) for p in parts]
    graphs = [conv.partition_graphs(p, dev) for p in parts]
    xs = [{t: x[t][layout.owned(t, p.rank)[0]:layout.owned(t, p.rank)[1]].contiguous() for t in conv.node_types} for p in parts]
    # pass 1 on every rank: GEMMs + root terms (what runs before the halo rows have arrived); then the exchange; then
    # the relation terms.  forward_partitioned does all three per call, so it is called with an exchange hook that
    # (on the LAST rank's call of pass 1) has every table's owned rows ready
    outs = [None] * world
    for i, part in enumerate(parts):      # fill every rank's owned rows first
        conv.forward_partitioned(xs[i], part, {}, table=tables[i], exchange=lambda t: None)
    P.simulate_exchange(tables, [p.plan for p in parts])
    for i, part in enumerate(parts):
        halo = tables[i][part.plan.n_local:part.plan.n_local + part.plan.n_halo].clone()

        def put_back(t, halo=halo, part=part):
            t[part.plan.n_local:part.plan.n_local + part.plan.n_halo] = halo
        outs[i] = conv.forward_partitioned(xs[i], part, graphs[i], table=tables[i], exchange=put_back)
    return {t: torch.cat([o[t] for o in outs]) for t in conv.node_types}, layout, parts


@pytest.mark.parametrize("world", [2, 8])
def test_regconv_partitioned_equals_unpartitioned_and_oracle_scaled(world):
    """The vertex-partitioned REGConv (egc_amd.partition.TypedLayout: every node type cut into per-rank ranges, one
    table of basis rows per rank, ONE exchange for all types) on a 2 % scale model of the typed ogbn-mag graph: the
    ranks' rows concatenated == the numpy oracle of rmag/models.py:112-148 on the whole graph."""
    import egc_amd
    from egc_amd.workloads import rmag_like
    dev = _dev()
    nodes, rel = rmag_like(seed=4, scale=0.02)
    torch.manual_seed(5)
    conv = egc_amd.REGConv(128, 64, 4, 4)
    x = {k: torch.randn(n, 128) for k, n in nodes.items()}
    sd = {k: v.detach().numpy() for k, v in conv.state_dict().items()}
    ref = orc.regconv_forward(
        {k: v.numpy() for k, v in x.items()}, {k: v.numpy() for k, v in rel.items()}, sd["bases_weight"],
        {f"{k[0]}_{k[1]}_{k[2]}": (sd[f"rel_combs.{k[0]}_{k[1]}_{k[2]}.weight"], sd[f"rel_combs.{k[0]}_{k[1]}_{k[2]}.bias"])
         for k in rel},
        {k: (sd[f"root_combs.{k}.weight"], sd[f"root_combs.{k}.bias"]) for k in nodes}, 4, 4)
    conv = conv.to(dev).eval()
    xd = {k: v.to(dev) for k, v in x.items()}
    with torch.no_grad():
        got, layout, parts = _regconv_partitioned(conv, xd, nodes, {k: v.to(dev) for k, v in rel.items()}, world, dev)
        whole = conv(xd, _rel_adj(rel, nodes, dev))
    for k in nodes:
        assert got[k].shape == whole[k].shape and bool(torch.isfinite(got[k]).all())
        assert rel_err(got[k].cpu().numpy(), ref[k]) <= TOL, (k, rel_err(got[k].cpu().numpy(), ref[k]))
        assert rel_err(got[k].cpu().numpy(), whole[k].cpu().numpy()) <= TOL
    assert sum(p.plan.n_local for p in parts) == sum(nodes.values())


def test_regconv_partitioned_at_full_typed_mag_size_sampled_rows_float64():
    """BASELINE config 5's ~21 M-edge typed graph as bench.py --gpus N runs it (workloads.rmag_like at full size, 4
    simulated ranks): sampled `paper`, `author` and `field_of_study` rows -- the heaviest hubs, rows at the cuts, random
    ones -- recomputed in float64 from the un-partitioned typed graph."""
    import egc_amd
    from egc_amd.workloads import rmag_like
    dev = _dev()
    world = 4
    nodes, rel = rmag_like(seed=0)
    rel = {k: v.to(dev) for k, v in rel.items()}
    torch.manual_seed(6)
    H, B, L = 8, 4, 16
    conv = egc_amd.REGConv(128, H * L, H, B).to(dev).eval()
    x = {k: torch.randn(n, 128, device=dev) for k, n in nodes.items()}
    g = torch.Generator().manual_seed(7)
    with torch.no_grad():
        out, layout, parts = _regconv_partitioned(conv, x, nodes, rel, world, dev)
        Wb = conv.bases_weight.double()
        worst = 0.0
        for t in ("paper", "author", "field_of_study"):
            cuts = [b for b in layout.type_bounds[t][1:-1]] + [b - 1 for b in layout.type_bounds[t][1:-1]]
            rows = torch.cat([torch.tensor([0, 1, nodes[t] - 1] + [min(max(c, 0), nodes[t] - 1) for c in cuts]),
                              torch.randint(0, nodes[t], (30,), generator=g)]).tolist()
            for r in rows:
                xr = x[t][r].double()
                lin = conv.root_combs[t]
                ref = torch.einsum("hb,bl->hl", (xr @ lin.weight.double().t() + lin.bias.double()).view(H, B),
                                   (xr @ Wb).view(B, L))
                for key, ei in rel.items():
                    if key[2] != t:
                        continue
                    nb = ei[0][ei[1] == r]
                    if nb.numel() == 0:
                        continue            # empty neighbourhood: mean and max are both 0
                    bj = x[key[0]][nb].double() @ Wb
                    agg = torch.stack([bj.mean(0), bj.max(0).values]).view(2 * B, L)
                    lin = conv.rel_combs[f"{key[0]}_{key[1]}_{key[2]}"]
                    w = (xr @ lin.weight.double().t() + lin.bias.double()).view(H, 2 * B)
                    ref = ref + w @ agg
                ref = ref.reshape(-1)
                worst = max(worst, float((out[t][r].double() - ref).abs().max() / ref.abs().max().clamp(min=1)))
    assert worst <= TOL, worst
    # one exchange per layer whatever the number of types: a rank's table holds owned + halo rows of ALL types
    assert all(p.n_table == p.plan.n_local + p.plan.n_halo for p in parts)

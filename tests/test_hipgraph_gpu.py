"""A training step of a stack of EGC blocks recorded as one hipGraph (egc_amd.GraphedStep) reproduces the eager step:
nothing on the library's training path allocates outside torch's allocator, reads back to the host or depends on
state a replay would not restore (workspaces are zero on exit; the per-batch graph build is stream-ordered)."""
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu


def _blocks(dev, kind, n_blocks=2, hidden=64):
    import egc_amd
    from egc_amd.fusion import FusedEGCBlock
    torch.manual_seed(0)
    def conv():
        if kind == "opt":
            return egc_amd.EGConv(hidden, hidden, aggrs=["sum", "max", "symnorm"], num_heads=4, num_bases=4)
        return egc_amd.EfficientGraphConv(hidden, hidden, 4, 4, False, aggrs=["add", "std", "max"])
    return nn.ModuleList([FusedEGCBlock(conv(), nn.BatchNorm1d(hidden)) for _ in range(n_blocks)]).to(dev).train()


def _grads(params):
    return [p.grad.detach().clone() for p in params]


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("kind", ["opt", "lay"])
def test_recorded_training_step_equals_the_eager_step(kind):
    import egc_amd
    from egc_amd import workloads as wl
    dev = torch.device("cuda:0")
    _, ei, n, _ = wl.zinc_like_batch(64, seed=3)
    ei = ei.to(dev)
    blocks = _blocks(dev, kind)
    params = list(blocks.parameters())
    graph = egc_amd.CSRGraph.from_edge_index(ei, n).trim_launches()
    x = torch.randn(n, 64, device=dev)
    gout = torch.randn(n, 64, device=dev)
    stats0 = [b.bn.running_mean.clone() for b in blocks]

    def step():
        h = x
        for b in blocks:
            h = b(h, graph)
        h.backward(gout)

    def eager():
        for p in params:
            p.grad = None
        step()
        return _grads(params)

    graphed = egc_amd.GraphedStep(step, params=params)
    for trial in range(3):
        x.copy_(torch.randn(n, 64, device=dev))          # new contents of the static buffers
        gout.copy_(torch.randn(n, 64, device=dev))
        graphed()
        got = _grads(params)
        held = [p.grad for p in params]
        ref = eager()
        for p, h in zip(params, held):                    # give the recording its gradient buffers back
            p.grad = h
        for a, b in zip(got, ref):
            assert torch.equal(a, b), (kind, trial, _rel(a, b))
    # the recorded BatchNorm updated its running statistics at every replay, as the eager module does
    assert all(not torch.equal(b.bn.running_mean, s) for b, s in zip(blocks, stats0))


@pytest.mark.parametrize("helper", [True, False])
def test_per_batch_graph_build_inside_the_recording(helper):
    """COO in: the CSR build (egc_graph_build) is part of the recorded step, so a replay rebuilds the graph from whatever
    the static edge_index buffer holds -- batches of the same padded size go through one recording.  The graph cache
    must not hand a recording a graph built before it (stale at the first replay): inside GraphedStep the layers share
    one recorded build, a plain torch.cuda.graph recording gets one build per layer call."""
    import egc_amd
    from egc_amd import workloads as wl
    dev = torch.device("cuda:0")
    batches = []
    for seed in range(3):
        _, ei, n, _ = wl.zinc_like_batch(48, seed=seed)
        batches.append((ei, n))
    n_pad = max(n for _, n in batches) + 1
    e_pad = max(ei.size(1) for ei, _ in batches)

    def padded(ei):     # pad with self-loops on a spare last node (isolated from every real graph)
        extra = e_pad - ei.size(1)
        return torch.cat([ei, torch.full((2, extra), n_pad - 1, dtype=ei.dtype)], dim=1).to(dev)

    blocks = _blocks(dev, "opt")
    params = list(blocks.parameters())
    edge_index = padded(batches[0][0]).clone()
    x = torch.randn(n_pad, 64, device=dev)
    gout = torch.randn(n_pad, 64, device=dev)

    def step():
        h = x
        for b in blocks:
            h = b(h, edge_index)
        h.backward(gout)

    if helper:
        graphed = egc_amd.GraphedStep(step, params=params)
    else:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                for p in params:
                    p.grad = None
                step()
        torch.cuda.current_stream().wait_stream(side)
        for p in params:
            p.grad = None
        cg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(cg):
            step()
        graphed = cg.replay
    for ei, _ in batches[1:] + batches[:1]:
        edge_index.copy_(padded(ei))
        x.copy_(torch.randn(n_pad, 64, device=dev))
        graphed()
        got = _grads(params)
        held = [p.grad for p in params]
        for p in params:
            p.grad = None
        step()
        ref = _grads(params)
        for p, h in zip(params, held):
            p.grad = h
        for a, b in zip(got, ref):
            assert _rel(a, b) <= 1e-6


def test_whole_iteration_with_the_optimizer_in_the_recording():
    """forward + loss + backward + Adam (capturable) + zero_grad as ONE recording: after k replays the parameters equal
    those of k eager iterations from the same start, bit for bit.  (GraphedStep's warm-up runs are real calls of the
    step: here the start state is restored after the recording to compare like with like.)"""
    import copy
    import egc_amd
    from egc_amd import workloads as wl
    dev = torch.device("cuda:0")
    _, ei, n, _ = wl.zinc_like_batch(32, seed=2)
    ei = ei.to(dev)
    x = torch.randn(n, 64, device=dev)
    tgt = torch.randn(n, 64, device=dev)

    def run(recorded, k=6):
        blocks = _blocks(dev, "opt")
        params = list(blocks.parameters())
        opt = torch.optim.Adam(params, lr=1e-2, capturable=True, foreach=True)
        start = copy.deepcopy(blocks.state_dict())

        def step():
            h = x
            for b in blocks:
                h = b(h, ei)
            ((h - tgt) ** 2).mean().backward()
            opt.step()
            opt.zero_grad(set_to_none=False)
        if not recorded:
            for p in params:
                p.grad = torch.zeros_like(p)
            for _ in range(k):
                step()
        else:
            g = egc_amd.GraphedStep(step, params=params, warmup=2)
            blocks.load_state_dict(start)
            for st in opt.state.values():
                for v in st.values():
                    if torch.is_tensor(v):
                        v.zero_()
            for p in params:
                p.grad.zero_()
            for _ in range(k):
                g()
        torch.cuda.synchronize()
        return [p.detach().clone() for p in params] + [b.bn.running_var.clone() for b in blocks]

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b)


def test_recorded_step_on_a_graph_with_hub_rows():
    """Hub rows go through the long-row chunk role: arrival counters in the workspace (zero on entry, zero on exit),
    partial records, float atomics in the backward's hub rows.  Replays of one recording must keep giving the eager
    step's result (to rounding: the hub rows' backward sums are order-dependent)."""
    import egc_amd
    from egc_amd.workloads import heavy_tailed_graph
    dev = torch.device("cuda:0")
    n = 20000
    ei = heavy_tailed_graph(n, 160000, seed=4).to(dev)
    deg = torch.bincount(ei[1], minlength=n)
    assert int(deg.max()) > 2000                     # rows well above the long-row threshold
    blocks = _blocks(dev, "opt")
    params = list(blocks.parameters())
    graph = egc_amd.CSRGraph.from_edge_index(ei, n)          # capacity-sized launches: nothing read back
    x = torch.randn(n, 64, device=dev)
    gout = torch.randn(n, 64, device=dev)

    def step():
        h = x
        for b in blocks:
            h = b(h, graph)
        h.backward(gout)

    graphed = egc_amd.GraphedStep(step, params=params)
    for trial in range(4):
        x.copy_(torch.randn(n, 64, device=dev))
        graphed()
        got = _grads(params)
        held = [p.grad for p in params]
        for p in params:
            p.grad = None
        step()
        ref = _grads(params)
        for p, h in zip(params, held):
            p.grad = h
        for a, b in zip(got, ref):      # (a conv bias in front of a BatchNorm has a zero gradient: absolute floor)
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max())), trial


def test_one_recording_serves_batches_of_different_size_with_batch_norm():
    """Batches padded to one static (N, E) -- isolated nodes behind the real ones, self-loops on the last of them -- go
    through ONE recording; with the number of real rows in a device scalar (n_valid) BatchNorm's batch statistics count
    the real rows only, so gradients and running statistics equal those of the eager step on the UNPADDED batch."""
    import egc_amd
    from egc_amd import workloads as wl
    dev = torch.device("cuda:0")
    batches = []
    for k, seed in ((40, 1), (56, 2), (33, 3)):
        _, ei, n, _ = wl.zinc_like_batch(k, seed=seed)
        batches.append((ei.to(dev), n, torch.randn(n, 64, device=dev), torch.randn(n, 64, device=dev)))
    n_pad = max(b[1] for b in batches) + 8
    e_pad = max(b[0].size(1) for b in batches) + 16
    blocks = _blocks(dev, "opt")
    params = list(blocks.parameters())
    # static buffers of the recording
    x = torch.zeros(n_pad, 64, device=dev)
    gout = torch.zeros(n_pad, 64, device=dev)          # zero on the padding rows: they do not reach the loss
    edge_index = torch.full((2, e_pad), n_pad - 1, dtype=torch.int64, device=dev)
    n_valid = torch.zeros((), dtype=torch.int64, device=dev)

    def load(ei, n, xb, gb):
        x.zero_(); gout.zero_()
        x[:n] = xb; gout[:n] = gb
        edge_index.fill_(n_pad - 1)
        edge_index[:, :ei.size(1)] = ei
        n_valid.fill_(n)

    def step():
        h = x
        for b in blocks:
            h = b(h, edge_index, n_valid=n_valid)
        h.backward(gout)

    load(*batches[0])
    graphed = egc_amd.GraphedStep(step, params=params)
    import copy
    for ei, n, xb, gb in batches[1:] + batches[:1]:
        ref_blocks = copy.deepcopy(blocks)
        for p in ref_blocks.parameters():
            p.grad = None
        load(ei, n, xb, gb)
        graphed()
        got = _grads(params) + [b.bn.running_mean.clone() for b in blocks] + [b.bn.running_var.clone() for b in blocks]
        h = xb
        for b in ref_blocks:                            # the eager step on the unpadded batch
            h = b(h, ei)
        h.backward(gb)
        ref = [p.grad for p in ref_blocks.parameters()] + [b.bn.running_mean for b in ref_blocks] + \
              [b.bn.running_var for b in ref_blocks]
        for a, b in zip(got, ref):
            assert float((a - b).abs().max()) <= 2e-5 * max(1.0, float(b.abs().max()))


def test_small_net_trained_on_padded_batches_through_one_recording():
    """The reference's small-batch loop end to end (zinc/main.py shape: embedding -> blocks -> mean pool -> head -> loss
    -> optimizer step): random-size batches copied into static padded buffers, ONE recording replayed per batch,
    against the eager loop on the unpadded batches -- same parameters after six steps."""
    import copy
    import egc_amd
    from egc_amd import workloads as wl
    dev = torch.device("cuda:0")
    hidden, n_types = 64, 28
    torch.manual_seed(0)
    emb = nn.Embedding(n_types, hidden).to(dev)
    blocks = _blocks(dev, "opt", n_blocks=2, hidden=hidden)
    head = nn.Linear(hidden, 1).to(dev)
    net = nn.ModuleList([emb, blocks, head])
    ref_net = copy.deepcopy(net)
    data = []
    for k, seed in ((24, 1), (40, 2), (31, 3), (40, 4), (17, 5), (36, 6)):
        atom, ei, n, batch = wl.zinc_like_batch(k, seed=seed)
        data.append((atom.to(dev), ei.to(dev), n, batch.to(dev), k, torch.randn(k, 1, device=dev)))
    n_pad = max(d[2] for d in data) + 4
    e_pad = max(d[1].size(1) for d in data) + 8
    g_pad = max(d[4] for d in data) + 1                   # + a spare graph that owns the padding rows

    def forward(net_, atom, ei, batch, n_graphs, counts, n_valid=None):
        emb_, blocks_, head_ = net_
        h = emb_(atom)
        for b in blocks_:
            h = b(h, ei, n_valid=n_valid)
        if n_valid is not None:      # the recorded side: the library's readout (size given: nothing read back)
            pooled = egc_amd.global_mean_pool(h, batch, size=n_graphs)
        else:
            pooled = torch.zeros(n_graphs, hidden, device=dev).index_add_(0, batch, h) / counts
        return head_(pooled)

    # static buffers
    s_atom = torch.zeros(n_pad, dtype=torch.int64, device=dev)
    s_ei = torch.full((2, e_pad), n_pad - 1, dtype=torch.int64, device=dev)
    s_batch = torch.full((n_pad,), g_pad - 1, dtype=torch.int64, device=dev)
    s_counts = torch.ones(g_pad, 1, device=dev)
    s_target = torch.zeros(g_pad, 1, device=dev)
    s_gmask = torch.zeros(g_pad, 1, device=dev)
    s_inv_g = torch.ones((), device=dev)
    s_nvalid = torch.zeros((), dtype=torch.int64, device=dev)
    params = list(net.parameters())
    opt = torch.optim.SGD(params, lr=0.05, foreach=True)
    ref_opt = torch.optim.SGD(list(ref_net.parameters()), lr=0.05, foreach=True)

    def load(atom, ei, n, batch, k, target):
        s_atom.zero_(); s_atom[:n] = atom
        s_ei.fill_(n_pad - 1); s_ei[:, :ei.size(1)] = ei
        s_batch.fill_(g_pad - 1); s_batch[:n] = batch
        s_counts.fill_(1.0); s_counts[:k, 0] = torch.bincount(batch, minlength=k).float()
        s_target.zero_(); s_target[:k] = target
        s_gmask.zero_(); s_gmask[:k] = 1.0
        s_inv_g.fill_(1.0 / k)
        s_nvalid.fill_(n)

    def step():
        out = forward(net, s_atom, s_ei, s_batch, g_pad, s_counts, s_nvalid)
        (((out - s_target) ** 2 * s_gmask).sum() * s_inv_g).backward()     # mean over the real graphs
        opt.step()
        opt.zero_grad(set_to_none=False)

    load(*data[0])
    start = copy.deepcopy(net.state_dict())
    graphed = egc_amd.GraphedStep(step, params=params, warmup=2)
    net.load_state_dict(start)                             # (the warm-up runs were real steps)
    for p in params:
        p.grad.zero_()
    for atom, ei, n, batch, k, target in data:
        load(atom, ei, n, batch, k, target)
        graphed()
        ref_opt.zero_grad(set_to_none=True)
        counts = torch.bincount(batch, minlength=k).float()[:, None]
        out = forward(ref_net, atom, ei, batch, k, counts)
        ((out - target) ** 2).mean().backward()
        ref_opt.step()
    torch.cuda.synchronize()
    for (name, a), (_, b) in zip(net.state_dict().items(), ref_net.state_dict().items()):
        if a.dtype.is_floating_point:
            assert float((a - b).abs().max()) <= 1e-4 * max(1.0, float(b.abs().max())), name
        else:
            assert torch.equal(a, b), name

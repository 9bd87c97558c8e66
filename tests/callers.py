"""Test-only stand-ins for the reference's CALLERS of the layer: a ZINC-style graph
regression net shaped like experiments/zinc/models.py:17-74 -- Embedding(28, d) -> L x [EGC ->
BatchNorm1d -> ReLU (-> + identity)] -> mean pool over graphs -> MLP [d, d/2, d/4, 1] (the MLP follows
experiments/utils.py:30-40: Linear, BatchNorm1d, act, Dropout per hidden layer).  The layer is injected,
so the same net runs on the gfx950 drop-in and on the CPU restatement."""
import torch
import torch.nn as nn


def mean_pool(x, batch, n_graphs):
    out = torch.zeros(n_graphs, x.size(1), dtype=x.dtype, device=x.device).index_add_(0, batch, x)
    cnt = torch.zeros(n_graphs, dtype=x.dtype, device=x.device).index_add_(0, batch, torch.ones_like(batch, dtype=x.dtype))
    return out / cnt.clamp(min=1).unsqueeze(1)


def head_mlp(sizes):
    mods = []
    for a, b in zip(sizes[:-2], sizes[1:-1]):
        mods += [nn.Linear(a, b), nn.BatchNorm1d(b), nn.ReLU(), nn.Dropout(0.0)]
    mods.append(nn.Linear(sizes[-2], sizes[-1]))
    return nn.Sequential(*mods)


class ZincStyleNet(nn.Module):
    def __init__(self, hidden, make_layer, n_layers=4, residual=True, n_atom_types=28):
        super().__init__()
        self.embedding = nn.Embedding(n_atom_types, hidden)
        self.convs = nn.ModuleList([make_layer(hidden) for _ in range(n_layers)])
        self.norms = nn.ModuleList([nn.BatchNorm1d(hidden) for _ in range(n_layers)])
        self.readout = head_mlp([hidden, hidden // 2, hidden // 4, 1])
        self.residual = residual

    def forward(self, atom_type, edge_index, batch, n_graphs):
        x = self.embedding(atom_type)
        for conv, bn in zip(self.convs, self.norms):
            h = torch.relu(bn(conv(x=x, edge_index=edge_index)))
            x = x + h if self.residual else h
        return self.readout(mean_pool(x, batch, n_graphs))


# ---------------------------------------------------------------------------------------------------------------
# Counterparts of the reference's nets with the reference's OWN module names, so that a state dict captured from the
# reference's classes (tests/golden/net_*.npz, make_golden_nets.py) loads with strict=True: what a user does who swaps
# the layer class under released checkpoints.  `conv_fn(conv, x, edge_index)` decides what runs the
# layer (the module itself on the GPU; the CPU restatement wearing the module's parameters in the CPU tests);
# `fuse(conv, bn)` may return a block that runs conv -> bn -> relu (-> + x) in one go (egc_amd.FusedEGCBlock).
# ---------------------------------------------------------------------------------------------------------------
def _default_conv(conv, x, edge_index):
    return conv(x=x, edge_index=edge_index) if hasattr(conv, "aggs") else conv(x, edge_index)


class ZincNetLike(nn.Module):
    """zinc/models.py:17-74: embedding -> L x [conv, BatchNorm1d, ReLU (+ identity)] -> mean pool -> mlp."""

    def __init__(self, hidden, n_layers, make_layer, residual=True):
        super().__init__()
        self.embedding = nn.Embedding(28, hidden)
        self.graph_layers = nn.ModuleList([nn.ModuleList([make_layer(hidden), nn.BatchNorm1d(hidden), nn.ReLU()])
                                           for _ in range(n_layers)])
        self.mlp = head_mlp([hidden, hidden // 2, hidden // 4, 1])
        self.residual = residual

    def forward(self, atom, edge_index, batch, n_graphs, conv_fn=_default_conv, fuse=None, pool=mean_pool):
        x = self.embedding(atom.view(-1))
        for conv, bn, act in self.graph_layers:
            if fuse is not None:
                x = fuse(conv, bn, self.residual)(x, edge_index)
            else:
                h = act(bn(conv_fn(conv, x, edge_index)))
                x = x + h if self.residual else h
        return self.mlp(pool(x, batch, n_graphs))


class ArxivNetLike(nn.Module):
    """arxiv/norm_models.py:13-43 with dropout 0: Linear -> L x [conv, bn, relu, + identity] -> Linear -> log_softmax."""

    def __init__(self, hidden, n_layers, make_layer, residual=True, n_features=128, n_classes=40):
        super().__init__()
        self.embed = nn.Sequential(nn.Linear(n_features, hidden))        # utils.mlp([128, hidden]) is one Linear
        self.convs = nn.ModuleList([make_layer(hidden) for _ in range(n_layers)])
        self.bns = nn.ModuleList([nn.BatchNorm1d(hidden) for _ in range(n_layers)])
        self.out = nn.Linear(hidden, n_classes)
        self.residual = residual

    def forward(self, x, edge_index, conv_fn=_default_conv, fuse=None):
        x = self.embed(x)
        for conv, bn in zip(self.convs, self.bns):
            if fuse is not None:
                x = fuse(conv, bn, self.residual)(x, edge_index)
            else:
                h = torch.relu(bn(conv_fn(conv, x, edge_index)))
                x = x + h if self.residual else h
        return self.out(x).log_softmax(dim=-1)


class MagNetLike(nn.Module):
    """mag/models.py:16-69 with dropout 0: EGConv(128 -> h) -> relu -> ... -> EGConv(h -> 352)[:, :349] -> log_softmax."""

    def __init__(self, hidden, n_layers, make_layer, in_features=128, out_rounded=352, out_true=349):
        super().__init__()
        dims = [in_features] + [hidden] * (n_layers - 1) + [out_rounded]
        self.convs = nn.ModuleList([make_layer(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.out_true = out_true

    def forward(self, x, adj_t, conv_fn=_default_conv):
        for conv in self.convs[:-1]:
            x = torch.relu(conv_fn(conv, x, adj_t))
        return conv_fn(self.convs[-1], x, adj_t)[:, :self.out_true].log_softmax(dim=-1)


# ---------------------------------------------------------------------------------------------------------------------
# The training loop of the batched nets (zinc/configs.py): this repository's counterpart of train / evaluate and of the
# optimizer / scheduler wiring around them, pinned by tests/golden/train_*.npz (the reference's own loop run under shims).
# ---------------------------------------------------------------------------------------------------------------------
def zinc_loss(out, y):
    """zinc/configs.py:48-50: L1 between the net's [n_graphs, 1] output and y viewed alike."""
    return torch.nn.functional.l1_loss(out, y.view_as(out))


def train_epoch(forward, modules, optimizer, batches):
    """zinc/configs.py:53-72: one pass over the training batches -- zero_grad, forward, L1 loss, backward, step; returns
    the mean of the per-batch losses.  ``forward(batch)`` runs the net on one batch (a dict of tensors)."""
    for m in modules:
        m.train()
    total, count = 0.0, 0
    for batch in batches:
        optimizer.zero_grad()
        loss = zinc_loss(forward(batch), batch["y"])
        loss.backward()
        optimizer.step()
        total += loss.item()
        count += 1
    return total / count


@torch.no_grad()
def evaluate(forward, modules, batches):
    """zinc/configs.py:75-90: eval mode, mean of the per-batch L1 losses."""
    for m in modules:
        m.eval()
    total, count = 0.0, 0
    for batch in batches:
        total += zinc_loss(forward(batch), batch["y"]).item()
        count += 1
    return total / count


def fit_zinc(forward, modules, params, data, lr, wd, iterations):
    """The per-iteration schedule of zinc/configs.py:128-154: Adam(lr, weight_decay=wd); ReduceLROnPlateau(mode 'min',
    factor 0.5, patience 10, min_lr 1e-5) stepped with the validation loss after every training pass; test loss at the
    end.  Returns (train losses, validation losses, learning rates, test loss)."""
    opt = torch.optim.Adam(params, lr=lr, weight_decay=wd)
    sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, "min", factor=0.5, patience=10, min_lr=1e-5)
    tr, va, lrs = [], [], []
    for _ in range(iterations):
        tr.append(train_epoch(forward, modules, opt, data["train"]))
        v = evaluate(forward, modules, data["val"])
        sched.step(v)
        va.append(v)
        lrs.append(opt.param_groups[0]["lr"])
    return tr, va, lrs, evaluate(forward, modules, data["test"])

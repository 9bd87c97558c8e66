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

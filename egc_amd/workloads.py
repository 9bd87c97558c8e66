"""Synthetic workloads shaped like the BASELINE.json configs.  Datasets are not
available offline, so every benchmark / full-size parity input is generated here, seeded, on the CPU."""
from __future__ import annotations

import math

import torch

ARXIV_NODES = 169_343
ARXIV_DIRECTED_EDGES = 1_166_243


def heavy_tailed_graph(n_nodes: int, n_directed: int, seed: int = 0, communities: int = 0, p_in: float = 0.0) -> torch.Tensor:
    """Directed pairs with a heavy-tailed destination distribution (dst = floor(N u^3)), uniform
    sources; then symmetrised, coalesced, self-pairs removed -- the preprocessing the reference
    applies to ogbn-arxiv (``to_undirected``, experiments/arxiv/configs.py:100).  Returns int64 [2, E]
    sorted by (source, destination).

    ``communities`` > 0 plants community structure the way citation graphs have it (papers cite inside their
    field): vertex v belongs to community v mod K -- interleaved, so that a contiguous split of the ids sees
    none of it -- and a fraction ``p_in`` of the pairs draws its source from the destination's community."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n_directed, generator=g, dtype=torch.float64)
    dst = (n_nodes * u ** 3).long().clamp_(max=n_nodes - 1)
    src = torch.randint(0, n_nodes, (n_directed,), generator=g)
    if communities > 0 and p_in > 0:
        inside = torch.rand(n_directed, generator=g) < p_in
        k = communities
        per = (n_nodes - 1 - dst % k) // k + 1                      # members of dst's community
        pick = (torch.rand(n_directed, generator=g, dtype=torch.float64) * per).long()
        src = torch.where(inside, dst % k + k * pick, src)
    a = torch.cat([src, dst])
    b = torch.cat([dst, src])
    keep = a != b
    key = torch.unique(a[keep] * n_nodes + b[keep])
    return torch.stack([key // n_nodes, key % n_nodes])


def arxiv_like(seed: int = 0) -> tuple[torch.Tensor, int]:
    """BASELINE config 2 graph: N = 169,343, ~2.3 M symmetrised edges."""
    return heavy_tailed_graph(ARXIV_NODES, ARXIV_DIRECTED_EDGES, seed), ARXIV_NODES


def partitioned_arxiv_like(rank: int, world: int, seed: int = 0, cross_frac: float = 0.05):
    """Weak-scaling workload for the vertex-partitioned run: a graph of ``world`` arxiv-sized vertex ranges.
    Rank r owns ids [r*N0, (r+1)*N0); inside its range it has its own arxiv-like graph (seed + r); on top,
    ``cross_frac`` x (directed edges of one range) x world/2 undirected pairs connect different ranges
    (uniform endpoint on one side, heavy-tailed endpoint on the other -- hubs attract remote neighbours too).
    Every rank draws the SAME cross list from a shared seed and keeps the directions whose destination it
    owns.  Returns (edges with GLOBAL ids whose destination rank `rank` owns, n_global)."""
    n0 = ARXIV_NODES
    n_global = n0 * world
    local = heavy_tailed_graph(n0, ARXIV_DIRECTED_EDGES, seed + rank) + rank * n0
    if world == 1 or cross_frac <= 0:
        return local, n_global
    g = torch.Generator().manual_seed(seed + 7919)
    n_pairs = int(cross_frac * ARXIV_DIRECTED_EDGES * world / 2)
    ra = torch.randint(0, world, (n_pairs,), generator=g)
    rb = (ra + torch.randint(1, world, (n_pairs,), generator=g)) % world
    a = ra * n0 + torch.randint(0, n0, (n_pairs,), generator=g)
    u = torch.rand(n_pairs, generator=g, dtype=torch.float64)
    b = rb * n0 + (n0 * u ** 3).long().clamp_(max=n0 - 1)
    src = torch.cat([a, b])
    dst = torch.cat([b, a])
    mine = (dst >= rank * n0) & (dst < (rank + 1) * n0)
    cross = torch.stack([src[mine], dst[mine]])
    return torch.cat([local, cross], dim=1), n_global


def molecule_batch(n_graphs: int = 2048, mean_nodes: float = 25.5, std_nodes: float = 12.0, min_nodes: int = 2,
                   max_nodes: int = 222, seed: int = 0) -> tuple[torch.Tensor, int, torch.Tensor]:
    """BASELINE config 3 (molhiv-like): disjoint union of small tree-plus-a-few-rings graphs, about
    2.15 directed edges per node.  Returns (edge_index, n_nodes, batch)."""
    g = torch.Generator().manual_seed(seed)
    sizes = (torch.randn(n_graphs, generator=g) * std_nodes + mean_nodes).round().long().clamp_(min_nodes, max_nodes)
    offs = torch.cumsum(sizes, 0) - sizes
    srcs, dsts = [], []
    for k in range(n_graphs):
        n, o = int(sizes[k]), int(offs[k])
        if n < 2:
            continue
        child = torch.arange(1, n)
        parent = (torch.rand(n - 1, generator=g) * child).long()  # random recursive tree
        extra = max(0, int(round(0.075 * n)))
        ea = torch.randint(0, n, (extra,), generator=g)
        eb = torch.randint(0, n, (extra,), generator=g)
        ok = ea != eb
        s = torch.cat([child, parent, ea[ok], eb[ok]]) + o
        d = torch.cat([parent, child, eb[ok], ea[ok]]) + o
        srcs.append(s)
        dsts.append(d)
    ei = torch.stack([torch.cat(srcs), torch.cat(dsts)])
    batch = torch.repeat_interleave(torch.arange(n_graphs), sizes)
    return ei, int(sizes.sum()), batch


def zinc_like_batch(n_graphs: int = 128, seed: int = 0):
    """BASELINE config 1 (ZINC-12k-like): molecule graphs of ~23 nodes, integer atom types in [0, 28).
    Returns (atom_type [N] int64, edge_index, n_nodes, batch)."""
    ei, n, batch = molecule_batch(n_graphs, mean_nodes=23.2, std_nodes=4.5, min_nodes=9, max_nodes=37, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    return torch.randint(0, 28, (n,), generator=g), ei, n, batch


def code_like_batch(n_graphs: int = 128, mean_nodes: float = 125.0, min_nodes: int = 20, max_nodes: int = 250, seed: int = 0):
    """ogbg-code2-shaped batch (code/models.py:103-115, run_pretrained.sh:47-48; batch size 128: code/configs.py:163): the AST of a
    Python function with its nodes in depth-first order -- a tree -- plus, as the reference's ``augment_edge`` adds them
    (code/utils.py:74-135), the inverse AST edges, next-token edges chaining the attributed nodes (about half the nodes) in
    depth-first order, and their inverses: about 3 directed edges per node, a graph's edges contiguous and in the order
    [AST | inverse AST | next-token | inverse next-token] -- NOT sorted by destination.  Sizes: log-normal around ``mean_nodes``
    (ogbg-code2: 125 nodes on average), clipped to [min_nodes, max_nodes].  Returns (edge_index, n_nodes, batch)."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.exp(torch.randn(n_graphs, generator=g) * 0.45 + math.log(mean_nodes) - 0.1).round().long().clamp_(min_nodes, max_nodes)
    offs = torch.cumsum(sizes, 0) - sizes
    parts = []
    for k in range(n_graphs):
        n, o = int(sizes[k]), int(offs[k])
        child = torch.arange(1, n)
        # depth-first numbering: a node's parent is a recent node (the current path of the traversal)
        back = (torch.rand(n - 1, generator=g) ** 2 * torch.clamp(child, max=12).float()).long()
        parent = child - 1 - torch.minimum(back, child - 1)
        attributed = torch.nonzero(torch.rand(n, generator=g) < 0.5).view(-1)
        nt_a, nt_b = attributed[:-1], attributed[1:]
        src = torch.cat([parent, child, nt_a, nt_b]) + o
        dst = torch.cat([child, parent, nt_b, nt_a]) + o
        parts.append(torch.stack([src, dst]))
    ei = torch.cat(parts, dim=1)
    batch = torch.repeat_interleave(torch.arange(n_graphs), sizes)
    return ei, int(sizes.sum()), batch


def knn_superpixel_batch(n_graphs: int = 2048, k: int = 8, lo: int = 85, hi: int = 150, seed: int = 0):
    """BASELINE config 4 (CIFAR10-superpixel-like): k-NN graphs of random 2-D points, k in-neighbours
    per node.  Returns (edge_index, n_nodes, batch)."""
    g = torch.Generator().manual_seed(seed)
    sizes = torch.randint(lo, hi + 1, (n_graphs,), generator=g)
    offs = torch.cumsum(sizes, 0) - sizes
    srcs, dsts = [], []
    for i in range(n_graphs):
        n, o = int(sizes[i]), int(offs[i])
        pts = torch.rand(n, 2, generator=g)
        d = torch.cdist(pts, pts)
        d.fill_diagonal_(float("inf"))
        nbr = d.topk(k, largest=False).indices  # [n, k] sources for each destination
        dsts.append(torch.arange(n).repeat_interleave(k) + o)
        srcs.append(nbr.reshape(-1) + o)
    ei = torch.stack([torch.cat(srcs), torch.cat(dsts)])
    batch = torch.repeat_interleave(torch.arange(n_graphs), sizes)
    return ei, int(sizes.sum()), batch


def algorithmic_bytes(n_nodes: int, e_eff: int, f_in: int, f_g: int, f_out: int, w_cols: int, symnorm: bool,
                      idx_bytes: int = 4) -> dict:
    """SURVEY.md 8(d) byte model, term by term.  ``layer`` is the figure of the survey (weightings
    counted as fused); ``aggregate_kernel`` is what the aggregate+combine launch itself must move
    (it reads the materialised weightings instead of x, and does not write bases)."""
    t = dict(
        gather=e_eff * f_g * 4, col=e_eff * idx_bytes, rowptr=(n_nodes + 1) * idx_bytes,
        deg=n_nodes * 4 if symnorm else 0, x=n_nodes * f_in * 4, bases_write=n_nodes * f_g * 4,
        out=n_nodes * f_out * 4, weightings=n_nodes * w_cols * 4)
    t["layer"] = t["gather"] + t["col"] + t["rowptr"] + t["deg"] + t["x"] + t["bases_write"] + t["out"]
    t["aggregate_kernel"] = t["gather"] + t["col"] + t["rowptr"] + t["deg"] + t["weightings"] + t["out"]
    t["gemm_kernel"] = t["x"] + t["bases_write"] + t["weightings"]
    return t


MAG_NODES = 736_389            # ogbn-mag paper nodes (what `main.py egc mag` trains on, mag/configs.py:73-88)
MAG_DIRECTED_EDGES = 5_416_271


def mag_like(seed: int = 0, communities: int = 0, p_in: float = 0.0) -> tuple[torch.Tensor, int]:
    """BASELINE config 5, homogeneous form: N = 736,389, ~10.8 M symmetrised heavy-tailed edges;
    ``communities`` / ``p_in``: the variant with planted community structure (see heavy_tailed_graph)."""
    return heavy_tailed_graph(MAG_NODES, MAG_DIRECTED_EDGES, seed, communities, p_in), MAG_NODES


RMAG_NODES = {"paper": 736_389, "author": 1_134_649, "institution": 8_740, "field_of_study": 59_965}
RMAG_RELATIONS = {  # ogbn-mag's four edge types and their sizes (21.1 M directed edges in all)
    ("author", "affiliated_with", "institution"): 1_043_998,
    ("author", "writes", "paper"): 7_145_660,
    ("paper", "cites", "paper"): 5_416_271,
    ("paper", "has_topic", "field_of_study"): 7_505_078,
}


def rmag_like(seed: int = 0, scale: float = 1.0):
    """BASELINE config 5, heterogeneous form (the "~21 M edge" graph the relational path takes,
    rmag/models.py:10-26): ogbn-mag's node and edge counts, heavy-tailed destinations, uniform sources, plus
    the reverse of every relation except ``cites`` (which is symmetrised) -- the seven relations of the
    reference.  Returns ({node type: count}, {(src, rel, dst): int64 [2, E] with row 0 = source ids, row 1 =
    destination ids}).  ``scale`` shrinks every count (tests)."""
    g = torch.Generator().manual_seed(seed)
    nodes = {k: max(2, int(v * scale)) for k, v in RMAG_NODES.items()}
    rel = {}
    for (s, name, d), e in RMAG_RELATIONS.items():
        e = max(1, int(e * scale))
        u = torch.rand(e, generator=g, dtype=torch.float64)
        dst = (nodes[d] * u ** 3).long().clamp_(max=nodes[d] - 1)
        src = torch.randint(0, nodes[s], (e,), generator=g)
        if s == d:
            a, b = torch.cat([src, dst]), torch.cat([dst, src])
            key = torch.unique(a * nodes[d] + b)
            rel[(s, name, d)] = torch.stack([key // nodes[d], key % nodes[d]])
        else:
            key = torch.unique(src * nodes[d] + dst)
            src, dst = key // nodes[d], key % nodes[d]
            rel[(s, name, d)] = torch.stack([src, dst])
            rel[(d, "to", s)] = torch.stack([dst, src])
    return nodes, rel

"""1-D vertex partition of a full graph over the ranks of one node, with a halo exchange of basis rows.

New capability relative to the reference (which is single-device: SURVEY.md 2.3); the contract is
``SURVEY.md`` 8(e): rank p owns a contiguous vertex range -- its rows of x / bases / weightings / out and
the CSR rows of its vertices.  Column ids are remapped to ``[owned rows | halo rows]``; between the basis
GEMM and the fused aggregate kernel every rank receives the ``bases`` rows of its halo vertices with ONE
all-to-all-v (``torch.distributed.all_to_all_single``: RCCL over xGMI on GPUs, gloo in the CPU tests).
No other collective is on the forward path.  Partition, send/receive lists and the halo ``deg^-1/2`` are
static and built once per graph (the analogue of the reference's ``cached=True``).

Overlap: with ``interior_first=True`` the owned vertices are renumbered so that INTERIOR rows (every source
owned) come first and BOUNDARY rows (at least one halo source) last.  The forward then starts the all-to-all-v
right after the GEMM, finishes the interior rows while the halo rows travel (``egc_aggregate_combine_rows_f32``),
and only the boundary rows wait for the exchange.  ``HaloPlan.order`` maps new local positions to the old ones
(``x_new = x_owned[order]``, ``out_owned[order] = out_new``): a network permutes its inputs once and its outputs once.

Correctness criterion: concatenating the per-rank outputs reproduces the single-GPU output (max/min
exactly, sums to rounding since a row's neighbour order is unchanged).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch


def cost_balanced_bounds(cost: torch.Tensor, world: int) -> List[int]:
    """Boundaries of `world` contiguous vertex ranges of (nearly) equal total `cost` (e.g. in-degree + a per-row
    constant: what a rank's aggregate launch and GEMM scale with), instead of equal vertex counts."""
    n = int(cost.numel())
    if n == 0:
        return [0] * (world + 1)
    c = torch.cumsum(cost.double(), 0)
    targets = c[-1] * torch.arange(1, world, dtype=torch.float64, device=cost.device) / world
    cuts = (torch.searchsorted(c, targets) + 1).clamp_(max=n).tolist()   # first prefix that reaches the share
    bounds = [0] + [int(v) for v in cuts] + [n]
    for i in range(1, world + 1):   # monotone even when single vertices outweigh a share
        bounds[i] = max(bounds[i], bounds[i - 1])
    return bounds


def _mix32(v: torch.Tensor, salt: int) -> torch.Tensor:
    """Cheap deterministic per-vertex hash (every rank must draw the same moves without a shared RNG)."""
    h = (v * 2654435761 + salt * 40503) & 0xFFFFFFFF
    h = ((h ^ (h >> 15)) * 2246822519) & 0xFFFFFFFF
    return h ^ (h >> 13)


def _label_sums(lab: torch.Tensor, cost64: torch.Tensor, n: int) -> torch.Tensor:
    """[n] float64: total cost of every label value.  Sort + running sum instead of index_add: once labels have merged
    into clusters of 10^5 vertices, float64 atomics onto a handful of addresses take seconds per call on the GPU (the
    setup of the ogbn-mag-sized partition spent 9 of its 14 s there at two ranks)."""
    order = torch.argsort(lab)
    ls = lab[order]
    cs = torch.cumsum(cost64[order], 0)
    u, counts = torch.unique_consecutive(ls, return_counts=True)
    ends = torch.cumsum(counts, 0) - 1
    upto = cs[ends]
    tot = upto.clone()
    tot[1:] -= upto[:-1]
    out = torch.zeros(n, dtype=torch.float64, device=lab.device)
    out[u] = tot
    return out


def _part_loads(part: torch.Tensor, cost64: torch.Tensor, parts: int) -> torch.Tensor:
    """[parts] float64 total cost per part: one masked reduction per part (never atomics onto `parts` addresses)."""
    return torch.stack([(cost64 * (part == p)).sum() for p in range(parts)])


def _cluster_labels(src, dst, n, cost, max_cost, rounds):
    """Size-capped label propagation: every vertex starts as its own cluster; per round a hashed half of the
    vertices adopts the most frequent label among its in-neighbours (ties by a hash), clusters whose cost has
    reached `max_cost` admit nobody new."""
    dev = src.device
    ids = torch.arange(n, device=dev)
    lab = ids.clone()
    cost64 = cost.double()
    for r in range(rounds):
        uk, cnt = torch.unique(dst * n + lab[src], return_counts=True)
        d, l = uk // n, uk % n
        ccost = _label_sums(lab, cost64, n)
        closed = (ccost[l] >= max_cost) & (l != lab[d])
        score = cnt.double() + (_mix32(l, r).double() / 4294967296.0)
        score = torch.where(closed, torch.full_like(score, -1.0), score)
        best = torch.full((n,), -2.0, dtype=torch.float64, device=dev).scatter_reduce(0, d, score, "amax", include_self=True)
        win = (score == best[d]) & (score > 0) & ((_mix32(d, 977 + r) & 1) == 0)
        lab[d[win]] = l[win]
    return lab


def _pack_lpt(costs_desc: torch.Tensor, parts: int, exact: int = 4096) -> torch.Tensor:
    """Packing of clusters (sorted by decreasing cost) onto `parts` bins: longest-processing-time first for the
    `exact` largest (a host loop), the long tail of small clusters dealt out so that every bin ends up with the same
    total (vectorised: position of the cluster's midpoint in the cumulative cost of the tail, shifted by what the bins
    already hold)."""
    import heapq
    m = int(costs_desc.numel())
    out = torch.empty(m, dtype=torch.int64)
    loads = [0.0] * parts
    heap = [(0.0, p) for p in range(parts)]
    k = min(m, exact)
    for i, c in enumerate(costs_desc[:k].tolist()):
        load, p = heapq.heappop(heap)
        out[i] = p
        loads[p] = load + c
        heapq.heappush(heap, (load + c, p))
    if m > k:
        tail = costs_desc[k:].double()
        total = float(tail.sum()) + sum(loads)
        share = total / parts
        room = torch.tensor([max(share - l, 0.0) for l in loads], dtype=torch.float64)   # what each bin still takes
        edges = torch.cumsum(room, 0)
        mid = torch.cumsum(tail, 0) - tail / 2
        out[k:] = torch.searchsorted(edges, mid.clamp(max=float(edges[-1]) * (1 - 1e-12))).clamp_(max=parts - 1)
    return out


def locality_partition(edge_index: torch.Tensor, n_nodes: int, world: int, rounds: int = 8, row_cost: float = 4.0,
                       slack: float = 1.03, cluster_rounds: int = 8):
    """Locality-improving, work-balanced vertex partition, computed with scatter operations on the device that holds
    `edge_index` ("locality-improving reorder before the contiguous split is needed").

    Two stages.  Communities are found by size-capped label propagation from singleton labels and packed onto the
    parts largest-first; then balanced label propagation refines the parts: every round builds the [N, world]
    histogram of the parts of each vertex's in-neighbours (one index_add over the edges) and lets a hashed half of
    the vertices move to the part that holds most of their neighbours, parts above `slack` x the mean cost being
    closed to newcomers.  cost(v) = in-degree + row_cost.  The result is returned as
    a RENUMBERING: `order` (old id at each new position; parts are contiguous ranges of new ids, vertices keep
    their relative old order inside a part), `new_of_old`, and `bounds` (cost-balanced range boundaries in new ids).
    Deterministic for a given input (integer-valued sums, hashed move sets), so every rank derives the same
    partition from the same edge list without communication.

    On a graph without community structure (SURVEY 8d's synthetic) the halo barely shrinks -- there is no locality to
    find -- but the cost balance still matters: the heavy-tailed hubs sit at low ids and a count-balanced
    contiguous split gives the first rank 2.5x the mean number of entries at world 8."""
    dev = edge_index.device
    src, dst = edge_index[0], edge_index[1]
    n, P = int(n_nodes), int(world)
    ones = torch.ones(src.numel(), dtype=torch.float32, device=dev)
    deg = torch.zeros(n, dtype=torch.float32, device=dev).index_add_(0, dst, ones)
    cost = deg + float(row_cost)
    ids = torch.arange(n, device=dev)
    if P > 1 and cluster_rounds > 0 and src.numel() > 0:
        # (1) communities: size-capped label propagation from singleton labels (sort-based mode of the in-neighbours'
        #     labels), (2) packed into the parts largest-first onto the least loaded part
        lab = _cluster_labels(src, dst, n, cost, float(cost.sum()) / (8.0 * P), cluster_rounds)
        cl, inv = torch.unique(lab, return_inverse=True)
        ccost = _label_sums(lab, cost.double(), n)[cl]
        o = torch.argsort(ccost, descending=True, stable=True)
        bins = _pack_lpt(ccost[o].cpu(), P)
        part_of_cluster = torch.empty(cl.numel(), dtype=torch.int64, device=dev)
        part_of_cluster[o] = bins.to(dev)
        part = part_of_cluster[inv]
    else:
        b0 = cost_balanced_bounds(cost, P)
        part = torch.bucketize(ids, torch.tensor(b0[1:-1], device=dev, dtype=ids.dtype), right=True)
    cap = float(cost.sum()) / P * slack
    for r in range(rounds if P > 1 else 0):
        hist = torch.zeros(n * P, dtype=torch.float32, device=dev).index_add_(0, dst * P + part[src], ones).view(n, P)
        load = _part_loads(part, cost.double(), P)
        open_ = (load < cap).to(hist.dtype)                                  # full parts take no newcomers
        own = hist.gather(1, part[:, None]).squeeze(1)
        score = hist * open_[None, :]
        best_val, best = score.max(dim=1)
        movers = (best_val > own) & ((_mix32(ids, r) & 1) == 0)              # strict gain, hashed half per round
        # admit movers into a part only while it stays under the cap (largest gain first)
        if bool(movers.any()):
            mv = movers.nonzero().squeeze(1)
            gain = (best_val - own)[mv]
            tgt = best[mv]
            key = tgt.double() * 1e9 - gain.double()
            o = torch.argsort(key, stable=True)
            mv, tgt = mv[o], tgt[o]
            c = cost[mv].double()
            seg_start = torch.searchsorted(tgt, torch.arange(P, device=dev))
            run = torch.cumsum(c, 0)
            base = torch.where(seg_start < mv.numel(), run[seg_start.clamp(max=max(mv.numel() - 1, 0))] - c[seg_start.clamp(max=max(mv.numel() - 1, 0))],
                               torch.zeros((), dtype=run.dtype, device=dev))
            within = run - base[tgt]
            ok = within <= (cap - load[tgt]).clamp(min=0)
            part[mv[ok]] = tgt[ok]
    order = torch.argsort(part, stable=True)
    new_of_old = torch.empty_like(order)
    new_of_old[order] = ids
    bounds = cost_balanced_bounds(cost[order], P)
    return order, new_of_old, bounds


def agree_on_partition(order: torch.Tensor, new_of_old: torch.Tensor, bounds: List[int], group=None):
    """Make rank 0's renumbering the one every rank uses (one broadcast of N ids at setup, not on the data path).
    locality_partition is deterministic for a given input, so this changes nothing when all ranks ran the same
    software on the same edge list -- it is there because a disagreement would not fail cleanly: ranks with
    different `bounds` ask each other for rows that do not exist and the all-to-all-v of the layer never matches up."""
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return order, new_of_old, list(bounds)
    dev = new_of_old.device
    staged = dev.type == "cuda" and dist.get_backend(group) == "gloo"     # (testing aid, see HaloPlan.exchange_start)
    noo = new_of_old.cpu() if staged else new_of_old.clone()
    b = torch.tensor(list(bounds), dtype=torch.int64, device=noo.device)
    dist.broadcast(noo, src=0, group=group)
    dist.broadcast(b, src=0, group=group)
    noo = noo.to(dev)
    order = torch.empty_like(noo)
    order[noo] = torch.arange(noo.numel(), device=dev, dtype=noo.dtype)
    return order, noo, [int(v) for v in b.tolist()]


def partition_quality(edge_index: torch.Tensor, bounds: List[int]) -> dict:
    """Halo statistics of a contiguous split with the given boundaries (diagnostics; what bench.py reports)."""
    dev = edge_index.device
    b = torch.tensor(bounds[1:-1], device=dev, dtype=edge_index.dtype)
    ps, pd = torch.bucketize(edge_index[0], b, right=True), torch.bucketize(edge_index[1], b, right=True)
    world = len(bounds) - 1
    cross = ps != pd
    n = bounds[-1]
    # distinct (destination part, remote source) pairs = halo rows summed over ranks
    key = torch.unique(pd[cross] * n + edge_index[0][cross])
    halo = torch.bincount(key // n, minlength=world)
    pair = torch.unique(key // n * world + torch.bucketize(key % n, b, right=True) + (key % n) * world * world)
    peer = torch.bincount((pair % (world * world)), minlength=world * world)
    entries = torch.bincount(pd, minlength=world)
    return dict(cross_edge_frac=float(cross.float().mean()) if cross.numel() else 0.0,
                halo_rows_per_rank=[int(v) for v in halo.tolist()],
                max_peer_rows=int(peer.max()) if peer.numel() else 0,
                entries_per_rank=[int(v) for v in entries.tolist()],
                rows_per_rank=[bounds[i + 1] - bounds[i] for i in range(world)])


def vertex_ranges(n_nodes: int, world: int) -> List[int]:
    """Boundaries of `world` contiguous, near-equal vertex ranges: [b_0 = 0, b_1, ..., b_world = N]."""
    base, rem = divmod(n_nodes, world)
    bounds = [0]
    for p in range(world):
        bounds.append(bounds[-1] + base + (1 if p < rem else 0))
    return bounds


@dataclass
class HaloPlan:
    """Everything one rank needs to exchange halo rows.  Index tensors live on the compute device."""
    rank: int
    world: int
    lo: int
    hi: int
    n_global: int
    halo_global_ids: torch.Tensor            # [n_halo] sorted global ids this rank gathers from others
    recv_splits: List[int]                   # halo rows owned by each rank (sums to n_halo)
    send_idx: torch.Tensor                   # [n_send] LOCAL row ids to ship, grouped by destination rank
    send_splits: List[int]                   # rows shipped to each rank
    group: Optional[object] = None           # torch.distributed process group (None = default / local simulation)
    stats: dict = field(default_factory=dict)
    # interior-first renumbering of the owned vertices (None: natural order)
    order: Optional[torch.Tensor] = None       # [n_local] old local id at each new position
    new_of_old: Optional[torch.Tensor] = None  # [n_local] inverse
    n_interior: Optional[int] = None           # rows [0, n_interior) read no halo row
    live: bool = False                         # built by build_distributed: the exchange calls go to the process group

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    @property
    def n_halo(self) -> int:
        return int(self.halo_global_ids.numel())

    # -- the one collective of the forward path ------------------------------------------------
    def exchange_start(self, table: torch.Tensor):
        """Begin filling rows [n_local, n_local + n_halo) of `table` with the owners' rows (rows [0, n_local) are
        this rank's); returns a handle for exchange_finish.  Work queued on the current stream after this call
        runs concurrently with the transfer and must not touch the halo rows."""
        import torch.distributed as dist
        n = self.n_local
        send = self.pack_send(table)
        recv = table[n:n + self.n_halo]
        width = 1 if table.dim() == 1 else table.size(1)
        out_splits = [r * width for r in self.recv_splits]
        in_splits = [s * width for s in self.send_splits]
        if table.is_cuda and dist.get_backend(self.group) == "gloo":
            # functional-testing aid only (several ranks sharing one GPU): gloo has no device all-to-all
            r_cpu = torch.empty(recv.numel(), dtype=recv.dtype)
            dist.all_to_all_single(r_cpu, send.view(-1).cpu(), out_splits, in_splits, group=self.group)
            recv.view(-1).copy_(r_cpu)
            return None
        work = dist.all_to_all_single(recv.view(-1), send.view(-1), out_splits, in_splits, group=self.group,
                                      async_op=True)
        return work, send  # `send` must stay alive until the transfer has completed

    def pack_send(self, table: torch.Tensor) -> torch.Tensor:
        """The send buffer of the exchange: the owned rows other ranks read, grouped by destination rank.  On the GPU
        one launch of the library's own gather (egc_gather_rows_f32: whole rows as 16-byte pieces); CPU tensors (the
        gloo tests) take torch's index_select."""
        n = self.n_local
        if (table.is_cuda and table.dim() == 2 and table.dtype == torch.float32 and table.size(1) % 4 == 0
                and table.stride(1) == 1 and table.stride(0) % 4 == 0 and table.data_ptr() % 16 == 0
                and self.send_idx.dtype == torch.int64 and self.send_idx.is_cuda):
            from . import _C
            from .graph import _stream_ptr
            k = int(self.send_idx.numel())
            send = torch.empty((k, table.size(1)), dtype=torch.float32, device=table.device)
            if k:
                _C.check(_C.load().egc_gather_rows_f32(table.data_ptr(), int(table.stride(0)), self.send_idx.data_ptr(), k,
                                                       int(table.size(1)), send.data_ptr(), _stream_ptr(table.device)),
                         "egc_gather_rows_f32")
            return send
        return table[:n].index_select(0, self.send_idx.to(table.device)).contiguous()

    def predicted_exchange_ms(self, row_bytes: int, link_gbs: float = 153.0) -> float:
        """xGMI model of SURVEY.md 8(e): every peer pair has its own link (7 x ~153 GB/s per GPU), so one all-to-all-v
        takes about the LARGEST single peer message (sent or received) / link bandwidth."""
        peer = max(max(self.recv_splits, default=0), max(self.send_splits, default=0))
        return peer * row_bytes / (link_gbs * 1e9) * 1e3

    @staticmethod
    def exchange_finish(handle):
        """Make the current stream wait for the transfer started by exchange_start."""
        if handle is not None:
            handle[0].wait()

    def exchange_reverse(self, table: torch.Tensor) -> torch.Tensor:
        """The adjoint of `exchange` (backward pass): ship rows [n_local, n_local + n_halo) of `table` -- gradients
        accumulated for OTHER ranks' vertices -- back to their owners.  Returns [n_send, width]: row k belongs to
        the owned vertex send_idx[k] (one vertex may appear several times: once per rank that reads it)."""
        import torch.distributed as dist
        n = self.n_local
        send = table[n:n + self.n_halo].contiguous()
        width = 1 if table.dim() == 1 else table.size(1)
        recv = torch.empty((int(self.send_idx.numel()),) + tuple(table.shape[1:]), dtype=table.dtype, device=table.device)
        out_splits = [s * width for s in self.send_splits]
        in_splits = [r * width for r in self.recv_splits]
        if table.is_cuda and dist.get_backend(self.group) == "gloo":   # testing aid, see exchange_start
            r_cpu = torch.empty(recv.numel(), dtype=recv.dtype)
            dist.all_to_all_single(r_cpu, send.view(-1).cpu(), out_splits, in_splits, group=self.group)
            recv.view(-1).copy_(r_cpu)
        else:
            dist.all_to_all_single(recv.view(-1), send.view(-1), out_splits, in_splits, group=self.group)
        return recv

    def exchange(self, table: torch.Tensor):
        """exchange_start + exchange_finish.  `table` is [n_local + n_halo, width] (bases) or
        [n_local + n_halo] (deg^-1/2)."""
        self.exchange_finish(self.exchange_start(table))
        return table


def remap_sources(src_global: torch.Tensor, lo: int, hi: int, halo_global_ids: torch.Tensor) -> torch.Tensor:
    """Global source ids -> [owned rows | halo rows] ids of one rank."""
    n_local = hi - lo
    local = (src_global >= lo) & (src_global < hi)
    pos = torch.searchsorted(halo_global_ids, src_global.clamp(min=0))
    return torch.where(local, src_global - lo, n_local + pos)


def local_edges(edge_index: torch.Tensor, lo: int, hi: int):
    """Edges whose DESTINATION is owned by [lo, hi), in input order (keeps the first-edge tie-break)."""
    dst = edge_index[1]
    keep = (dst >= lo) & (dst < hi)
    return edge_index[:, keep]


def _halo_ids(ei_local: torch.Tensor, lo: int, hi: int, bounds: List[int]):
    src = ei_local[0]
    remote = src[(src < lo) | (src >= hi)]
    halo = torch.unique(remote)  # sorted => grouped by owner because ranges are contiguous
    b = torch.tensor(bounds, device=halo.device, dtype=halo.dtype)
    owner_counts = (torch.searchsorted(halo, b[1:], right=False) - torch.searchsorted(halo, b[:-1], right=False))
    return halo, [int(c) for c in owner_counts.tolist()]


def _interior_first(ei_local: torch.Tensor, plan: "HaloPlan"):
    """Renumber the owned vertices: rows without a halo source first.  Returns the renamed edge list."""
    n = plan.n_local
    dev = ei_local.device
    boundary = torch.zeros(n, dtype=torch.bool, device=dev)
    boundary[ei_local[1][ei_local[0] >= n]] = True
    order = torch.argsort(boundary.to(torch.int8), stable=True)          # old ids, interior rows first
    new_of_old = torch.empty_like(order)
    new_of_old[order] = torch.arange(n, device=dev)
    src = ei_local[0]
    src = torch.where(src < n, new_of_old[src.clamp(max=max(n - 1, 0))], src)
    plan.order, plan.new_of_old, plan.n_interior = order, new_of_old, int((~boundary).sum())
    if plan.send_idx.numel() > 0:
        plan.send_idx = new_of_old[plan.send_idx.to(dev)].contiguous()
    plan.stats["n_interior"] = plan.n_interior
    return torch.stack([src, new_of_old[ei_local[1]]])


def build_distributed(edge_index_owned: torch.Tensor, n_global: int, group=None, interior_first: bool = False,
                      bounds: Optional[List[int]] = None):
    """Collective setup.  `edge_index_owned`: the edges (GLOBAL ids, int64 [2, E_p]) whose destination this
    rank owns.  Returns (edge_index with [owned|halo] source ids and local destination ids, HaloPlan).
    `bounds`: the contiguous ranges (default: equal vertex counts; locality_partition returns cost-balanced ones)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = vertex_ranges(n_global, world) if bounds is None else list(bounds)
    lo, hi = bounds[rank], bounds[rank + 1]
    dev = edge_index_owned.device
    halo, recv_splits = _halo_ids(edge_index_owned, lo, hi, bounds)
    # tell every owner which of its rows we need: counts first, then the id lists
    cpu_stage = dev.type == "cuda" and dist.get_backend(group) == "gloo"  # see HaloPlan.exchange
    cdev = torch.device("cpu") if cpu_stage else dev
    want = torch.tensor(recv_splits, dtype=torch.int64, device=cdev)
    give = torch.empty_like(want)
    dist.all_to_all_single(give, want, group=group)
    send_splits = [int(c) for c in give.tolist()]
    req = torch.empty(sum(send_splits), dtype=torch.int64, device=cdev)
    dist.all_to_all_single(req, halo.contiguous().to(cdev), send_splits, recv_splits, group=group)
    req = req.to(dev)
    plan = HaloPlan(rank, world, lo, hi, n_global, halo, recv_splits, (req - lo).contiguous(), send_splits, group)
    plan.live = True
    plan.stats = dict(n_local=hi - lo, n_halo=int(halo.numel()), n_send=int(req.numel()),
                      max_peer_rows=max(recv_splits) if recv_splits else 0,
                      max_peer_rows_sent=max(send_splits) if send_splits else 0)
    src = remap_sources(edge_index_owned[0], lo, hi, halo)
    ei_local = torch.stack([src, edge_index_owned[1] - lo])
    return (_interior_first(ei_local, plan) if interior_first else ei_local), plan


def build_local_simulation(edge_index: torch.Tensor, n_global: int, world: int, interior_first: bool = False,
                           bounds: Optional[List[int]] = None):
    """All `world` partitions of a global graph inside ONE process (tests, single-GPU validation): returns
    a list of (edge_index_local, HaloPlan); exchange them with `simulate_exchange`."""
    bounds = vertex_ranges(n_global, world) if bounds is None else list(bounds)
    parts = []
    for p in range(world):
        lo, hi = bounds[p], bounds[p + 1]
        ei = local_edges(edge_index, lo, hi)
        halo, recv_splits = _halo_ids(ei, lo, hi, bounds)
        plan = HaloPlan(p, world, lo, hi, n_global, halo, recv_splits, torch.empty(0, dtype=torch.int64), [0] * world)
        src = remap_sources(ei[0], lo, hi, halo)
        ei_local = torch.stack([src, ei[1] - lo])
        parts.append((_interior_first(ei_local, plan) if interior_first else ei_local, plan))
    return parts


def simulate_exchange(tables: List[torch.Tensor], plans: List[HaloPlan]):
    """What the all-to-all-v does, without a process group: copy owners' rows into every halo region."""
    for t, plan in zip(tables, plans):
        n = plan.n_local
        for q, other in enumerate(plans):
            sel = (plan.halo_global_ids >= other.lo) & (plan.halo_global_ids < other.hi)
            if bool(sel.any()):
                rows = plan.halo_global_ids[sel] - other.lo
                if other.new_of_old is not None:
                    rows = other.new_of_old.to(rows.device)[rows]
                t[n:n + plan.n_halo][sel] = tables[q][:other.n_local].index_select(0, rows)
    return tables


# ---------------------------------------------------------------------------------------------------------------
# Typed (heterogeneous) graphs: relational EGC on a vertex partition (rmag/models.py:18-26,75-148 distributed)
# ---------------------------------------------------------------------------------------------------------------
@dataclass
class TypedLayout:
    """Every node type is cut into `world` contiguous ranges of its own ids; rank p owns range p of EVERY type.  The
    ranks' owned sets are laid end to end in one COMBINED id space (rank-major, then type in `node_types` order), so
    that the machinery of the homogeneous partition applies unchanged: one table of basis rows per rank,
    [owned rows of all types | halo rows of all types], ONE all-to-all-v per layer whatever the number of types."""
    node_types: List[str]
    counts: dict                       # type -> number of nodes
    type_bounds: dict                  # type -> [b_0 .. b_world]
    world: int
    rank_bounds: List[int] = field(default_factory=list)   # combined space: rank p owns [rank_bounds[p], rank_bounds[p+1])
    type_off: dict = field(default_factory=dict)           # type -> [world] offset of the type's block inside rank p's rows

    def __post_init__(self):
        self.rank_bounds, self.type_off = [0], {t: [] for t in self.node_types}
        for p in range(self.world):
            off = 0
            for t in self.node_types:
                self.type_off[t].append(off)
                off += self.type_bounds[t][p + 1] - self.type_bounds[t][p]
            self.rank_bounds.append(self.rank_bounds[-1] + off)

    @property
    def n_total(self) -> int:
        return self.rank_bounds[-1]

    def owned(self, t: str, rank: int):
        return self.type_bounds[t][rank], self.type_bounds[t][rank + 1]

    def combined_ids(self, t: str, ids: torch.Tensor) -> torch.Tensor:
        """Combined-space id of the type-`t` nodes `ids`."""
        b = torch.tensor(self.type_bounds[t], dtype=ids.dtype, device=ids.device)
        q = torch.bucketize(ids, b[1:-1], right=True)
        base = torch.tensor([self.rank_bounds[p] + self.type_off[t][p] - self.type_bounds[t][p] for p in range(self.world)],
                            dtype=ids.dtype, device=ids.device)
        return ids + base[q]


def typed_layout(counts: dict, relations: dict, world: int, node_types: Optional[List[str]] = None,
                 row_cost: float = 4.0) -> TypedLayout:
    """Cost-balanced cuts per node type: cost(v) = in-degree of v over every relation that targets its type +
    `row_cost` (what the rank's aggregate launches and GEMM scale with).  Deterministic from the edge lists, so every
    rank derives the same layout without communication.  No locality renumbering here (the homogeneous path has
    locality_partition): the typed synthetic has no community structure to find."""
    node_types = sorted(counts) if node_types is None else list(node_types)
    bounds = {}
    for t in node_types:
        dev = next((ei.device for (s, r, d), ei in relations.items() if d == t), torch.device("cpu"))
        cost = torch.full((counts[t],), float(row_cost), dtype=torch.float32, device=dev)
        for (s, r, d), ei in relations.items():
            if d == t and ei.numel():
                cost.index_add_(0, ei[1], torch.ones(ei.size(1), dtype=torch.float32, device=dev))
        bounds[t] = cost_balanced_bounds(cost, world)
    return TypedLayout(node_types, dict(counts), bounds, world)


@dataclass
class TypedPartition:
    """One rank's share of a typed graph: the halo plan over the combined id space and, per relation, the edge list
    with LOCAL target rows (position inside the rank's range of the target type) and source ids that index the
    rank's basis table [owned rows of all types | halo rows]."""
    layout: TypedLayout
    rank: int
    plan: HaloPlan
    rel_edges: dict                    # (src type, name, dst type) -> int64 [2, E_p]: table row of the source, local target row

    @property
    def n_table(self) -> int:
        return self.plan.n_local + self.plan.n_halo

    def n_owned(self, t: str) -> int:
        lo, hi = self.layout.owned(t, self.rank)
        return hi - lo

    def table_rows(self, t: str):
        """Row range of type `t`'s owned block inside the rank's table."""
        o = self.layout.type_off[t][self.rank]
        return o, o + self.n_owned(t)


def _typed_owned_edges(relations: dict, layout: TypedLayout, rank: int):
    """The rank's edges of every relation (target owned), in combined ids, concatenated; with the segment sizes."""
    keys, segs, parts = [], [], []
    for key, ei in relations.items():
        s, _, d = key
        lo, hi = layout.owned(d, rank)
        own = local_edges(ei, lo, hi)
        keys.append(key)
        segs.append(int(own.size(1)))
        parts.append(torch.stack([layout.combined_ids(s, own[0]), layout.combined_ids(d, own[1])]))
    dev = parts[0].device if parts else torch.device("cpu")
    return keys, segs, (torch.cat(parts, dim=1) if parts else torch.empty((2, 0), dtype=torch.int64, device=dev))


def _typed_split(keys, segs, ei_local, layout: TypedLayout, rank: int) -> dict:
    out, at = {}, 0
    for key, sz in zip(keys, segs):
        e = ei_local[:, at:at + sz]
        at += sz
        out[key] = torch.stack([e[0], e[1] - layout.type_off[key[2]][rank]])   # target row inside its type's block
    return out


def build_typed_distributed(relations: dict, layout: TypedLayout, group=None) -> TypedPartition:
    """Collective setup of the typed partition (two small all-to-alls, as build_distributed).  `relations`:
    {(src type, name, dst type): int64 [2, E] (source ids, target ids of the types' own id spaces)} -- the whole
    graph on every rank, or at least this rank's edges."""
    import torch.distributed as dist
    rank = dist.get_rank(group)
    keys, segs, combined = _typed_owned_edges(relations, layout, rank)
    ei_local, plan = build_distributed(combined, layout.n_total, group, interior_first=False, bounds=layout.rank_bounds)
    return TypedPartition(layout, rank, plan, _typed_split(keys, segs, ei_local, layout, rank))


def build_typed_local_simulation(relations: dict, layout: TypedLayout) -> List[TypedPartition]:
    """All ranks' typed partitions inside one process (tests, one-GPU validation); exchange with simulate_exchange
    over [p.plan for p in parts]."""
    parts = []
    for p in range(layout.world):
        keys, segs, combined = _typed_owned_edges(relations, layout, p)
        lo, hi = layout.rank_bounds[p], layout.rank_bounds[p + 1]
        halo, recv_splits = _halo_ids(combined, lo, hi, layout.rank_bounds)
        plan = HaloPlan(p, layout.world, lo, hi, layout.n_total, halo, recv_splits, torch.empty(0, dtype=torch.int64),
                        [0] * layout.world)
        plan.stats = dict(n_local=hi - lo, n_halo=int(halo.numel()), max_peer_rows=max(recv_splits) if recv_splits else 0)
        ei_local = torch.stack([remap_sources(combined[0], lo, hi, halo), combined[1] - lo])
        parts.append(TypedPartition(layout, p, plan, _typed_split(keys, segs, ei_local, layout, p)))
    return parts

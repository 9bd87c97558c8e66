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
        send = table[:n].index_select(0, self.send_idx).contiguous()
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


def build_distributed(edge_index_owned: torch.Tensor, n_global: int, group=None, interior_first: bool = False):
    """Collective setup.  `edge_index_owned`: the edges (GLOBAL ids, int64 [2, E_p]) whose destination this
    rank owns.  Returns (edge_index with [owned|halo] source ids and local destination ids, HaloPlan)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    bounds = vertex_ranges(n_global, world)
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
    plan.stats = dict(n_local=hi - lo, n_halo=int(halo.numel()), n_send=int(req.numel()),
                      max_peer_rows=max(recv_splits) if recv_splits else 0)
    src = remap_sources(edge_index_owned[0], lo, hi, halo)
    ei_local = torch.stack([src, edge_index_owned[1] - lo])
    return (_interior_first(ei_local, plan) if interior_first else ei_local), plan


def build_local_simulation(edge_index: torch.Tensor, n_global: int, world: int, interior_first: bool = False):
    """All `world` partitions of a global graph inside ONE process (tests, single-GPU validation): returns
    a list of (edge_index_local, HaloPlan); exchange them with `simulate_exchange`."""
    bounds = vertex_ranges(n_global, world)
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

"""Device-resident CSR graph container for the EGC hot path.

``CSRGraph`` owns the destination-keyed CSR (stable inside a row), the deg^-1/2 arrays for the
symnorm aggregator and the long-row work plan -- everything ``libegc_hip.so`` needs about the graph.
It replaces what the reference keeps in ``_AggLayer.cached_vals`` (experiments/layers.py:163,186-188),
``EGConv._cached_edge_index`` / ``_cached_adj_t`` (optimized_layers.py:71-72,138-175) and
``torch_sparse.SparseTensor`` storage (experiments/utils.py:107-113).  All buffers are torch tensors
(caching allocator); the kernels never allocate.
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _C


def _stream_ptr(device) -> int:
    """Raw hipStream_t of the caller's current stream on `device` (every launch of the library goes there)."""
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None:
        idx = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(idx)


class _NoGuard:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(device):
    """``torch.cuda.device(device)`` only when `device` is not already current (the context manager costs
    several microseconds per layer call, which is what small batched graphs are bound by)."""
    idx = device.index if isinstance(device, torch.device) else device
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(device)


class _IndexFlag:
    """The sticky out-of-range word of the graph builds (include/egc_hip.h: ``host_flag``): one int32 in pinned host
    memory per process.  A build that meets a node id outside its range drops the edge and stores 1 here from the
    device; the host reads the word -- a plain memory read, no synchronisation -- at the start of every later graph
    build and layer call and raises.  The reference's PyG path raises at ``index_select`` (optimized_layers.py:191-193)
    in the call itself; here the error surfaces at the next call into the package at the latest (like an
    asynchronous device error), never not at all."""
    _word = None
    _view = None

    @classmethod
    def ptr(cls) -> int:
        if cls._word is None:
            cls._word = torch.zeros(1, dtype=torch.int32).pin_memory()
            cls._view = C.c_int32.from_address(cls._word.data_ptr())
        return cls._word.data_ptr()

    @classmethod
    def poll(cls):
        if cls._view is not None and cls._view.value != 0:
            cls._view.value = 0
            raise RuntimeError("egc_amd: an earlier call reported malformed graph input: an edge_index with node ids outside "
                               "[0, num_nodes): index out of range (those edges were dropped) -- or, for a GraphBatch, edges that "
                               "leave their graph / a graph larger than max_nodes (its rows were written as zeros; "
                               "GraphBatch.check() tells which)")


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"egc_amd: {what} must live on a ROCm device (got {t.device}); the EGC hot path is HIP-only "
            "and has no CPU fallback")


class CSRGraph:
    """CSR by destination + degree statistics + long-row plan, all on one GPU."""

    def __init__(self, n_nodes, n_edges, rowptr, col, edge_id, dis_raw, dis_looped, max_index, plan, n_src_rows=None):
        self.n_nodes, self.n_edges = int(n_nodes), int(n_edges)
        # rows of the tables `col` indexes: owned rows first, then halo rows (vertex-partitioned runs)
        self.n_src_rows = int(n_src_rows) if n_src_rows is not None else int(n_nodes)
        self.halo = None  # egc_amd.partition.HaloPlan when the graph is one rank's partition
        self.rowptr, self.col, self.edge_id = rowptr, col, edge_id
        self.dis_raw, self.dis_looped, self.max_index, self.plan = dis_raw, dis_looped, max_index, plan
        self.device = rowptr.device
        self.edge_dis_raw = self.edge_dis_looped = None  # dis_*[col[p]] per entry (refresh_edge_dis)
        self._workspaces = {}
        self._n_chunks = None  # host copy of plan[1]; -1 = not read back (see c_struct)

    # -- construction ---------------------------------------------------------------------
    @classmethod
    def from_edge_index(cls, edge_index: torch.Tensor, num_nodes: int, num_src_rows: int | None = None,
                        build: str = "auto", _poll: bool = True) -> "CSRGraph":
        """COO ``edge_index`` (int64 [2, E], row 0 = source, row 1 = destination) -> CSRGraph.
        ``num_src_rows`` (>= num_nodes) is the size of the source index space when it is larger than the
        set of rows (owned + halo vertices of a partition).

        ``build``: "fast" = egc_graph_build (histogram / scan / scatter over edge tiles, no library sort: what a
        collated batch of small graphs wants -- its edges arrive grouped by graph, so a tile's destinations fall in
        one LDS window; 46-71 us for the molhiv / CIFAR batches of 2048 graphs against 76-175 us), "sort" = the
        radix-sort pipeline (edges in arbitrary order over a big graph: 240 us for ogbn-arxiv, where the fast build's
        per-edge atomics and 20,000-entry hub rows take 1.4 ms), "auto" = fast for small edge lists and for sparse ones
        of up to 4M edges (E <= 10 N: batches of molecules / superpixel graphs), sort otherwise.  The environment
        variable EGC_GRAPH_BUILD (fast | sort) overrides the argument.  Both produce identical graphs."""
        _require_cuda(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError("egc_amd: edge_index must be an int64 tensor of shape [2, E]")
        if _poll:
            _IndexFlag.poll()
        lib = _C.load()
        dev = edge_index.device
        ei = edge_index.contiguous()
        n, e = int(num_nodes), int(ei.size(1))
        build = os.environ.get("EGC_GRAPH_BUILD", build)
        if build not in ("auto", "fast", "sort"):
            raise RuntimeError(f"egc_amd: build must be 'auto', 'fast' or 'sort', got {build!r}")
        if build == "auto":
            build = "fast" if (e <= _FAST_BUILD_SMALL or (e <= _FAST_BUILD_MAX_EDGES and e <= 10 * n)) else "sort"
        if build == "fast" and e <= _FAST_BUILD_MAX_EDGES:
            return cls._build_fast(ei, n, e, num_src_rows)
        with _device_guard(dev):
            # one int32 slab for everything integer (per-batch graphs: allocator calls and fills cost as much as
            # the kernels): rowptr | col | edge_id | max_index | long-row plan, each piece 16-byte aligned
            e1 = max(e, 1)
            plan_ints = int(lib.egc_plan_ints(n, e))
            sizes = (n + 1, e1, e1, 1, plan_ints, 1)
            offs, total = [], 0
            for sz in sizes:
                offs.append(total)
                total += (sz + 3) & ~3
            slab = torch.empty(total, dtype=torch.int32, device=dev)
            rowptr, col, edge_id, max_index, plan, status = (slab[o:o + sz] for o, sz in zip(offs, sizes))
            ws_bytes = lib.egc_coo_to_csr_workspace_bytes(n, e)
            ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
            _C.check(lib.egc_coo_to_csr_checked(ei[0].data_ptr(), ei[1].data_ptr(), e, n,
                                                int(num_src_rows) if num_src_rows is not None else 0, rowptr.data_ptr(),
                                                col.data_ptr(), edge_id.data_ptr(), max_index.data_ptr(), status.data_ptr(),
                                                _IndexFlag.ptr(), ws.data_ptr(), ws.numel(), _stream_ptr(dev)),
                     "egc_coo_to_csr_checked")
            g = cls._prepare(n, e, rowptr, col, edge_id, max_index, num_src_rows, plan=plan)
            g._status = status
            if os.environ.get("EGC_CHECK_INDICES", "0") not in ("", "0"):
                g.check_indices()
            return g

    @classmethod
    def _build_fast(cls, ei: torch.Tensor, n: int, e: int, num_src_rows) -> "CSRGraph":
        """egc_graph_build: CSR + degree tables + per-entry deg^-1/2 + long-row plan in ONE library call (five launches,
        no library sort) -- the per-batch path of the reference's batched nets.  Node ids are range-checked on the
        device; the flag is read at the graph's next synchronisation point (check_indices)."""
        lib = _C.load()
        dev = ei.device
        ns = n if num_src_rows is None else int(num_src_rows)
        if ns <= 0 and e > 0:
            raise RuntimeError("egc_amd: num_src_rows must be positive")
        square = ns == n
        with _device_guard(dev):
            e1 = max(e, 1)
            plan_ints = int(lib.egc_plan_ints(n, e))
            sizes = (n + 1, e1, e1, 1, plan_ints, 1)
            offs, total = [], 0
            for sz in sizes:
                offs.append(total)
                total += (sz + 3) & ~3
            slab = torch.empty(total, dtype=torch.int32, device=dev)
            rowptr, col, edge_id, max_index, plan, status = (slab[o:o + sz] for o, sz in zip(offs, sizes))
            nrow = max(n, ns, 1)
            nd, ee = (nrow + 3) & ~3, (e1 + 3) & ~3
            fslab = torch.empty(2 * nd + (2 * ee if square else 0), dtype=torch.float32, device=dev)
            dis_raw, dis_looped = fslab[:nrow], fslab[nd:nd + nrow]
            if nrow > n:      # halo entries: filled by their owners later (partition.HaloPlan.exchange)
                fslab[:2 * nd].zero_()
            edr = fslab[2 * nd:2 * nd + e1] if square else None
            edl = fslab[2 * nd + ee:2 * nd + ee + e1] if square else None
            ws = _build_workspace(dev, int(lib.egc_graph_build_workspace_bytes(n, e)))
            scratch = torch.empty(int(lib.egc_graph_build_scratch_bytes(e)), dtype=torch.uint8, device=dev)
            _C.check(lib.egc_graph_build(ei[0].data_ptr(), ei[1].data_ptr(), e, n, ns, rowptr.data_ptr(), col.data_ptr(),
                                         edge_id.data_ptr(), max_index.data_ptr(), dis_raw.data_ptr(), dis_looped.data_ptr(),
                                         edr.data_ptr() if edr is not None else None,
                                         edl.data_ptr() if edl is not None else None, plan.data_ptr(), status.data_ptr(),
                                         _IndexFlag.ptr(), ws.data_ptr(), ws.numel(), scratch.data_ptr(), scratch.numel(),
                                         _stream_ptr(dev)),
                     "egc_graph_build")
        g = cls(n, e, rowptr, col, edge_id, dis_raw, dis_looped, max_index, plan, ns)
        g.edge_dis_raw, g.edge_dis_looped = edr, edl
        g._status = status
        if os.environ.get("EGC_CHECK_INDICES", "0") not in ("", "0"):
            g.check_indices()
        return g

    def check_indices(self) -> "CSRGraph":
        """Raise if the graph was built from node ids outside [0, N) -- what the reference's PyG path does at
        ``index_select`` (optimized_layers.py:191-193).  Synchronises; called wherever the graph synchronises anyway
        (trim_launches: cached layers, adj_t inputs) and on every build under EGC_CHECK_INDICES=1.  Paths that never
        synchronise (per-batch graphs, recorded steps) are covered by the sticky host-visible flag (_IndexFlag), polled
        at every later build and layer call."""
        st = getattr(self, "_status", None)
        if st is not None and int(st.item()) != 0:
            if _IndexFlag._view is not None:
                _IndexFlag._view.value = 0      # reported here
            raise RuntimeError("egc_amd: edge_index holds node ids outside [0, num_nodes): index out of range")
        return self

    @classmethod
    def from_partition(cls, edge_index_local: torch.Tensor, plan, global_max_index: int | None = None,
                       exchange_dis: bool = True) -> "CSRGraph":
        """One rank's share of a vertex-partitioned graph (egc_amd.partition): rows = owned vertices,
        source ids in [owned | halo] order.  The halo entries of the deg^-1/2 tables come from their owners
        (one all-to-all-v at setup; pass exchange_dis=False when the caller simulates the exchange)."""
        g = cls.from_edge_index(edge_index_local, plan.n_local, plan.n_local + plan.n_halo)
        g.halo = plan
        if global_max_index is not None:  # add_remaining_self_loops infers N from the GLOBAL largest index
            if plan.order is not None and int(global_max_index) < plan.n_global - 1:
                # `row <= max_index` names the vertices with a global id <= the largest index only while local row i
                # is global vertex lo + i; interior-first renumbering breaks that for layers with loops_all_nodes = 0
                raise RuntimeError("egc_amd: interior_first partitions need global_max_index == n_global - 1 "
                                   "(trailing isolated vertices: build the partition with interior_first=False)")
            g.max_index.fill_(max(-1, min(int(global_max_index) - plan.lo, plan.n_local - 1)))
        if exchange_dis and plan.n_halo >= 0 and plan.world > 1:
            plan.exchange(g.dis_raw)
            plan.exchange(g.dis_looped)
            g.refresh_edge_dis()
        return g

    @classmethod
    def from_csr(cls, rowptr: torch.Tensor, col: torch.Tensor, num_nodes: int | None = None,
                 num_src_rows: int | None = None) -> "CSRGraph":
        """Existing CSR keyed by destination (``adj_t``: row = destination, col = source)."""
        _require_cuda(rowptr, "rowptr")
        dev = rowptr.device
        n = int(rowptr.numel() - 1) if num_nodes is None else int(num_nodes)
        e = int(col.numel())
        rowptr = rowptr.to(torch.int32).contiguous()
        col32 = col.to(torch.int32).contiguous() if e > 0 else torch.empty(1, dtype=torch.int32, device=dev)
        with _device_guard(dev):
            edge_id = torch.arange(max(e, 1), dtype=torch.int32, device=dev)
            max_index = torch.full((1,), n - 1, dtype=torch.int32, device=dev)
            return cls._prepare(n, e, rowptr, col32, edge_id, max_index, num_src_rows)

    @classmethod
    def _prepare(cls, n, e, rowptr, col, edge_id, max_index, n_src_rows=None, plan=None) -> "CSRGraph":
        lib = _C.load()
        dev = rowptr.device
        ns = n if n_src_rows is None else int(n_src_rows)
        if ns <= 0 and e > 0:
            raise RuntimeError("egc_amd: num_src_rows must be positive")
        # one float slab: deg^-1/2 tables (rows first, then -- partitioned runs -- the halo entries filled in by
        # their owners) | their per-entry copies (refresh_edge_dis)
        nd, e1 = (max(n, ns, 1) + 3) & ~3, (max(e, 1) + 3) & ~3
        square = ns == n
        fslab = torch.empty(2 * nd + (2 * e1 if square else 0), dtype=torch.float32, device=dev)
        dis_raw, dis_looped = fslab[:max(n, ns, 1)], fslab[nd:nd + max(n, ns, 1)]
        if max(n, ns, 1) > n:      # entries beyond the rows are not written by egc_csr_prepare
            fslab[:2 * nd].zero_()
        if plan is None:
            plan = torch.empty(lib.egc_plan_ints(n, e), dtype=torch.int32, device=dev)
        _C.check(lib.egc_csr_prepare(n, e, rowptr.data_ptr(), col.data_ptr(), dis_raw.data_ptr(),
                                     dis_looped.data_ptr(), plan.data_ptr(), _stream_ptr(dev)), "egc_csr_prepare")
        g = cls(n, e, rowptr, col, edge_id, dis_raw, dis_looped, max_index, plan, ns)
        if square:  # square adjacency: symnorm is meaningful; partitions refresh after the halo exchange of dis
            g.edge_dis_raw = fslab[2 * nd:2 * nd + max(e, 1)]
            g.edge_dis_looped = fslab[2 * nd + e1:2 * nd + e1 + max(e, 1)]
            g.refresh_edge_dis()
        return g

    def refresh_edge_dis(self):
        """(Re)build the per-entry copies of the source-side deg^-1/2 (egc_csr_edge_dis): the aggregate kernel then
        streams them instead of gathering dis[col[p]].  Call again whenever dis_raw / dis_looped change."""
        lib = _C.load()
        dev = self.device
        e = self.n_edges
        with _device_guard(dev):
            if self.edge_dis_raw is None:
                self.edge_dis_raw = torch.empty(max(e, 1), dtype=torch.float32, device=dev)
                self.edge_dis_looped = torch.empty(max(e, 1), dtype=torch.float32, device=dev)
            _C.check(lib.egc_csr_edge_dis(e, self.col.data_ptr(), self.dis_raw.data_ptr(), self.dis_looped.data_ptr(),
                                          self.edge_dis_raw.data_ptr(), self.edge_dis_looped.data_ptr(),
                                          _stream_ptr(dev)), "egc_csr_edge_dis")

    def transposed(self) -> "CSRGraph":
        """The transposed graph as a CSRGraph (rows = SOURCES, entries = destinations, with its own long-row
        plan), needed by the backward: d bases[j] sums the destination tables over j's out-neighbours.
        Built on first use from the destination-keyed CSR and kept."""
        if getattr(self, "_transposed", None) is None:
            dev = self.device
            n, e, ns = self.n_nodes, self.n_edges, self.n_src_rows
            with _device_guard(dev):
                # swap roles: "source" = the entry's row (becomes the entry), "destination" = its column (becomes the
                # row); one launch, no read-back (a per-batch graph pays this in every training step)
                coo = torch.empty((2, e), dtype=torch.int64, device=dev)
                _C.check(_C.load().egc_csr_transposed_coo(n, e, self.rowptr.data_ptr(), self.col.data_ptr(),
                                                          coo[0].data_ptr(), coo[1].data_ptr() if e else None,
                                                          _stream_ptr(dev)), "egc_csr_transposed_coo")
                self._transposed = CSRGraph.from_edge_index(coo, ns, max(n, 1), _poll=False)   # (derived, not user input)
        if self._n_chunks is not None and self._n_chunks >= 0:   # a static graph: its transpose is one too
            self._transposed.trim_launches()
        return self._transposed

    def workspace(self, nbytes: int, zero_bytes: int | None = None) -> torch.Tensor:
        """Scratch for egc_aggregate_combine_f32, kept per (size, stream).  The C ABI wants the first ``zero_bytes``
        bytes (egc_aggregate_workspace_zero_bytes: the long-row arrival counters) zero before the first use and leaves
        the buffer reusable afterwards; the rest -- capacity-sized chunk records, 60 MB for a CIFAR batch of 2048
        graphs -- is written before it is read and is not filled."""
        key = (int(nbytes), _stream_ptr(self.device))
        ws = self._workspaces.get(key)
        if ws is None:
            n = max(int(nbytes), 1)
            if zero_bytes is None or zero_bytes >= n:
                ws = torch.zeros(n, dtype=torch.uint8, device=self.device)
            else:
                ws = torch.empty(n, dtype=torch.uint8, device=self.device)
                ws[:int(zero_bytes)].zero_()
            self._workspaces[key] = ws
        return ws

    # -- C view -----------------------------------------------------------------------------
    def c_struct(self) -> _C.EgcGraph:
        _IndexFlag.poll()      # every layer call passes here: a bad edge_index of an earlier build surfaces now
        if self._n_chunks is None:
            # A host copy of the chunk count would trim the launch to the chunks that exist, at the price of one
            # synchronisation per graph.  Measured (ogbn-mag shape: 208 k chunk slots, 52 k idle workgroups per
            # launch), the idle workgroups cost nothing that shows: nothing is read back and no call ever synchronises.
            self._n_chunks = -1
        return _C.EgcGraph(self.n_nodes, self.n_edges, self.rowptr.data_ptr(), self.col.data_ptr(),
                           self.edge_id.data_ptr(), self.dis_raw.data_ptr(), self.dis_looped.data_ptr(),
                           self.max_index.data_ptr(), self.plan.data_ptr(), self._n_chunks, self.n_src_rows,
                           self.edge_dis_raw.data_ptr() if self.edge_dis_raw is not None else None,
                           self.edge_dis_looped.data_ptr() if self.edge_dis_looped is not None else None)

    def tensors(self) -> list:
        """The device tensors the C view points into (what a compiled autograd node keeps alive next to its copy of the struct)."""
        ts = [self.rowptr, self.col, self.edge_id, self.dis_raw, self.dis_looped, self.max_index, self.plan]
        return ts + [t for t in (self.edge_dis_raw, self.edge_dis_looped) if t is not None]

    def c_addr(self) -> int:
        """Address of a C view of this graph that stays valid (and current) while the graph lives: what the compiled
        binding takes (egc_amd/_native.py).  Rebuilt when something the struct carries has changed."""
        key = (self._n_chunks, 0 if self.edge_dis_raw is None else self.edge_dis_raw.data_ptr(),
               0 if self.edge_dis_looped is None else self.edge_dis_looped.data_ptr())
        hit = getattr(self, "_c_cached", None)
        if hit is None or hit[0] != key or self._n_chunks is None:
            st = self.c_struct()           # (polls the index flag, settles _n_chunks)
            key = (self._n_chunks, key[1], key[2])
            hit = self._c_cached = (key, st, C.addressof(st))
        else:
            _IndexFlag.poll()
        return hit[2]

    def workspace_for(self, spec) -> torch.Tensor:
        """The aggregate workspace of (this graph, this layer), sized once."""
        cache = self.__dict__.setdefault("_ws_for", {})
        key = (id(spec), _stream_ptr(self.device), self._n_chunks)
        ws = cache.get(key)
        if ws is None:
            lib = _C.load()
            g = self.c_struct()
            ws = self.workspace(lib.egc_aggregate_workspace_bytes_for(C.byref(spec.c), C.byref(g)),
                                lib.egc_aggregate_workspace_zero_bytes(C.byref(spec.c), self.n_nodes, self.n_edges))
            cache[(id(spec), _stream_ptr(self.device), self._n_chunks)] = ws
            cache.setdefault("_keep", []).append(spec)      # id(spec) stays unique while the entry lives
        return ws

    def trim_launches(self) -> "CSRGraph":
        """Read the long-row chunk count back to the host (ONE synchronisation) so that later launches carry only
        the chunk workgroups that exist instead of the plan's capacity -- for a graph that is built once and used
        many times (``cached=True`` layers, full-graph training); about 4 us per launch at ogbn-arxiv size."""
        if self._n_chunks is None or self._n_chunks < 0:
            self._n_chunks = int(self.plan[1].item())
            self.check_indices()
        return self

    def long_row_stats(self):
        """(n_long_rows, n_chunks) -- synchronises; diagnostics only."""
        h = self.plan[:2].cpu()
        return int(h[0]), int(h[1])


_FAST_BUILD_MAX_EDGES = 4_000_000   # beyond: always the radix-sort pipeline (full graphs: built once; never measured on the fast build)
_FAST_BUILD_SMALL = 262_144         # up to here the fast build wins whatever the edge order (launch-bound either way)
_BUILD_WS: "dict[tuple, torch.Tensor]" = {}


def _build_workspace(dev, nbytes: int) -> torch.Tensor:
    """Scratch of egc_graph_build: zero before its first use, left zero by every call -> one buffer per (device,
    stream), grown (and zeroed again) when a bigger graph comes along.  While a hipGraph is being recorded the buffer
    and its zero-fill belong to THAT recording (capture-pool memory and a memset node of the graph being captured):
    it is allocated and zeroed inside the recording and never enters the process-wide cache -- a later recording
    must not inherit a buffer whose zero-fill exists only as a node of an earlier graph."""
    if torch.cuda.is_current_stream_capturing():
        return torch.zeros(max(int(nbytes), 1 << 12), dtype=torch.uint8, device=dev)
    key = (str(dev), _stream_ptr(dev))
    ws = _BUILD_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(max(int(nbytes * 1.25), 1 << 20), dtype=torch.uint8, device=dev)
        _BUILD_WS[key] = ws
    return ws


def drop_stream_workspaces(stream) -> None:
    """Forget the build scratch cached for `stream` (a torch.cuda.Stream about to go away)."""
    raw = stream.cuda_stream
    for key in [k for k in _BUILD_WS if k[1] == raw]:
        del _BUILD_WS[key]


class SparseTensor:
    """Stand-in for ``torch_sparse.SparseTensor`` restricted to what the EGC layers need: an
    ``adj_t`` whose rows are destinations and columns are sources (experiments/utils.py:107-113,
    mag/configs.py:84-85).  Values are ignored -- the reference drops them too (utils.py:103-104)."""

    def __init__(self, row: torch.Tensor = None, col: torch.Tensor = None, value=None, sparse_sizes=None,
                 is_sorted: bool = False, rowptr: torch.Tensor = None):
        if sparse_sizes is None:
            raise RuntimeError("egc_amd.SparseTensor: sparse_sizes=(N_dst, N_src) is required")
        self._sizes = (int(sparse_sizes[0]), int(sparse_sizes[1]))
        if rowptr is not None:
            self.graph = CSRGraph.from_csr(rowptr, col, self._sizes[0], self._sizes[1])
        else:
            # row = destination, col = source  ->  edge_index = [col; row]
            self.graph = CSRGraph.from_edge_index(torch.stack([col.long(), row.long()]), self._sizes[0], self._sizes[1])
        self.graph.trim_launches()   # an adj_t is built once per graph (ToSparseTensor): one synchronisation here

    def sparse_sizes(self):
        return self._sizes

    def sparse_size(self, dim):
        return self._sizes[dim]

    def size(self, dim):
        return self._sizes[dim]


_TILE_SETUPS: "dict[tuple, object]" = {}     # (layer, batch shape) -> (slot, lds_nodes, tmax, emax) | False


# Status words of the batches: handed out from a zero-filled pool, one word per GraphBatch, never reused -- a torch.zeros(1)
# per batch is a fill launch of its own in front of the layer's launch (4.5 us of GPU time on the MI355X, a tenth of a molhiv
# batch's whole layer).  A pool is 1,024 words; a used-up pool lives as long as the batches that hold its words.
_STATUS_POOLS: dict = {}


def _status_word(device: torch.device) -> torch.Tensor:
    key = (device.type, device.index)
    pool = _STATUS_POOLS.get(key)
    if pool is None or pool[1] >= pool[0].numel():
        pool = [torch.zeros(1024, dtype=torch.int32, device=device), 0]
        _STATUS_POOLS[key] = pool
    i = pool[1]
    pool[1] = i + 1
    return pool[0][i:i + 1]


class GraphBatch:
    """A PyG-style BATCH of small graphs as the ``edge_index`` argument of the layers (``conv(x, GraphBatch(...))``),
    for the kernels that work on tiles of whole graphs (egc_aggregate_combine_batch_f32): no CSR is built per batch --
    each workgroup builds its tile's in LDS -- and a tile's basis rows are gathered from LDS.

    What the reference's batched nets hand to the layer is ``batch.edge_index`` of a PyG ``Batch``
    (zinc/models.py:60-74, mol/pna_style_models.py:64-79, cifar/models.py:61-75); the same ``Batch`` carries the graphs'
    node offsets (``batch.ptr``) -- pass them here:

        gb = egc_amd.GraphBatch(batch.edge_index, ptr=batch.ptr, max_nodes=max graph size)    # or batch=batch.batch,
        x = conv(x, gb)                                                                         # num_graphs=batch.num_graphs

    Requirements (checked on the device, reported like an out-of-range index -- a RuntimeError at the next call into the
    package at the latest; call ``gb.check()`` after the last layer of a batch to get it at once, with the cause; the rows
    of a tile that is reported are written as ZEROS, never left uninitialised): graphs are numbered one after the other and the edges of one graph are contiguous in
    ``edge_index`` (PyG's collation); a tile (a run of graphs of about ``slot`` nodes + one graph) fits the LDS areas.
    ``max_nodes``: an upper bound of the largest graph's node count (default 256) -- it sizes the per-tile CSR areas;
    tiles whose basis rows exceed the LDS area (a run of unusually large graphs) gather from memory instead, tiles beyond
    slot + max_nodes nodes or edges_per_node x that many edges are reported.  Layers outside the tile kernels' envelope, and every call that needs gradients, use the CSR of the same
    edge list instead (built on first need, ``csr()``): results are the same, the speed is the ordinary path's."""

    def __init__(self, edge_index: torch.Tensor, ptr: torch.Tensor | None = None, batch: torch.Tensor | None = None,
                 num_graphs: int | None = None, num_nodes: int | None = None, max_nodes: int = 256, edges_per_node: int = 16,
                 edge_ptr: torch.Tensor | None = None):
        _require_cuda(edge_index, "edge_index")
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.size(0) != 2:
            raise RuntimeError("egc_amd: edge_index must be an int64 tensor of shape [2, E]")
        self.edge_index = edge_index.contiguous()
        self.device = edge_index.device
        if ptr is None:
            if batch is None or num_graphs is None:
                raise RuntimeError("egc_amd.GraphBatch: pass ptr (node offsets of the graphs, int64 [G + 1]) or batch "
                                   "together with num_graphs")
            b = batch.to(self.device)
            ptr = torch.searchsorted(b, torch.arange(int(num_graphs) + 1, device=self.device, dtype=b.dtype))
            if num_nodes is None:
                num_nodes = int(batch.numel())
        self.ptr = ptr.to(device=self.device, dtype=torch.int64).contiguous()
        self.n_graphs = int(self.ptr.numel()) - 1
        # the graphs' edge offsets, when the caller has them (PyG's collation keeps them: batch._slice_dict["edge_index"]);
        # without them the plan finds each tile's first edge by searching the destination row
        self.edge_ptr = None if edge_ptr is None else edge_ptr.to(device=self.device, dtype=torch.int64).contiguous()
        if self.edge_ptr is not None and self.edge_ptr.numel() != self.ptr.numel():
            raise RuntimeError("egc_amd.GraphBatch: edge_ptr must have one entry per graph + 1, like ptr")
        self.n_nodes = None if num_nodes is None else int(num_nodes)     # else: the row count of the first x seen
        self.max_nodes, self.edges_per_node = int(max_nodes), int(edges_per_node)
        self.n_edges = int(edge_index.size(1))
        self.halo = None
        self._plans = {}
        self._setups = {}
        self._csr = None
        self._status = None
        self._max_index = None
        self._rows = None

    def rows(self):
        """(edge_index[0], edge_index[1]) as dense views, made once per batch (the compiled binding takes them as tensors)."""
        if self._rows is None:
            self._rows = (self.edge_index[0], self.edge_index[1])
        return self._rows

    @property
    def n_src_rows(self):
        return self.n_nodes

    def trim_launches(self) -> "GraphBatch":
        return self

    def csr(self) -> "CSRGraph":
        """The ordinary CSR of the same edge list (training, layers outside the tile kernels' envelope)."""
        if self._csr is None:
            self._csr = CSRGraph.from_edge_index(self.edge_index, self.n_nodes)
        return self._csr

    def max_index(self) -> torch.Tensor:
        if self._max_index is None:
            self._max_index = (self.edge_index.max().to(torch.int32).reshape(1) if self.n_edges
                               else torch.full((1,), -1, dtype=torch.int32, device=self.device))
        return self._max_index

    def status(self) -> torch.Tensor:
        if self._status is None:
            self._status = _status_word(self.device)
        return self._status

    def check(self) -> "GraphBatch":
        """Raise if a tile kernel has found the batch malformed (synchronises)."""
        if self._status is not None and int(self._status.item()) != 0:
            code = int(self._status.item())
            self._status.zero_()
            if _IndexFlag._view is not None:
                _IndexFlag._view.value = 0
            raise RuntimeError("egc_amd.GraphBatch: " + ("edge_index is not grouped by graph, or holds node ids outside its "
                               "graph: index out of range" if code & 1 else "a tile exceeds the per-tile areas (raise "
                               "max_nodes / edges_per_node, or pass the plain edge_index)" if code & 2 else
                               "the one-launch kernel's internal hand-over timed out (status bit 2): its output is undefined"))
        return self

    def tile_setup(self, spec_c, with_post: bool):
        """Everything egc_aggregate_combine_batch_f32 needs for a layer on this batch, or None when the layer is outside the
        tile kernels' envelope: (tiles, n_tiles (device), n_slots, lds_nodes, max_tile_nodes, max_tile_edges)."""
        key = (C.string_at(C.addressof(spec_c), C.sizeof(spec_c)), bool(with_post))
        hit = self._setups.get(key)
        if hit is None:
            # the sizing depends on the layer and on the batch's SHAPE only: batches of one loader share it
            gkey = key + (self.n_nodes, self.n_graphs, self.max_nodes, self.edges_per_node)
            hit = _TILE_SETUPS.get(gkey)
            if hit is not None:
                self._setups[key] = hit
        if hit is None:
            lib = _C.load()
            n, gcount = self.n_nodes, max(self.n_graphs, 1)
            typical = min(self.max_nodes, max(2 * -(-n // gcount), 8))
            epn = self.edges_per_node

            def areas(slot):
                tmax = min(2048, slot + self.max_nodes - 1)
                return tmax, min(tmax * epn, 16384)
            tmax, emax = areas(64)
            lds = int(lib.egc_batch_tile_nodes(C.byref(spec_c), tmax, emax, int(with_post)))
            if lds <= 0:
                hit = False
            else:
                # a slot + one typical graph fills the LDS area: tiles as large as LDS allows (every tile pays a fixed
                # latency chain -- requests, three LDS passes of the CSR build -- whatever its size); smaller only when the
                # batch would otherwise leave CUs without a tile
                slot = max(8, min(lds - typical + 1, max(16, -(-n // 256))))
                tmax, emax = areas(slot)
                lds = int(lib.egc_batch_tile_nodes(C.byref(spec_c), tmax, emax, int(with_post)))
                hit = (slot, lds, tmax, emax) if lds > 0 else False
            self._setups[key] = hit
            if len(_TILE_SETUPS) > 256:
                _TILE_SETUPS.clear()
            _TILE_SETUPS[key + (self.n_nodes, self.n_graphs, self.max_nodes, self.edges_per_node)] = hit
        if hit is False:
            return None
        slot, lds, tmax, emax = hit
        tiles, count, n_slots = self.plan(slot)
        return tiles, count, n_slots, lds, tmax, emax

    def fused_setup(self, spec_c, with_post: bool):
        """(tile_nodes, max_tile_edges) for egc_layer_forward_batch_fused_f32 -- the whole layer in one launch, no `bases` /
        `weightings` in memory, no plan launch -- or None when the layer is outside that kernel's envelope or the declared
        largest graph (``max_nodes``) does not fit the LDS image of a tile (the two-launch tile path or the CSR path then)."""
        key = (C.string_at(C.addressof(spec_c), C.sizeof(spec_c)), bool(with_post), "fused")
        hit = self._setups.get(key)
        if hit is None:
            gkey = key + (self.max_nodes, self.edges_per_node)
            hit = _TILE_SETUPS.get(gkey)
            if hit is None:
                lib = _C.load()
                emax, cap = 4096, 0
                for _ in range(3):
                    cap = int(lib.egc_batch_fused_tile_nodes(C.byref(spec_c), emax, int(with_post)))
                    if cap <= 0 or cap * self.edges_per_node <= emax:
                        break
                    emax = min(16384, -(-cap * self.edges_per_node // 1024) * 1024)
                hit = (cap, emax) if (cap >= self.max_nodes and cap > 0 and cap * self.edges_per_node <= emax) else False
                if hit is False and cap > 0:
                    # the generous edge areas cost rows: the tightest areas that hold the declared largest graph
                    quantum = max(16, int(lib.egc_batch_fused_tile_quantum(C.byref(spec_c))))
                    need = -(-self.max_nodes // quantum) * quantum
                    emax = max(64, -(-need * self.edges_per_node // 64) * 64)
                    cap = int(lib.egc_batch_fused_tile_nodes(C.byref(spec_c), emax, int(with_post))) if emax <= 16384 else 0
                    if cap >= need:
                        hit = (need, emax)
                if len(_TILE_SETUPS) > 256:
                    _TILE_SETUPS.clear()
                _TILE_SETUPS[gkey] = hit
            self._setups[key] = hit
        return hit or None

    def fused_bwd_setup(self, spec_c):
        """(tile_nodes, max_tile_edges) for egc_layer_backward_batch_fused_f32 -- the layer's backward as one tile-local launch --
        or None when the layer is outside that kernel's envelope or the declared largest graph does not fit its LDS image
        (which also holds the d bases rows: smaller tiles than the forward's)."""
        key = (C.string_at(C.addressof(spec_c), C.sizeof(spec_c)), "fused_bwd")
        hit = self._setups.get(key)
        if hit is None:
            gkey = key + (self.max_nodes, self.edges_per_node)
            hit = _TILE_SETUPS.get(gkey)
            if hit is None:
                lib = _C.load()
                need = -(-self.max_nodes // 16) * 16
                hit = False
                for emax in (4096, max(64, -(-need * self.edges_per_node // 64) * 64)):
                    if emax > 16384:
                        continue
                    cap = int(lib.egc_batch_fused_bwd_tile_nodes(C.byref(spec_c), emax))
                    if cap >= need and cap * self.edges_per_node <= emax:
                        hit = (cap, emax)
                        break
                    if cap >= need and need * self.edges_per_node <= emax:
                        hit = (need, emax)
                        break
                if len(_TILE_SETUPS) > 256:
                    _TILE_SETUPS.clear()
                _TILE_SETUPS[gkey] = hit
            self._setups[key] = hit
        return hit or None

    def plan(self, slot: int):
        """(tiles int32 [n_slots, 4], n_tiles device scalar, n_slots) for slots of `slot` nodes; one launch, built once."""
        hit = self._plans.get(slot)
        if hit is None:
            lib = _C.load()
            n = self.n_nodes
            n_slots = (n + slot - 1) // slot
            with _device_guard(self.device):
                buf = torch.empty(4 * max(n_slots, 1) + 4, dtype=torch.int32, device=self.device)
                tiles, count = buf[:4 * max(n_slots, 1)].view(-1, 4), buf[4 * max(n_slots, 1):]
                _C.check(lib.egc_batch_plan(self.ptr.data_ptr(), self.edge_ptr.data_ptr() if self.edge_ptr is not None else None,
                                            self.n_graphs, self.edge_index[1].data_ptr(), self.n_edges, n, slot,
                                            tiles.data_ptr(), n_slots, count.data_ptr(), _stream_ptr(self.device)),
                         "egc_batch_plan")
            hit = (tiles, count, n_slots)
            self._plans[slot] = hit
        return hit


class GraphCache:
    """Small LRU of CSRGraphs keyed by the identity of the ``edge_index`` tensor, so that the
    layers of one network share one COO->CSR conversion per batch (SURVEY.md call stack d).
    The cache keeps a reference to the key tensor, so its storage cannot be recycled under us."""

    def __init__(self, capacity: int = 4):
        self.capacity = capacity
        self._items: "OrderedDict[tuple, tuple]" = OrderedDict()

    def get(self, edge_index: torch.Tensor, num_nodes: int, static: bool = False) -> CSRGraph:
        # a STATIC graph (cached=True / cache=True layers: one full graph, built once) beyond the small sizes takes the
        # sort pipeline whatever its density: its edges come in arbitrary order and its hubs are long, which is where the
        # tile-based build degrades (1.4 ms against 240 us at the ogbn-arxiv shape); per-batch graphs keep "auto"
        build = "sort" if (static and int(edge_index.size(1)) > _FAST_BUILD_SMALL) else "auto"
        scope = None
        if edge_index.is_cuda and torch.cuda.is_current_stream_capturing():
            # A recording (hipGraph) must CONTAIN the build: a replay reads whatever the edge_index buffer holds then,
            # so a graph built before the recording would be stale.  Inside egc_amd.GraphedStep the layers of the step
            # share one recorded build (entries keyed by the recording); a recording made any other way gets a build
            # per layer call -- slower, never stale.
            scope = _RECORDING[0]
            if scope is None:
                return CSRGraph.from_edge_index(edge_index, num_nodes, build=build)
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape), int(num_nodes),
               edge_index.device, scope)
        hit = self._items.get(key)
        if hit is not None and hit[0] is edge_index:
            self._items.move_to_end(key)
            return hit[1]
        g = CSRGraph.from_edge_index(edge_index, num_nodes, build=build)
        self._items[key] = (edge_index, g)
        while len(self._items) > self.capacity:
            self._items.popitem(last=False)
        return g

    def clear(self):
        self._items.clear()

    def drop_recording(self, scope):
        """Forget the graphs built inside a finished recording (their buffers belong to its memory pool)."""
        for key in [k for k in self._items if k[-1] is scope]:
            del self._items[key]


_RECORDING = [None]     # the recording in progress (egc_amd.hipgraph.GraphedStep), or None
GLOBAL_GRAPH_CACHE = GraphCache()


class recording_scope:
    """Marks the hipGraph recording in progress for the graph cache (used by egc_amd.hipgraph.GraphedStep)."""

    def __enter__(self):
        self._token = object()
        self._outer, _RECORDING[0] = _RECORDING[0], self._token
        return self

    def __exit__(self, *exc):
        _RECORDING[0] = self._outer
        GLOBAL_GRAPH_CACHE.drop_recording(self._token)
        return False


def graph_from_input(edge_index, num_nodes: int, static: bool = False) -> CSRGraph:
    """Dispatch on the two input forms of the reference layers (Tensor COO or SparseTensor adj_t).  ``static``: the
    caller keeps the graph (cached layers)."""
    if isinstance(edge_index, CSRGraph):
        return edge_index
    if isinstance(edge_index, GraphBatch):
        if edge_index.n_nodes is None:
            edge_index.n_nodes = int(num_nodes)
        elif edge_index.n_nodes != int(num_nodes):
            raise RuntimeError(f"egc_amd.GraphBatch: built for {edge_index.n_nodes} nodes, x has {int(num_nodes)} rows")
        return edge_index
    if isinstance(edge_index, SparseTensor):
        return edge_index.graph
    if not isinstance(edge_index, torch.Tensor) and callable(getattr(edge_index, "csr", None)):
        # a real ``torch_sparse.SparseTensor`` (mag/configs.py:84-85 builds ``adj_t`` with ``ToSparseTensor``; utils.py:107-113):
        # duck-typed on its public API -- ``.csr() -> (rowptr, col, value)`` with rows = destinations of adj_t,
        # ``.sparse_sizes() -> (N_dst, N_src)``.  Values are ignored, as the reference drops them (utils.py:103-104).
        # The conversion is cached on the object: an adj_t is built once per graph.
        hit = getattr(edge_index, "_egc_amd_graph", None)
        if hit is None:
            rowptr, col, _ = edge_index.csr()
            sizes = tuple(edge_index.sparse_sizes()) if callable(getattr(edge_index, "sparse_sizes", None)) else (int(num_nodes),) * 2
            if int(sizes[0]) != int(num_nodes):
                raise RuntimeError(f"egc_amd: adj_t has {int(sizes[0])} rows, x has {int(num_nodes)}")
            hit = CSRGraph.from_csr(rowptr, col, int(sizes[0]), int(sizes[1])).trim_launches()
            try:
                edge_index._egc_amd_graph = hit
            except AttributeError:      # (an object with __slots__: convert on every call)
                pass
        return hit
    if isinstance(edge_index, torch.Tensor):
        if edge_index.layout == torch.sparse_csr:
            return CSRGraph.from_csr(edge_index.crow_indices(), edge_index.col_indices(), num_nodes)
        return GLOBAL_GRAPH_CACHE.get(edge_index, num_nodes, static)
    raise RuntimeError(f"egc_amd: unsupported edge_index type {type(edge_index)}")

"""A whole step of a network of EGC layers -- forward, backward, per-batch graph build -- as ONE hipGraph.

The reference trains its batched nets at 32-128 graphs per batch (zinc/configs.py, molhiv/configs.py): a few
thousand nodes per step, for which every kernel of this library takes microseconds and the step is bound by what
launches them (Python, the autograd engine, ~25 launches per layer and direction).  Everything the library does on
the training path is stream-ordered device work with host-known sizes -- no allocation outside torch's caching
allocator, no host read-back (the index check of a fresh graph is deferred, graph.check_indices) -- so the step can
be recorded once and replayed: 4 x [EGConv -> BatchNorm1d(train) -> ReLU -> + x], forward + backward on a 128-molecule
batch (graph build included), 1.1-1.7 ms eager -> 0.41-0.48 ms replayed on MI355X, gradients bit-identical (DESIGN.md
section 5).

Usage (static shapes: the tensors the step reads and writes are the SAME objects at every replay; a batch with fewer
nodes / edges is padded by the caller, e.g. with isolated nodes behind the real ones and self-loops on the last of
them -- the EGC layers keep padded rows to themselves, but anything that reduces over ALL rows sees them.  For
BatchNorm's batch statistics ``FusedEGCBlock(...)(x, edge_index, n_valid=t)`` takes the number of real rows as a device
scalar ``t``: statistics, running statistics and gradients are then nn.BatchNorm1d's on the real rows, and ONE
recording serves every batch up to the padded size (tests/test_hipgraph_gpu.py).  A readout must likewise use its
``batch`` vector, with the padding in a spare graph)::

    x, edge_index = static input buffers
    def step():
        out = model(x, edge_index)          # COO in, CSR built inside the recorded step
        loss_fn(out).backward()
    graphed = egc_amd.GraphedStep(step, params=model.parameters())
    for batch in loader:
        x.copy_(batch.x); edge_index.copy_(batch.edge_index)
        graphed()                            # .grad of every parameter now holds this batch's gradient
        optimizer.step()

The reference has no counterpart (PyTorch eager throughout, experiments/*/main.py); this is the CDNA-side answer to its
launch-bound small-batch regime (task statement: "capture launch-bound inner loops in hipGraphs").
"""
from __future__ import annotations

from typing import Callable, Iterable, Optional

import torch

from .graph import recording_scope


class GraphedStep:
    """``fn`` (no arguments, no return value; reads and writes fixed tensors) recorded into a hipGraph.

    ``params``: parameters whose ``.grad`` the step produces.  Their gradients are cleared before the recording, so
    the recorded backward WRITES each ``.grad`` (a buffer of the graph's memory pool that stays attached to the
    parameter) instead of accumulating into it: every replay leaves exactly that step's gradients, as
    ``optimizer.zero_grad(set_to_none=True)`` + ``backward()`` does in the reference's loops (zinc/main.py).
    Do not set these ``.grad`` to None between replays (``zero_grad(set_to_none=False)`` or nothing at all).

    ``warmup`` eager runs on a side stream come first (torch's recipe): one-time work -- kernel attribute calls,
    workspaces cached on graph objects, the allocator's pools -- must not land in the recording.  They are real calls
    of ``fn``: a step that also runs the optimizer (``torch.optim.Adam(..., capturable=True)`` followed by
    ``zero_grad(set_to_none=False)`` -- the whole iteration is then one replay, parameters bit-identical to the eager
    loop's, tests/test_hipgraph_gpu.py) has trained ``warmup + 1`` iterations when the constructor returns."""

    def __init__(self, fn: Callable[[], None], params: Optional[Iterable[torch.nn.Parameter]] = None, warmup: int = 3,
                 pool=None):
        if not torch.cuda.is_available():
            raise RuntimeError("egc_amd.GraphedStep needs a GPU (hipGraph capture)")
        self._fn = fn
        self._params = [p for p in (params or []) if p.requires_grad]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                self._clear_grads()
                fn()
        torch.cuda.current_stream().wait_stream(side)
        from . import graph as _graph
        _graph.drop_stream_workspaces(side)     # the warm-up stream is gone after this: so is its build scratch
        self._clear_grads()
        self.graph = torch.cuda.CUDAGraph()
        with recording_scope(), torch.cuda.graph(self.graph, pool=pool):
            fn()

    def _clear_grads(self):
        for p in self._params:
            p.grad = None

    def __call__(self):
        self.graph.replay()

    def pool(self):
        """Memory pool of the recording (pass as ``pool=`` to share it with another GraphedStep)."""
        return self.graph.pool()

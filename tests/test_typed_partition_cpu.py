"""Relational EGC on a vertex partition, on the CPU: world-2 and world-4 gloo processes run the collective setup of
egc_amd.partition.build_typed_distributed and the ONE all-to-all-v of the layer over a small typed graph shaped like
ogbn-mag's (rmag/models.py:18-26); the numpy oracle of REGConv (rmag/models.py:112-148) confirms that the per-rank
pieces -- table of basis rows [owned rows of all types | halo rows], per-relation edge lists with table-row sources and
local target rows -- reproduce the rows of the single-device result (SURVEY.md 8e's correctness criterion)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egc_amd import partition as P
from egc_amd.workloads import rmag_like
from oracle import egc_oracle as orc

F_IN, F_OUT, H, B = 12, 8, 2, 2
TOL = 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _problem():
    nodes, rel = rmag_like(seed=3, scale=0.0004)       # ~780 nodes of 4 types, ~17 k entries in 7 relations
    rng = np.random.default_rng(1)
    x = {k: rng.standard_normal((n, F_IN)).astype(np.float32) for k, n in nodes.items()}
    L = F_OUT // H
    params = dict(
        bases_weight=rng.standard_normal((F_IN, B * L)).astype(np.float32),
        rel={f"{k[0]}_{k[1]}_{k[2]}": (rng.standard_normal((2 * H * B, F_IN)).astype(np.float32),
                                       rng.standard_normal(2 * H * B).astype(np.float32)) for k in rel},
        root={k: (rng.standard_normal((H * B, F_IN)).astype(np.float32), rng.standard_normal(H * B).astype(np.float32))
              for k in nodes})
    return nodes, rel, x, params


def _reference(nodes, rel, x, params):
    return orc.regconv_forward(x, {k: v.numpy() for k, v in rel.items()}, params["bases_weight"], params["rel"],
                               params["root"], H, B)


def _rank_check(part, layout, nodes, rel, x, params, table, ref):
    """What one rank holds after the exchange, against the single-device quantities.  Returns True / raises."""
    p = part.rank
    # the combined id space: every node of every type has exactly one id, ranks own contiguous ranges of it
    gid = {t: layout.combined_ids(t, torch.arange(nodes[t])) for t in layout.node_types}
    feat = np.zeros((layout.n_total, F_IN), dtype=np.float32)
    for t in layout.node_types:
        feat[gid[t].numpy()] = x[t]
    all_ids = torch.cat([gid[t] for t in layout.node_types])
    assert torch.equal(torch.sort(all_ids).values, torch.arange(layout.n_total))
    ext_ids = torch.cat([torch.arange(part.plan.lo, part.plan.hi), part.plan.halo_global_ids])
    # (1) the exchange delivered the owners' rows
    want = feat[ext_ids.numpy()] @ params["bases_weight"]
    assert np.array_equal(table.numpy(), want.astype(np.float32)), "table rows differ from the owners' bases"
    # (2) every local edge names the same (source, target) pair as the global list, in input order
    for key, e in part.rel_edges.items():
        s, _, d = key
        lo, hi = layout.owned(d, p)
        own = P.local_edges(rel[key], lo, hi)
        assert torch.equal(ext_ids[e[0]], layout.combined_ids(s, own[0])) and torch.equal(e[1] + lo, own[1])
        if e.size(1):
            assert int(e[1].max()) < hi - lo and int(e[0].max()) < part.n_table
    # (3) the oracle on the rank's own structures == the rank's rows of the single-device output
    xr = {t: x[t][layout.owned(t, p)[0]:layout.owned(t, p)[1]] for t in layout.node_types}
    xr["tbl"] = feat[ext_ids.numpy()]
    adj = {("tbl", f"{k[0]}_{k[1]}", k[2]): e.numpy() for k, e in part.rel_edges.items()}
    rel_w = {f"tbl_{k[0]}_{k[1]}_{k[2]}": params["rel"][f"{k[0]}_{k[1]}_{k[2]}"] for k in part.rel_edges}
    root_w = dict(params["root"], tbl=(np.zeros((H * B, F_IN), np.float32), np.zeros(H * B, np.float32)))
    got = orc.regconv_forward(xr, adj, params["bases_weight"], rel_w, root_w, H, B)
    for t in layout.node_types:
        lo, hi = layout.owned(t, p)
        err = np.abs(got[t] - ref[t][lo:hi]).max(initial=0.0) / max(1.0, np.abs(ref[t]).max())
        assert err <= TOL, (t, err)
    return True


def test_typed_layout_is_a_bijection_and_balances_cost():
    nodes, rel, _, _ = _problem()
    for world in (1, 2, 3, 8):
        layout = P.typed_layout(nodes, rel, world, node_types=sorted(nodes))
        assert layout.n_total == sum(nodes.values()) and len(layout.rank_bounds) == world + 1
        gid = torch.cat([layout.combined_ids(t, torch.arange(nodes[t])) for t in layout.node_types])
        assert torch.equal(torch.sort(gid).values, torch.arange(layout.n_total))
        for t in layout.node_types:      # a node's combined id lies in its owner's range
            for p in range(world):
                lo, hi = layout.owned(t, p)
                if hi > lo:
                    g = layout.combined_ids(t, torch.tensor([lo, hi - 1]))
                    assert layout.rank_bounds[p] <= int(g[0]) <= int(g[1]) < layout.rank_bounds[p + 1]
        if world > 1:                    # hubs sit at low ids (heavy-tailed targets): the cuts follow the cost, not the count
            entries = [sum(int(((ei[1] >= layout.owned(k[2], p)[0]) & (ei[1] < layout.owned(k[2], p)[1])).sum())
                           for k, ei in rel.items()) for p in range(world)]
            assert max(entries) <= 1.6 * sum(entries) / world


@pytest.mark.parametrize("world", [2, 5])
def test_typed_local_simulation_reproduces_single_device_rows(world):
    nodes, rel, x, params = _problem()
    ref = _reference(nodes, rel, x, params)
    layout = P.typed_layout(nodes, rel, world, node_types=sorted(nodes))
    parts = P.build_typed_local_simulation(rel, layout)
    tables = []
    for part in parts:
        t = torch.zeros((part.n_table, params["bases_weight"].shape[1]))
        for nt in layout.node_types:
            lo, hi = part.table_rows(nt)
            a, b = layout.owned(nt, part.rank)
            t[lo:hi] = torch.from_numpy(x[nt][a:b] @ params["bases_weight"])
        tables.append(t)
    P.simulate_exchange(tables, [p.plan for p in parts])
    for part, t in zip(parts, tables):
        assert _rank_check(part, layout, nodes, rel, x, params, t, ref)
    assert sum(sum(int(e.size(1)) for e in p.rel_edges.values()) for p in parts) == sum(int(v.size(1)) for v in rel.values())


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nodes, rel, x, params = _problem()
        ref = _reference(nodes, rel, x, params)
        layout = P.typed_layout(nodes, rel, world, node_types=sorted(nodes))    # same on every rank, no communication
        part = P.build_typed_distributed(rel, layout)                            # collective setup (two all-to-alls)
        table = torch.zeros((part.n_table, params["bases_weight"].shape[1]))
        for nt in layout.node_types:
            lo, hi = part.table_rows(nt)
            a, b = layout.owned(nt, rank)
            table[lo:hi] = torch.from_numpy(x[nt][a:b] @ params["bases_weight"])
        handle = part.plan.exchange_start(table)          # THE collective of the layer: one for all node types
        part.plan.exchange_finish(handle)
        ok = _rank_check(part, layout, nodes, rel, x, params, table, ref)
        ok = ok and part.plan.send_splits[rank] == 0 and sum(part.plan.send_splits) == part.plan.send_idx.numel()
        ok = ok and part.plan.predicted_exchange_ms(table.size(1) * 4) >= 0.0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4])
def test_typed_partition_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {r: True for r in range(world)}

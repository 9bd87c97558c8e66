"""Multi-GPU path on CPU: world_size-2 gloo processes exercise the collective setup and the halo
all-to-all-v of egc_amd.partition, and the oracle confirms the partition semantics:
concatenated per-rank results == single-device result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from egc_amd import partition as P
from oracle import egc_oracle as orc


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _graph(n=97, e=900, seed=3):
    g = torch.Generator().manual_seed(seed)
    ei = torch.randint(0, n, (2, e), generator=g)
    ei[1, :60] = 5          # a hub owned by rank 0 with sources everywhere
    ei[0, 60:90] = n - 2    # a source owned by the last rank feeding many destinations
    return ei, n


def test_vertex_ranges_cover_everything():
    for n, w in [(10, 3), (169343, 8), (5, 8), (0, 2)]:
        b = P.vertex_ranges(n, w)
        assert b[0] == 0 and b[-1] == n and len(b) == w + 1
        assert all(0 <= b[i + 1] - b[i] <= n // w + 1 for i in range(w))


def test_local_simulation_matches_global_tables():
    ei, n = _graph()
    world = 3
    parts = P.build_local_simulation(ei, n, world)
    table = torch.arange(n * 4, dtype=torch.float32).view(n, 4)
    tables = []
    for ei_l, plan in parts:
        t = torch.zeros(plan.n_local + plan.n_halo, 4)
        t[:plan.n_local] = table[plan.lo:plan.hi]
        tables.append(t)
        assert sum(plan.recv_splits) == plan.n_halo
        assert int(ei_l[1].max()) < plan.n_local and int(ei_l[0].max()) < plan.n_local + plan.n_halo
    P.simulate_exchange(tables, [p for _, p in parts])
    for (ei_l, plan), t in zip(parts, tables):
        ext_ids = torch.cat([torch.arange(plan.lo, plan.hi), plan.halo_global_ids])
        assert torch.equal(t, table[ext_ids])
        # every local edge points at the same global source as before
        kept = P.local_edges(ei, plan.lo, plan.hi)
        assert torch.equal(ext_ids[ei_l[0]], kept[0]) and torch.equal(ei_l[1] + plan.lo, kept[1])


def _oracle_partitioned(x, ei, n, world, params, H, B, aggrs):
    """EGConv forward assembled from per-rank pieces exactly as the distributed run does it: basis rows
    for [owned | halo] vertices, aggregation over the rank's CSR rows, deg^-1/2 taken from the owners."""
    full_bases = (x @ params["bases_weight"]).astype(np.float32)
    ei_full, w_full = orc.egconv_edge_set(ei.numpy(), n, aggrs, True)
    deg = np.bincount(ei_full[1], minlength=n).astype(np.float32)
    outs = []
    for ei_l, plan in P.build_local_simulation(ei, n, world):
        ext_ids = np.concatenate([np.arange(plan.lo, plan.hi), plan.halo_global_ids.numpy()])
        # the rank's edge set: its owned in-edges (self loops replaced by one per owned node)
        e = ei_l.numpy()
        keep = e[0] != e[1]
        loops = np.arange(plan.n_local)
        e = np.concatenate([e[:, keep], np.stack([loops, loops])], axis=1)
        sw = (deg[ext_ids[e[0]]] ** -0.5 * deg[plan.lo + e[1]] ** -0.5).astype(np.float32)
        agg, _ = orc.egconv_aggregate(full_bases[ext_ids][e[0]], e[1], plan.n_local, aggrs, sw)
        wts = (x[plan.lo:plan.hi] @ params["comb_w"].T + params["comb_b"]).astype(np.float32)
        w3 = wts.reshape(plan.n_local, H, B * len(aggrs))
        a3 = agg.reshape(plan.n_local, len(aggrs) * B, -1)
        outs.append(np.matmul(w3, a3).reshape(plan.n_local, -1) + params["bias"])
    return np.concatenate(outs)


def test_partitioned_semantics_equal_single_device_in_the_oracle():
    ei, n = _graph()
    rng = np.random.default_rng(0)
    fin, fout, H, B = 16, 16, 4, 2
    aggrs = ["sum", "mean", "max", "symnorm"]
    x = rng.standard_normal((n, fin)).astype(np.float32)
    params = dict(bases_weight=rng.standard_normal((fin, B * fout // H)).astype(np.float32),
                  comb_w=rng.standard_normal((H * B * len(aggrs), fin)).astype(np.float32),
                  comb_b=rng.standard_normal(H * B * len(aggrs)).astype(np.float32),
                  bias=rng.standard_normal(fout).astype(np.float32))
    ref = orc.egconv_forward(x, ei.numpy(), params["bases_weight"], params["comb_w"], params["comb_b"], params["bias"],
                             H, B, aggrs)
    for world in (2, 3, 8):
        got = _oracle_partitioned(x, ei, n, world, params, H, B, aggrs)
        assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ei, n = _graph()
        bounds = P.vertex_ranges(n, world)
        lo, hi = bounds[rank], bounds[rank + 1]
        owned = P.local_edges(ei, lo, hi)
        ei_l, plan = P.build_distributed(owned, n, interior_first=(rank % 2 == 1))  # collective setup (two all-to-alls)
        table = torch.arange(n * 6, dtype=torch.float32).view(n, 6)
        own = torch.arange(lo, hi) if plan.order is None else torch.arange(lo, hi)[plan.order]  # renumbered ranks
        ext = torch.zeros(plan.n_local + plan.n_halo, 6)
        ext[:plan.n_local] = table[own]
        handle = plan.exchange_start(ext)                    # THE forward-path collective, split form
        plan.exchange_finish(handle)
        vec = torch.zeros(plan.n_local + plan.n_halo)
        vec[:plan.n_local] = own.float()
        plan.exchange(vec)                                   # 1-D tables (deg^-1/2) go the same way
        ext_ids = torch.cat([own, plan.halo_global_ids])
        ok = torch.equal(ext, table[ext_ids]) and torch.equal(vec, ext_ids.float())
        ok = ok and torch.equal(ext_ids[ei_l[0]], owned[0]) and torch.equal(own[ei_l[1]], owned[1])
        if plan.order is not None:   # interior rows: no halo source
            interior_rows = ei_l[1] < plan.n_interior
            ok = ok and bool((ei_l[0][interior_rows] < plan.n_local).all())
        ok = ok and sum(plan.send_splits) == plan.send_idx.numel() and plan.send_splits[rank] == 0
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_halo_exchange_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_cost_balanced_bounds():
    cost = torch.tensor([10.0, 1, 1, 1, 1, 1, 1, 1, 1, 2])
    b = P.cost_balanced_bounds(cost, 2)
    assert b[0] == 0 and b[-1] == 10 and 0 < b[1] < 10
    assert b == sorted(b)
    assert P.cost_balanced_bounds(torch.zeros(0), 3) == [0, 0, 0, 0]
    # one vertex heavier than a whole share: boundaries stay monotone
    b = P.cost_balanced_bounds(torch.tensor([100.0, 1, 1, 1]), 4)
    assert b == sorted(b) and b[-1] == 4


@pytest.mark.parametrize("world", [2, 8])
def test_locality_partition_is_a_valid_balanced_renumbering(world):
    from egc_amd.workloads import heavy_tailed_graph
    n = 6000
    ei = heavy_tailed_graph(n, 45000, seed=5, communities=32, p_in=0.8)
    order, new_of_old, bounds = P.locality_partition(ei, n, world)
    assert torch.equal(torch.sort(order).values, torch.arange(n))
    assert torch.equal(new_of_old[order], torch.arange(n))
    assert bounds[0] == 0 and bounds[-1] == n and len(bounds) == world + 1 and bounds == sorted(bounds)
    # deterministic: every rank derives the same partition without communicating
    o2, n2, b2 = P.locality_partition(ei.clone(), n, world)
    assert torch.equal(order, o2) and b2 == bounds
    q_naive = P.partition_quality(ei, P.vertex_ranges(n, world))
    q = P.partition_quality(new_of_old[ei], bounds)
    assert sum(q["entries_per_rank"]) == ei.size(1)
    mean = ei.size(1) / world
    assert max(q["entries_per_rank"]) <= 1.15 * mean < max(q_naive["entries_per_rank"])   # hubs no longer on one rank
    assert q["cross_edge_frac"] < 0.6 * q_naive["cross_edge_frac"]                        # planted communities found
    assert sum(q["halo_rows_per_rank"]) < sum(q_naive["halo_rows_per_rank"])


def test_partition_with_explicit_bounds_reproduces_tables():
    """build_local_simulation with the renumbered graph + cost-balanced bounds: same invariants as the equal split."""
    from egc_amd.workloads import heavy_tailed_graph
    n, world = 3000, 4
    ei = heavy_tailed_graph(n, 20000, seed=6)
    order, new_of_old, bounds = P.locality_partition(ei, n, world)
    ei2 = new_of_old[ei]
    parts = P.build_local_simulation(ei2, n, world, bounds=bounds)
    table = torch.arange(n * 3, dtype=torch.float32).view(n, 3)
    tables = []
    for ei_l, plan in parts:
        assert (plan.lo, plan.hi) == (bounds[plan.rank], bounds[plan.rank + 1])
        t = torch.zeros(plan.n_local + plan.n_halo, 3)
        t[:plan.n_local] = table[plan.lo:plan.hi]
        tables.append(t)
    P.simulate_exchange(tables, [p for _, p in parts])
    for (ei_l, plan), t in zip(parts, tables):
        ext_ids = torch.cat([torch.arange(plan.lo, plan.hi), plan.halo_global_ids])
        assert torch.equal(t, table[ext_ids])
        kept = P.local_edges(ei2, plan.lo, plan.hi)
        assert torch.equal(ext_ids[ei_l[0]], kept[0]) and torch.equal(ei_l[1] + plan.lo, kept[1])


def _worker_bounds(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egc_amd.workloads import heavy_tailed_graph
        n = 2000
        ei = heavy_tailed_graph(n, 14000, seed=8, communities=16, p_in=0.7)
        order, new_of_old, bounds = P.locality_partition(ei, n, world)      # same on every rank, no communication
        ei2 = new_of_old[ei]
        lo, hi = bounds[rank], bounds[rank + 1]
        owned = P.local_edges(ei2, lo, hi)
        ei_l, plan = P.build_distributed(owned, n, interior_first=True, bounds=bounds)
        table = torch.arange(n * 5, dtype=torch.float32).view(n, 5)
        own = torch.arange(lo, hi)[plan.order]
        ext = torch.zeros(plan.n_local + plan.n_halo, 5)
        ext[:plan.n_local] = table[own]
        plan.exchange(ext)
        ext_ids = torch.cat([own, plan.halo_global_ids])
        ok = torch.equal(ext, table[ext_ids]) and torch.equal(ext_ids[ei_l[0]], owned[0])
        ok = ok and (plan.lo, plan.hi) == (lo, hi) and plan.n_local == hi - lo
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_renumbered_partition_gloo_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_bounds, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _worker_agree(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egc_amd.workloads import heavy_tailed_graph
        n = 500
        ei = heavy_tailed_graph(n, 3000, seed=5, communities=8, p_in=0.7)
        order, new_of_old, bounds = P.locality_partition(ei, n, world)
        ref = (order.clone(), new_of_old.clone(), list(bounds))
        if rank == 1:     # a rank that (for whatever reason) derived something else must end up with rank 0's
            new_of_old = new_of_old.flip(0).contiguous()
            order = torch.empty_like(new_of_old)
            order[new_of_old] = torch.arange(n)
            bounds = [0, n // 3, n]
        o2, n2, b2 = P.agree_on_partition(order, new_of_old, bounds)
        ret[rank] = bool(torch.equal(o2, ref[0]) and torch.equal(n2, ref[1]) and b2 == ref[2])
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_ranks_agree_on_rank0s_partition_gloo_world2():
    """bench.py's multi-GPU setup: whatever a rank computed, the renumbering in use is rank 0's."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_agree, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}
    # without a process group it is the identity
    o, m, b = P.agree_on_partition(torch.arange(4), torch.arange(4), [0, 2, 4])
    assert b == [0, 2, 4] and torch.equal(o, torch.arange(4))

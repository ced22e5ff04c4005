"""N > 1 path on CPU: world_size-2 gloo, baseline sharding with fan-out /
fan-in of slabs and no data-path collective.  The per-rank flagger is a CPU
stand-in (the oracle) -- what is under test is the sharding plumbing."""
import os
import socket

import numpy as np
import pytest


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import oracle
    from tricolour_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (5, 2, 24, 40)
        kw = dict(num_major_iterations=1)
        rs = np.random.RandomState(0)
        vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
        vis[..., 11] *= 7
        flags = rs.uniform(size=shape) < 0.02

        def cpu_flagger(v, f, **k):
            out = oracle.sum_threshold_flagger(v.numpy(), f.numpy(), **k)
            return torch.from_numpy(out)

        v = torch.from_numpy(vis) if rank == 0 else None
        f = torch.from_numpy(flags) if rank == 0 else None
        full = D.sharded_sum_threshold_flagger(v, f, shape, cpu_flagger, src=0, **kw)
        if rank == 0:
            exp = oracle.sum_threshold_flagger(vis, flags, **kw)
            q.put(("ok", bool(np.array_equal(full.numpy(), exp)), D.shard_bounds(5, world)))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_sharded_flagger_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    tag, equal, bounds = q.get(timeout=10)
    assert tag == "ok" and equal
    assert bounds == [0, 2, 5]


def _stream_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from tricolour_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = (7, 2, 24, 40)
        rs = np.random.RandomState(3)
        vis = (rs.standard_normal(shape) + 1j * rs.standard_normal(shape)).astype(np.complex64)
        flags = rs.uniform(size=shape) < 0.1
        per_bl = 2 * 24 * 40 * 9
        budget = 2 * 2 * per_bl + 100            # two rounds in flight x two peers x ONE baseline: the set is 7 baselines
        calls = []

        def reader(b0, b1):                      # the root reads slab pieces on demand: it never holds the set on its device
            calls.append((b0, b1))
            return vis[b0:b1], flags[b0:b1]

        stats = {}
        v, f = D.scatter_windows_streamed(reader if rank == 0 else None, shape, budget, src=0, stats=stats)
        sl = D.shard_slice(shape[0], world, rank)
        ok = bool(np.array_equal(v.numpy(), vis[sl]) and np.array_equal(f.numpy(), flags[sl]))
        full = D.gather_flags(f, shape, dst=0)
        if rank == 0:
            ok = ok and bool(np.array_equal(full.numpy(), flags))
            q.put(("root", ok, stats, calls, budget))
        else:
            q.put(("peer", ok, stats, None, budget))
    finally:
        dist.destroy_process_group()


def test_streamed_scatter_of_a_set_larger_than_the_root_budget():
    """BASELINE configs[2] (304 GB) cannot sit on one 288 GB GPU, so the root streams slab pieces from
    host memory within a stated budget.  Here: 7 baselines, a budget of one baseline per peer and round (two rounds in
    flight) -> 4 rounds, the staging memory alive on the root never exceeds the budget, every rank ends up with its slab."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stream_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for tag, ok, stats, calls, budget in got:
        assert ok, tag
        assert stats["rounds"] == 4 and stats["baselines_per_round_and_peer"] == 1
        if tag == "root":
            assert 0 < stats["peak_root_bytes"] <= budget
            assert len(calls) == 7 and all(b1 - b0 == 1 for b0, b1 in calls)       # piecewise reads, nothing read twice


def test_stream_plan_for_the_full_meerkat_set():
    from tricolour_amd.distributed import stream_plan
    # configs[2]: 2016 bl x 4 corr x 1024 x 4096 over 8 ranks, 32 GiB of staging memory on the root
    q, rounds = stream_plan((2016, 4, 1024, 4096), 8, 32 << 30)
    per_bl = 4 * 1024 * 4096 * 9
    assert q == (32 << 30) // 2 // 8 // per_bl == 14 and rounds == 18
    assert 2 * 8 * q * per_bl <= 32 << 30


def test_shard_bounds_cover_everything():
    from tricolour_amd.distributed import shard_bounds, shard_slice
    for nbl in (0, 1, 7, 2016, 130000):
        for world in (1, 2, 3, 8):
            b = shard_bounds(nbl, world)
            assert b[0] == 0 and b[-1] == nbl and all(x <= y for x, y in zip(b, b[1:]))
            assert sum(shard_slice(nbl, world, r).stop - shard_slice(nbl, world, r).start
                       for r in range(world)) == nbl
    assert shard_bounds(2016, 8) == [252 * i for i in range(9)]

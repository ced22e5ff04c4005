"""Multi-GPU driver: baselines shard embarrassingly (every (bl, corr) window
is independent through sum_threshold_flagger, flagging.py:765-774), one
process per GPU.  No collective sits on the data path; RCCL (torch.distributed
backend "nccl" on ROCm) is only used to fan baseline slabs out from a root
and to fan the uint8 flag slabs back in -- point-to-point transfers, so each
peer's slab travels over its own xGMI link.  With host-resident inputs prefer
feeding every rank its slab directly (H2D) and skip the scatter.
"""
import numpy as np


def shard_bounds(nbl, world_size):
    """Contiguous baseline slabs: rank g owns [b[g], b[g+1])."""
    return [(nbl * g) // world_size for g in range(world_size + 1)]


def shard_slice(nbl, world_size, rank):
    b = shard_bounds(nbl, world_size)
    return slice(b[rank], b[rank + 1])


def scatter_windows(vis, flags, shape, src=0, group=None, device=None):
    """Root `src` holds (bl, corr, time, chan) ``vis`` / ``flags`` tensors;
    every rank returns its baseline slab.  `shape` is the full 4-D shape
    (known on all ranks).  Uses batched point-to-point sends (one per peer)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nbl = shape[0]
    b = shard_bounds(nbl, world)
    mine = (b[rank + 1] - b[rank],) + tuple(shape[1:])
    if rank == src:
        dev = vis.device
        ops = []
        vr = torch.view_as_real(vis) if vis.is_complex() else vis
        f8 = flags.view(torch.uint8) if flags.dtype == torch.bool else flags
        for peer in range(world):
            if peer == src or b[peer + 1] == b[peer]:
                continue
            ops.append(dist.P2POp(dist.isend, vr[b[peer]:b[peer + 1]].contiguous(), peer, group))
            ops.append(dist.P2POp(dist.isend, f8[b[peer]:b[peer + 1]].contiguous(), peer, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for r in reqs:
            r.wait()
        return vis[b[rank]:b[rank + 1]], flags[b[rank]:b[rank + 1]]
    dev = device if device is not None else torch.device("cpu")
    vr = torch.empty(mine + (2,), dtype=torch.float32, device=dev)
    f8 = torch.empty(mine, dtype=torch.uint8, device=dev)
    if mine[0] > 0:
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.irecv, vr, src, group),
                                       dist.P2POp(dist.irecv, f8, src, group)])
        for r in reqs:
            r.wait()
    return torch.view_as_complex(vr), f8.view(torch.bool)


def gather_flags(out_local, shape, dst=0, group=None):
    """Inverse fan-in of the per-rank output flag slabs to `dst`; returns the
    full (bl, corr, time, chan) bool tensor on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = shard_bounds(shape[0], world)
    o8 = out_local.view(torch.uint8) if out_local.dtype == torch.bool else out_local
    if rank == dst:
        full = torch.empty(tuple(shape), dtype=torch.uint8, device=out_local.device)
        full[b[rank]:b[rank + 1]] = o8
        ops, bufs = [], []
        for peer in range(world):
            if peer == dst or b[peer + 1] == b[peer]:
                continue
            buf = torch.empty((b[peer + 1] - b[peer],) + tuple(shape[1:]), dtype=torch.uint8,
                              device=out_local.device)
            bufs.append((peer, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer, group))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        for r in reqs:
            r.wait()
        for peer, buf in bufs:
            full[b[peer]:b[peer + 1]] = buf
        return full.view(torch.bool)
    if o8.shape[0] > 0:
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, o8.contiguous(), dst, group)])
        for r in reqs:
            r.wait()
    return None


def sharded_sum_threshold_flagger(vis, flags, shape, flagger, src=0, group=None, device=None, **kw):
    """scatter -> per-rank ``flagger(vis_slab, flag_slab, **kw)`` -> gather.
    `flagger` is ``tricolour_amd.sum_threshold_flagger`` on GPUs (the tests
    pass a CPU stand-in to exercise the N > 1 plumbing over gloo)."""
    v, f = scatter_windows(vis, flags, shape, src=src, group=group, device=device)
    out = flagger(v, f, **kw) if v.shape[0] > 0 else f.new_zeros(f.shape)
    return gather_flags(out, shape, dst=src, group=group)

"""Multi-GPU sharding of independent LWE bootstraps (one process per GPU, torch.distributed).

Every bootstrap is independent given the read-only keys, so the path shards by
contiguous slices of the batch with keys replicated per GPU.  There is no collective inside the
computation; RCCL (backend "nccl" on ROCm) is used only for the trivial scatter / gather of
ciphertext batches.  The same code runs over gloo on CPU tensors, which is how it is tested
without GPUs (tests/test_sharding_gloo.py).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def shard_range(batch: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous slice [start, stop) of rank `rank`; the first batch % world ranks get one more."""
    base, extra = divmod(batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def scatter_rows(full, root: int = 0, *, batch: Optional[int] = None, width: Optional[int] = None,
                 like=None):
    """Root holds `full` [batch][width]; every rank returns its contiguous slice.  Non-root ranks
    pass full=None and give batch/width/like (dtype+device template)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    if rank == root:
        batch, width, like = full.shape[0], full.shape[1], full
    start, stop = shard_range(batch, world, rank)
    mine = torch.empty((stop - start, width), dtype=like.dtype, device=like.device)
    # one grouped point-to-point exchange (dist.batch_isend_irecv = one ncclGroupStart/End on RCCL): the root's sends to
    # its peers are posted together and travel over the peers' xGMI links at the same time, instead of one isend after
    # the other on one communicator
    ops = []
    if rank == root:
        for r in range(world):
            s, e = shard_range(batch, world, r)
            if r == root:
                mine.copy_(full[s:e])
            elif e > s:
                ops.append(dist.P2POp(dist.isend, full[s:e].contiguous(), r))
    elif stop > start:
        ops.append(dist.P2POp(dist.irecv, mine, root))
    if ops:
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return mine


def gather_rows(mine, batch: int, root: int = 0):
    """Inverse of scatter_rows: root returns [batch][width], the others None."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    ops, full = [], None
    if rank != root:
        if mine.shape[0]:
            ops.append(dist.P2POp(dist.isend, mine.contiguous(), root))
    else:
        full = torch.empty((batch, mine.shape[1]), dtype=mine.dtype, device=mine.device)
        for r in range(world):
            s, e = shard_range(batch, world, r)
            if r == root:
                full[s:e].copy_(mine)
            elif e > s:
                ops.append(dist.P2POp(dist.irecv, full[s:e], r))  # a contiguous row slice: received in place
    if ops:  # grouped like the scatter: the peers' sends arrive over their own links at the same time
        for q in dist.batch_isend_irecv(ops):
            q.wait()
    return full


def replicate_keys(tensors, shapes, root: int = 0, like=None):
    """Key upload for N ranks: the root holds the read-only tensors (BSK, KSK, test
    vector ...), every other rank passes None and receives a copy -- one broadcast per tensor, at key
    load time only.  `shapes` lists the tensor shapes for the ranks that have nothing yet; `like`
    gives their dtype / device.  Returns the list of tensors on every rank."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    out = []
    for i, shape in enumerate(shapes):
        if rank == root:
            t = tensors[i].contiguous()
            assert tuple(t.shape) == tuple(shape)
        else:
            t = torch.empty(tuple(shape), dtype=like.dtype, device=like.device)
        dist.broadcast(t, src=root)
        out.append(t)
    return out


def bootstrap_sharded(bootstrap_fn: Callable, full_lwe, tv, *, root: int = 0, batch: Optional[int] = None,
                      width: Optional[int] = None, like=None):
    """scatter -> local bootstrap_fn(lwe_shard, tv) on every rank -> gather on root."""
    import torch.distributed as dist
    if dist.get_rank() == root:
        batch = full_lwe.shape[0]
    mine = scatter_rows(full_lwe, root, batch=batch, width=width, like=like)
    out = bootstrap_fn(mine, tv) if mine.shape[0] else mine
    return gather_rows(out, batch, root)

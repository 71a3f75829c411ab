"""world_size-2 (and 3) gloo tests of the multi-GPU sharding path on CPU tensors: slices are
contiguous and cover the batch, scatter -> per-rank work -> gather reproduces a single-rank run,
including ragged batches and batches smaller than the world."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_common import ROOT, pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_bootstrap(lwe, tv):
    # stands in for ctx.bootstrap on CPU: any per-row function shows the plumbing is row-exact
    return (lwe * 3 + tv[: lwe.shape[1]].unsqueeze(0)).to(lwe.dtype)


def _worker(rank, world, port, batch, width, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    pkg()
    sh = importlib.import_module("tfhe_research_amd.sharding")
    g = torch.Generator().manual_seed(5)
    full = torch.randint(-(1 << 31), (1 << 31) - 1, (batch, width), dtype=torch.int32, generator=g)
    tv = torch.arange(width, dtype=torch.int32)
    like = torch.empty(0, dtype=torch.int32)
    # key upload: the root's read-only tensors reach every rank unchanged (one broadcast each)
    shapes = [(3, 2, 2, 8), (5, width)]
    keys = [torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, generator=g) for s in shapes]
    got = sh.replicate_keys(keys if rank == 0 else None, shapes, root=0, like=like)
    keys_ok = all(torch.equal(a, b) for a, b in zip(got, keys))  # same generator seed on every rank
    out = sh.bootstrap_sharded(_fake_bootstrap, full if rank == 0 else None, tv, root=0,
                               batch=batch, width=width, like=like)
    ok = torch.tensor([int(keys_ok)])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(bool(torch.equal(out, _fake_bootstrap(full, tv))) and bool(ok.item()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,batch", [(2, 10), (2, 7), (3, 8), (2, 1)])
def test_scatter_compute_gather(world, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, batch, 6, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def _worker_cfg4(rank, world, port, q):
    """BASELINE configs[3]: 2^20 ciphertexts over 8 ranks.  Rows are 4 words wide here (the real 631
    would move 2.6 GB through loopback gloo); what is checked is the partition: every rank gets
    exactly 2^17 contiguous rows, row r of the result is f(row r of the input), keys reach all 8."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import importlib
    pkg()
    sh = importlib.import_module("tfhe_research_amd.sharding")
    batch, width = 1 << 20, 4
    like = torch.empty(0, dtype=torch.int32)
    full = None
    if rank == 0:
        full = (torch.arange(batch * width, dtype=torch.int64) * 2654435761 % (1 << 31)).to(torch.int32).reshape(batch, width)
    tv = torch.arange(width, dtype=torch.int32)
    shapes = [(6, 3, 2, 16), (40, width)]
    g = torch.Generator().manual_seed(8)
    keys = [torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, generator=g) for s in shapes]
    got = sh.replicate_keys(keys if rank == 0 else None, shapes, root=0, like=like)
    seen = {}

    def fn(shard, tvv):
        seen["rows"] = shard.shape[0]
        return _fake_bootstrap(shard, tvv)
    out = sh.bootstrap_sharded(fn, full, tv, root=0, batch=batch, width=width, like=like)
    ok = torch.tensor([int(seen.get("rows") == 1 << 17 and sh.shard_range(batch, world, rank) == (rank << 17, (rank + 1) << 17)
                           and all(torch.equal(a, b) for a, b in zip(got, keys)))])
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if rank == 0:
        q.put(bool(ok.item()) and tuple(out.shape) == (batch, width) and bool(torch.equal(out, _fake_bootstrap(full, tv))))
    dist.barrier()
    dist.destroy_process_group()


def test_cfg4_partition_two_to_the_twenty_over_eight_ranks():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_cfg4, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_shard_ranges_cover_the_batch():
    import importlib
    pkg()
    sh = importlib.import_module("tfhe_research_amd.sharding")
    for batch in (0, 1, 7, 8, 4096, 1 << 20, 1000003):
        for world in (1, 2, 3, 4, 8):
            edges = [sh.shard_range(batch, world, r) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == batch
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [e - s for s, e in edges]
            assert max(sizes) - min(sizes) <= 1

"""Child of tests/test_gpu_pool.py::test_sharded_bootstrap_over_rccl_two_ranks (launched under torch.distributed.run
with one rank per GPU, backend nccl = RCCL): the one-process-per-GPU path of SURVEY 8(e) on real devices --
replicate_keys (broadcast), scatter_rows / gather_rows (grouped isend/irecv over xGMI) around a local bootstrap.
Rank 0 checks the gathered rows against a single-context run and the golden fixture's outputs."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
import __graft_entry__ as entry  # noqa: E402
import golden_common as gc  # noqa: E402


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    m = entry.load_package()
    sharding = importlib.import_module("tfhe_research_amd.sharding")
    pd, a = gc.load_set("n1024_full_word")
    params = m.TfheParams(pd["k"], pd["log_n"], pd["n"], m.DecomposerParams(*pd["pbs"]), m.DecomposerParams(*pd["ks"]),
                          log_p=pd["log_p"], padding_bits=pd["padding_bits"])
    to_d = lambda x: torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).to(dev)
    like = torch.empty(0, dtype=torch.int32, device=dev)
    # the keys exist on rank 0 only and reach the other rank by broadcast
    held = [to_d(a["bsk"]), to_d(a["ksk"]), to_d(a["tv"])] if rank == 0 else None
    bsk, ksk, tv = sharding.replicate_keys(held, [a["bsk"].shape, a["ksk"].shape, a["tv"].shape], root=0, like=like)
    # 8 golden rows repeated to a ragged 1,003-row batch: slices of 502 + 501
    rows = 1003
    idx = np.arange(rows) % a["lwe_in"].shape[0]
    full = to_d(a["lwe_in"][idx]) if rank == 0 else None
    with m.Context(params, device=local) as ctx:
        ctx.use_torch_stream()
        ctx.load_bootstrapping_key(bsk, ksk)
        out = sharding.bootstrap_sharded(lambda shard, tvv: ctx.bootstrap(shard, tvv), full, tv, root=0, batch=rows,
                                         width=params.n + 1, like=like)
        ok = True
        if rank == 0:
            got = out.cpu().numpy().view(np.uint32)
            ok = bool(np.array_equal(got, a["lwe_out"][idx]))
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    names = [None] * world
    dist.all_gather_object(names, f"rank {rank}: cuda:{local} {torch.cuda.get_device_name(local)}")
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(names)
        print("rccl two-rank OK" if flag.item() else "rccl two-rank MISMATCH", flush=True)
    sys.exit(0 if flag.item() else 1)


if __name__ == "__main__":
    main()

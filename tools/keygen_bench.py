"""Times tfhe_bootstrapping_key_gen_device (GPU keygen) per BASELINE config next to the
oracle's O(n R k N^2) CPU keygen (timed on 2 GGSWs, extrapolated to n).  Run on the GPU box:
    python tools/keygen_bench.py > gpurun_out/keygen.txt
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from gpu_common import to_pkg_params  # noqa: E402
from oracle import oracle  # noqa: E402

m = entry.load_package()
dev = torch.device("cuda:0")
print("# config  n  bsk_MB  ksk_MB  gpu_keygen_ms(incl. NTT-domain install)  gpu_keygen_only_ms  cpu_oracle_s(1 thread, extrapolated)")
for name in ("cfg2", "cfg3", "cfg5"):
    p = oracle.CONFIGS[name]
    rng = np.random.default_rng(1)
    lwe_sk = rng.integers(0, 2, size=p.n).astype(np.uint32)
    glwe_sk = rng.integers(0, 2, size=(p.k, p.N)).astype(np.uint32)
    bsk = torch.randint(-2**31, 2**31 - 1, p.bsk_shape(), dtype=torch.int32, device=dev)
    ksk = torch.randint(-2**31, 2**31 - 1, p.ksk_shape(), dtype=torch.int32, device=dev)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.use_torch_stream()
        res = {}
        for load in (True, False):
            ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, bsk, ksk, load=load)  # warm-up
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, bsk, ksk, load=load)
            torch.cuda.synchronize()
            res[load] = (time.perf_counter() - t0) / 3 * 1e3
    samples = rng.integers(0, 1 << 32, size=(2, p.R, p.k + 1, p.N), dtype=np.uint64).astype(np.uint32)
    t0 = time.perf_counter()
    oracle.encrypt_ggsw_from_samples(p, glwe_sk, lwe_sk[:2], samples)
    cpu = (time.perf_counter() - t0) / 2 * p.n
    print(f"{name} {p.n} {bsk.numel() * 4 / 1e6:.1f} {ksk.numel() * 4 / 1e6:.1f} {res[True]:.2f} {res[False]:.2f} {cpu:.1f}")

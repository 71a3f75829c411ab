"""Dev helper (GPU box): blind-rotation kernel time of every backend that admits a parameter set, to check AUTO's choice
(capi.cpp) at shapes outside BASELINE.  usage: auto_choice_bench.py [batch]   (random key and ciphertext words)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
SHAPES = [  # k, logN, n, (logB, levels)
    (2, 11, 630, (2, 5)), (1, 11, 630, (8, 3)), (2, 11, 630, (8, 4)), (2, 9, 722, (3, 6)), (2, 9, 722, (4, 6)), (2, 9, 630, (8, 2)),
    (1, 9, 630, (4, 6)), (1, 9, 500, (8, 2)), (1, 10, 630, (2, 10)), (1, 10, 630, (7, 3)), (2, 10, 630, (8, 2)),
]
BACKENDS = [("auto", m.BACKEND_AUTO), ("fp64-fft", m.BACKEND_FP64_FFT), ("fp64-p49", m.BACKEND_FP64_P49), ("fp64-p42", m.BACKEND_FP64)]
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
for k, logn, n, pbs in SHAPES:
    P = m.TfheParams(k, logn, n, m.DecomposerParams(*pbs))
    b = batch if logn < 11 else batch // 4
    lw, bk, kk = rw(b, n + 1), rw(*P.bsk_shape()), rw(*P.ksk_shape())
    tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
    out = torch.empty_like(lw)
    line = [f"N={1 << logn} k={k} l={pbs[1]} logB={pbs[0]} rows={(k + 1) * pbs[1]} batch={b}:"]
    for name, be in BACKENDS:
        try:
            ctx = m.Context(P, backend=be)
        except m.TfheError as e:
            line.append(f"{name} -")
            continue
        ctx.use_torch_stream(); ctx.load_bootstrapping_key(bk, kk); ctx.reserve(b); ctx.set_timing(True)
        ctx.bootstrap(lw, tvd, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            ctx.bootstrap(lw, tvd, out=out); ts.append(ctx.last_kernel_ms()[0])
        line.append(f"{name}{'=' + ctx.backend if name == 'auto' else ''} {np.mean(ts):.2f} ms")
        ctx.close()
    print("  ".join(line), flush=True)

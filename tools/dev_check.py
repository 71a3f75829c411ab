"""Dev helper (GPU box): parity of a dev build (TFHE_HIP_LIB=...) on a small cfg2-shaped case for both
backends, then kernel timings at the BASELINE cfg2 workload."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
from oracle import oracle as orc
m = g.load_package()
orc.set_poly_mul_mode(1)
p = orc.Params(1, 10, 5, orc.Decomposer(7, 3))
pp = m.TfheParams(1, 10, 5, m.DecomposerParams(7, 3))
lwe, bsk, ksk, tv = orc.synthetic_inputs(p, 9, cfg_index=3)
want = np.stack([orc.bootstrap(p, lwe[b], bsk, ksk, tv) for b in range(9)])
for name, be in (("goldilocks", m.BACKEND_GOLDILOCKS), ("fp64", m.BACKEND_FP64)):
    with m.Context(pp, backend=be) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        got = ctx.bootstrap(lwe, tv)
        print(name, ctx.backend, "parity:", bool(np.array_equal(got, want)), flush=True)
import torch
P = m.TfheParams(1, 10, 630, m.DecomposerParams(7, 3))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
batch = int(os.environ.get("BATCH", "4096"))
lw, bk, kk = rw(batch, 631), rw(*P.bsk_shape()), rw(*P.ksk_shape())
tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
for name, be in (("fp64", m.BACKEND_FP64), ("goldilocks", m.BACKEND_GOLDILOCKS)):
    if os.environ.get("ONLY") and os.environ["ONLY"] != name: continue
    ctx = m.Context(P, backend=be); ctx.use_torch_stream(); ctx.load_bootstrapping_key(bk, kk); ctx.reserve(batch); ctx.set_timing(True)
    out = torch.empty_like(lw)
    ctx.bootstrap(lw, tvd, out=out); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        ctx.bootstrap(lw, tvd, out=out); ts.append(ctx.last_kernel_ms())
    br = np.mean([t[0] for t in ts]); ks = np.mean([t[1] for t in ts])
    print(f"{name}: blind_rotate {br:.2f} ms  key_switch {ks:.2f} ms  -> {batch / ((br + ks) * 1e-3):.0f} PBS/s", flush=True)
    ctx.close()

"""Dev helper (GPU box): kernel timings of a dev build for one workload shape. usage: dev_bench.py cfg2|cfg3|cfg5
(DEV_BACKEND=BACKEND_AUTO|BACKEND_FP64|BACKEND_FP64_P49|... picks the field, default BACKEND_FP64)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()
from oracle import oracle as orc
orc.set_poly_mul_mode(1)
which = sys.argv[1]
BACKEND = getattr(m, os.environ.get('DEV_BACKEND', 'BACKEND_FP64'))
cfg = {"cfg1": (1, 9, 500, (8, 2), 4096), "cfg2": (1, 10, 630, (7, 3), 4096), "cfg3": (2, 9, 722, (4, 6), 4096), "cfg5": (2, 11, 630, (8, 4), 1024),
       # shapes outside BASELINE (the other instantiations of the kernels): N = 1024 with k = 2, N = 2048 with k = 1
       "k2n1024": (2, 10, 600, (4, 7), 2048), "k1n2048": (1, 11, 600, (8, 3), 2048)}[which]
k, logn, n, pbs, batch = cfg
batch = int(os.environ.get("DEV_BATCH", batch))
# parity on a short key first
ps = orc.Params(k, logn, 4, orc.Decomposer(*pbs)); pp = m.TfheParams(k, logn, 4, m.DecomposerParams(*pbs))
lwe, bsk, ksk, tv = orc.synthetic_inputs(ps, 5, cfg_index=3)
want = np.stack([orc.bootstrap(ps, lwe[b], bsk, ksk, tv) for b in range(5)])
with m.Context(pp, backend=BACKEND) as ctx:
    ctx.load_bootstrapping_key(bsk, ksk); ok = bool(np.array_equal(ctx.bootstrap(lwe, tv), want))
P = m.TfheParams(k, logn, n, m.DecomposerParams(*pbs))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
lw, bk, kk = rw(batch, n + 1), rw(*P.bsk_shape()), rw(*P.ksk_shape())
tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
ctx = m.Context(P, backend=BACKEND); ctx.use_torch_stream(); ctx.load_bootstrapping_key(bk, kk); ctx.reserve(batch); ctx.set_timing(True)
ctx.set_kernel_shape({"auto": m.SHAPE_AUTO, "wide": m.SHAPE_WIDE, "team": m.SHAPE_TEAM}[os.environ.get("DEV_SHAPE", "auto")])
if os.environ.get("DEV_ALIGNED"): ctx.set_decomposer_alignment(True)
out = torch.empty_like(lw)
ctx.bootstrap(lw, tvd, out=out); torch.cuda.synchronize()
ts = []
for _ in range(3):
    ctx.bootstrap(lw, tvd, out=out); ts.append(ctx.last_kernel_ms())
br = np.mean([t[0] for t in ts]); ks = np.mean([t[1] for t in ts])
print(f"{os.path.basename(os.environ.get('TFHE_HIP_LIB','default'))} {which} shape={os.environ.get('DEV_SHAPE', 'auto')} batch={batch}: parity {ok} blind_rotate {br:.2f} ms key_switch {ks:.2f} ms -> {batch / ((br + ks) * 1e-3):.0f} PBS/s", flush=True)
# key switch parity on random data (oracle)
rng = np.random.default_rng(1)
pk = orc.Params(k, logn, 37, orc.Decomposer(*pbs), orc.Decomposer(4, 5))
ksk_s = rng.integers(0, 1 << 32, size=pk.ksk_shape(), dtype=np.uint64).astype(np.uint32)
big = rng.integers(0, 1 << 32, size=(45, pk.big_n + 1), dtype=np.uint64).astype(np.uint32)
bsk_s = np.zeros(pk.bsk_shape(), dtype=np.uint32)
with m.Context(m.TfheParams(k, logn, 37, m.DecomposerParams(*pbs))) as c2:
    c2.load_bootstrapping_key(bsk_s, ksk_s)
    got = c2.key_switch(big)
okk = all(np.array_equal(got[b], orc.key_switch_lwe(big[b], pk.big_n, pk.n, pk.ks, ksk_s)) for b in (0, 7, 44))
print("key_switch parity", okk)

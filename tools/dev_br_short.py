"""Dev helper (GPU box): blind rotation with a short LWE key (n products per team) for comparison with
the standalone external-product kernel at the same number of products per resident team."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()
n, batch = int(sys.argv[1]), int(sys.argv[2])
P = m.TfheParams(1, 10, n, m.DecomposerParams(7, 3))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
lw, bk, kk = rw(batch, n + 1), rw(*P.bsk_shape()), rw(*P.ksk_shape())
tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
with m.Context(P, backend=m.BACKEND_FP64) as ctx:
    ctx.load_bootstrapping_key(bk, kk)
    out = ctx.blind_rotate(lw, tvd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ctx.blind_rotate(lw, tvd, out=out)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f"blind_rotate n={n} batch={batch}: {us:.1f} us per launch, {us / n / max(1, batch // 1024):.2f} us per product per resident team")

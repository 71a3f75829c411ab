"""GPU box: A/B of the unrolled (BMMP) blind rotation against the reference's loop on the same shape.
usage: python tools/bmmp_bench.py [cfg3|cfg1] [batch] -- prints kernel times (HIP events on the
kernel's stream, mean of 5 launches after warm-up) for every field that is exact for the shape."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()
which = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
k, logn, n, pbs = {"cfg3": (2, 9, 722, (4, 6)), "cfg1": (1, 9, 500, (8, 2))}[which]
P = m.TfheParams(k, logn, n, m.DecomposerParams(*pbs))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
lw, kk = rw(batch, n + 1), rw(*P.ksk_shape())
keys = {"loop": rw(*P.bsk_shape()), "bmmp": rw(*P.bsk_bmmp_shape())}
tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
# the backends that offer BMMP; BMMP_BENCH_BACKENDS=fp64-fft adds the complex transform for dev builds with -DTFHE_BMMP_FFT=1
BACKENDS = [("fp64-p49", m.BACKEND_FP64_P49), ("goldilocks", m.BACKEND_GOLDILOCKS)]
if "fp64-fft" in os.environ.get("BMMP_BENCH_BACKENDS", ""):
    BACKENDS = [("fp64-fft", m.BACKEND_FP64_FFT)]
for name, be in BACKENDS:
    try:
        ctx = m.Context(P, backend=be)
    except m.TfheError:
        continue
    ctx.set_kernel_shape(m.SHAPE_TEAM)
    ctx.reserve(batch); ctx.set_timing(True)
    out = torch.empty_like(lw)
    line = [f"{which} batch {batch} {ctx.backend:<12}"]
    for mode in ("loop", "bmmp"):
        (ctx.load_bootstrapping_key_bmmp if mode == "bmmp" else ctx.load_bootstrapping_key)(keys[mode], kk)
        ctx.bootstrap(lw, tvd, out=out); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            ctx.bootstrap(lw, tvd, out=out); ts.append(ctx.last_kernel_ms()[0])
        line.append(f"{mode}: blind rotation {np.mean(ts):8.2f} ms = {batch / (np.mean(ts) * 1e-3):9.0f} /s")
    print("  ".join(line), flush=True)
    ctx.close()

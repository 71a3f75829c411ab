"""Dev helper (GPU box): the persistent external-product kernel at batch > resident teams against the
same products computed in chunks small enough to get one workgroup per sample, plus oracle spot checks."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()
from oracle import oracle as orc
orc.set_poly_mul_mode(1)
k, logn, pbs = 1, 10, (7, 3)
p = orc.Params(k, logn, 4, orc.Decomposer(*pbs)); P = m.TfheParams(k, logn, 4, m.DecomposerParams(*pbs))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(3)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
batch = int(os.environ.get("BATCH", "4096"))
with m.Context(P, backend=m.BACKEND_FP64) as ctx:
    for per_sample in (False, True):
        count = batch if per_sample else 1
        ggsw = rw(count, p.R, k + 1, p.N)
        prep = ctx.prepare_ggsw_device(ggsw)
        glwe = rw(batch, k + 1, p.N)
        full = ctx.external_product_prepared(prep, glwe)
        parts = [ctx.external_product_prepared(prep[i:i + 256] if per_sample else prep, glwe[i:i + 256].contiguous())
                 for i in range(0, batch, 256)]
        torch.cuda.synchronize()
        same = torch.equal(full, torch.cat(parts))
        gg, gl, fu = ggsw.cpu().numpy().view(np.uint32), glwe.cpu().numpy().view(np.uint32), full.cpu().numpy().view(np.uint32)
        ok = all(np.array_equal(fu[b], orc.external_product(p, gg[b if per_sample else 0], gl[b])) for b in (0, 1, batch // 2 + 3, batch - 1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ctx.external_product_prepared(prep, glwe, out=full)
        e1.record(); e1.synchronize()
        print(f"per_sample_ggsw={per_sample} batch={batch}: persistent == chunked {same}, oracle rows {ok}, {e0.elapsed_time(e1) / 20 * 1e3:.1f} us per launch", flush=True)

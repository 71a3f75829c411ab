import os, sys
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import __graft_entry__ as g
m = g.load_package()
from oracle import oracle as orc
orc.build(); orc.set_poly_mul_mode(1)
p = orc.Params(1, 10, 5, orc.Decomposer(8, 4))
lwe, bsk, ksk, tv = orc.synthetic_inputs(p, 7, cfg_index=9)
tvs = np.stack([np.roll(tv, 5 * b) for b in range(7)])
pp = m.TfheParams(1, 10, 5, m.DecomposerParams(8, 4))
with m.Context(pp, backend=m.BACKEND_FP64_FFT) as ctx:
    ctx.load_bootstrapping_key(bsk, ksk)
    out = ctx.bootstrap(lwe, tvs)
    acc = ctx.blind_rotate(lwe, tvs)
ok = True
for b in range(7):
    want, tr = orc.bootstrap(p, lwe[b], bsk, ksk, tvs[b], trace=True)
    ok &= bool(np.array_equal(out[b], want)) and bool(np.array_equal(acc[b], tr["acc_final"]))
print("split kernel, key-dependent rotation, 7 samples with per-sample test vectors: parity", ok)

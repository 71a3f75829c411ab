"""Child process of tests/test_gpu_parity.py::test_short_launches_with_two_samples_per_team: the blind rotation of a batch
goes out in launches of TFHE_BR_CHUNK samples (kernels.hip::blind_rotate_plan reads the variables once per process, hence
the child); with an ODD chunk every launch of a two-samples-per-team kernel ends in a team that is one sample short."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from gpu_common import pkg, to_pkg_params  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()
orc.set_poly_mul_mode(1)
m = pkg()
ok = True
for k, logn, n, pbs, log_p in ((2, 9, 4, (4, 6), 2), (2, 11, 2, (8, 4), 4), (1, 10, 3, (8, 4), 2)):
    p = orc.Params(k, logn, n, orc.Decomposer(*pbs), log_p=log_p)
    lwe, bsk, ksk, tv = orc.synthetic_inputs(p, 8, cfg_index=30 + logn)
    tvs = np.stack([np.roll(tv, 7 * b) for b in range(8)])
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.set_kernel_shape(m.SHAPE_TEAM)   # the launch plan under test is the team kernel's (a batch of 8 would go wide)
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tvs)
        acc = ctx.blind_rotate(lwe, tvs)
    for b in range(8):
        want, tr = orc.bootstrap(p, lwe[b], bsk, ksk, tvs[b], trace=True)
        ok &= bool(np.array_equal(out[b], want)) and bool(np.array_equal(acc[b], tr["acc_final"]))
print("short launches parity", ok, "chunk", os.environ.get("TFHE_BR_CHUNK"))
sys.exit(0 if ok else 1)

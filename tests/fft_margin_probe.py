"""Child process of tests/test_gpu_fft_margin.py: loads the PROBE build of the HIP library (libtfhe_hip_probe.so,
-DTFHE_FFT_TRACK_ERROR: FftField::to_u32 records |t - rint(t)| on gfx950), runs external products and blind rotations
of the fp64-fft backend on operands at the magnitude bound (constant-sign and random-sign), on random operands and on
a full-length key-dependent blind rotation, checks the outputs against the oracle and prints one JSON line per shape:
{"shape", "bound", "margin_external_product", "margin_blind_rotation", "exact"}.  Test infrastructure only."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
os.environ["TFHE_HIP_LIB"] = os.path.join(ROOT, "tfhe-research_amd", "libtfhe_hip_probe.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from gpu_common import pkg, rand_u32, to_pkg_params  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from test_emu_kernels import FFT_EDGE_SHAPES, extreme_operands, fft_error_bound  # noqa: E402

# name, k, logN, (logB, levels), n of the blind rotation, aligned decomposer
SHAPES = [("cfg1", 1, 9, (8, 2), 500, False), ("cfg2", 1, 10, (7, 3), 630, True), ("cfg3", 2, 9, (4, 6), 64, False),
          ("cfg5", 2, 11, (8, 4), 16, False)]
# (log2 B does not divide 32 at the edge shapes: in the reference's literal mode a trivially encrypted accumulator then
# decomposes to zeros -- SURVEY D4 -- so their blind rotations run with the aligned decomposer, like cfg2's)
SHAPES += [(f"edge_N{1 << logn}_k{k}_B{pbs[0]}_l{pbs[1]}", k, logn, pbs, 16, 32 % pbs[0] != 0) for k, logn, pbs, _ in FFT_EDGE_SHAPES]


def main():
    m = pkg()
    orc.build()
    orc.set_poly_mul_mode(1)
    for name, k, logn, pbs, n, aligned in SHAPES:
        p = orc.Params(k, logn, n, orc.Decomposer(*pbs))
        rng = np.random.default_rng(1000 + logn + k)
        bound = fft_error_bound(logn, p.R, pbs[0])
        exact = True
        with m.Context(to_pkg_params(p), backend=m.BACKEND_FP64_FFT) as ctx:
            ctx.fft_margin(reset=True)
            cases = extreme_operands(orc, p, pbs, rng, False)
            for _ in range(26):
                cases += extreme_operands(orc, p, pbs, rng, True)
            ggsw = np.stack([c[0] for c in cases])
            glwe = np.stack([c[1] for c in cases])
            got = ctx.external_product(ggsw, glwe)
            for b in list(range(8)) + [len(cases) - 1]:
                exact &= bool(np.array_equal(got[b], orc.external_product(p, ggsw[b], glwe[b])))
            # random operands too (typical magnitudes)
            rg, rl = rand_u32(rng, (4,) + ggsw.shape[1:]), rand_u32(rng, (4,) + glwe.shape[1:])
            got = ctx.external_product(rg, rl)
            exact &= bool(np.array_equal(got[0], orc.external_product(p, rg[0], rl[0])))
            margin_ep = ctx.fft_margin(reset=True)
            # blind rotation: a key of extreme words only, in the mode where the rotation depends on the key
            if aligned:
                ctx.set_decomposer_alignment(True)
            key_words = np.array([0x7FFF7FFF, 0x80008000, 0x7FFF8000, 0x80007FFF], dtype=np.uint32)
            bsk = rng.choice(key_words, size=p.bsk_shape())
            ksk = rand_u32(rng, p.ksk_shape())
            batch = 64
            lwe = rand_u32(rng, (batch, n + 1))
            tvs = rng.integers(0, 1 << p.log_p, size=(batch, p.N)).astype(np.uint32)
            ctx.load_bootstrapping_key(bsk, ksk)
            acc = ctx.blind_rotate(lwe, tvs)
            with orc.decomposer_aligned(aligned):
                for b in range(2 if n > 100 else 4):
                    _, tr = orc.bootstrap(p, lwe[b], bsk, ksk, tvs[b], trace=True)
                    exact &= bool(np.array_equal(acc[b], tr["acc_final"]))
            margin_br = ctx.fft_margin(reset=True)
        print(json.dumps({"shape": name, "bound": bound, "margin_external_product": margin_ep,
                          "margin_blind_rotation": margin_br, "exact": exact}), flush=True)


if __name__ == "__main__":
    main()

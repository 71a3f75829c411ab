"""GPU (-m gpu): the rounding margin of the fp64-fft backend measured ON gfx950.

The complex-FFT backend returns the exact integer convolution because every value it lifts is within 1/2 of an
integer; csrc/field_fft.h proves a bound on that distance and the context admits the backend below 1/4.  The emulator
measures the distance on x86 (tests/test_emu_kernels.py::test_fft_rounding_margin); this test measures it on the
hardware whose v_fma_f64 / v_mul_f64 / v_add_f64 the proof is about: a probe build of the same sources
(libtfhe_hip_probe.so, -DTFHE_FFT_TRACK_ERROR) keeps an atomic maximum of |t - rint(t)| in FftField::to_u32.  The probe
library is loaded in a child process (tests/fft_margin_probe.py) so that this process keeps the product library."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "tfhe-research_amd", "libtfhe_hip_probe.so")


def test_rounding_margin_on_gfx950():
    assert os.path.exists(PROBE), "probe library missing: __graft_entry__.build() builds it"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fft_margin_probe.py")], env=env,
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-2000:]
    rows = [json.loads(line) for line in run.stdout.splitlines() if line.startswith("{")]
    names = {r["shape"] for r in rows}
    assert {"cfg1", "cfg2", "cfg3", "cfg5", "edge_N1024_k2_B11_l2", "edge_N512_k1_B13_l2"} <= names, names
    for r in rows:
        assert r["exact"], r
        assert r["bound"] < 0.25, r
        # something was recorded, and it sits orders of magnitude inside the proven bound
        assert 0.0 < r["margin_external_product"] < r["bound"] / 100, r
        assert 0.0 < r["margin_blind_rotation"] < r["bound"] / 100, r
        print(f"gfx950 rounding margin {r['shape']}: external product {r['margin_external_product']:.3g}, "
              f"blind rotation {r['margin_blind_rotation']:.3g}, proven bound {r['bound']:.3g}")
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "fft_margin_gfx950.json"), "w") as f:
            json.dump(rows, f, indent=1)


def test_product_library_has_no_probe(oracle):
    """the shipped library must not pay for the instrumentation"""
    from gpu_common import pkg, to_pkg_params
    m = pkg()
    p = oracle.Params(1, 10, 4, oracle.Decomposer(7, 3))
    with m.Context(to_pkg_params(p), backend=m.BACKEND_FP64_FFT) as ctx:
        with pytest.raises(m.TfheError) as e:
            ctx.fft_margin()
        assert e.value.status == m.TFHE_ERR_UNSUPPORTED

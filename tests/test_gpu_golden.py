"""GPU (-m gpu): the HIP path, through the C ABI, against the COMMITTED golden fixtures (tests/golden/,
frozen by tools/make_golden.py) -- not against a fresh oracle run.  An oracle regression and a kernel
regression in the same direction can therefore not hide each other, and the same files replay
through the reference crate (rust/reference_patch/golden_replay.rs)."""
import numpy as np
import pytest

import golden_common as gc
from gpu_common import pkg

pytestmark = pytest.mark.gpu

SETS = ["ref_test", "misaligned", "n1024_full_word"]
BACKENDS = ["auto", "fp64", "fp64-p49", "fp64-fft", "goldilocks", "goldilocks-split"]


def pkg_params(p):
    m = pkg()
    return m.TfheParams(p["k"], p["log_n"], p["n"], m.DecomposerParams(*p["pbs"]), m.DecomposerParams(*p["ks"]),
                        log_p=p["log_p"], padding_bits=p["padding_bits"])


def backend_id(name):
    m = pkg()
    return {"fp64": m.BACKEND_FP64, "goldilocks": m.BACKEND_GOLDILOCKS, "auto": m.BACKEND_AUTO,
            "goldilocks-split": m.BACKEND_GOLDILOCKS_SPLIT, "fp64-p49": m.BACKEND_FP64_P49,
            "fp64-fft": m.BACKEND_FP64_FFT}[name]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", SETS)
def test_hip_path_reproduces_the_golden_trace(name, backend):
    """every stage of bootstrapping.rs:58-120 against the frozen trace: mod switch, blind rotation
    (acc_final), each CMUX on its own (acc_after_each), sample extract, key switch, whole bootstrap"""
    m = pkg()
    pd, a = gc.load_set(name)
    try:
        ctx = m.Context(pkg_params(pd), backend=backend_id(backend))
    except m.TfheError as e:
        if e.status == m.TFHE_ERR_EXACTNESS:
            pytest.skip("set outside this field's exactness bound")
        if e.status == m.TFHE_ERR_UNSUPPORTED and backend == "fp64-fft":
            pytest.skip("the complex-FFT backend has kernels at N = 512 and 1024")
        raise
    with ctx:
        ctx.load_bootstrapping_key(a["bsk"], a["ksk"])
        rows, n, N = a["lwe_in"].shape[0], pd["n"], 1 << pd["log_n"]
        assert np.array_equal(ctx.bootstrap(a["lwe_in"], a["tv"]), a["lwe_out"])
        assert np.array_equal(ctx.blind_rotate(a["lwe_in"], a["tv"]), a["acc_final"])
        assert np.array_equal(ctx.switch_modulus(a["lwe_in"], 32, pd["log_n"] + 1).reshape(rows, n + 1), a["approximate_lwe"])
        assert np.array_equal(ctx.sample_extract(a["acc_final"], 0), a["extracted_lwe"])
        assert np.array_equal(ctx.key_switch(a["extracted_lwe"]), a["lwe_out"])
        # acc_init = X^{-b~} * (0, .., 0, tv << shift): glwe.rs:20-34, 141-151, 232-243
        trivial = np.zeros((rows, pd["k"] + 1, N), dtype=np.uint32)
        trivial[:, pd["k"]] = a["tv"].astype(np.uint32) << np.uint32(32 - pd["log_p"] - pd["padding_bits"])
        minus_b = -a["approximate_lwe"][:, n].astype(np.int64)
        assert np.array_equal(ctx.glwe_mul_monomial(trivial, minus_b), a["acc_init"])
        # CMUX i alone (ggsw.rs:164-178): all rows of iteration i share GGSW i, like the blind rotation
        acc = a["acc_init"]
        for i in range(n):
            rotated = ctx.glwe_mul_monomial(acc, a["approximate_lwe"][:, i].astype(np.int64))
            acc, clobbered = ctx.cmux(a["bsk"][i], acc, rotated)
            assert np.array_equal(acc, a["acc_after_each"][:, i]), (name, backend, i)
            assert np.array_equal(clobbered, (rotated - a["acc_after_each"][:, i - 1] if i else rotated - a["acc_init"]).astype(np.uint32))


def test_golden_rows_decrypt_on_the_gpu():
    """rows 0-3 of ref_test hold 0..3 under the committed secret key: GPU decryption of the GPU's own
    bootstrap output gives the messages back (bootstrapping_works, bootstrapping.rs:194-230)"""
    m = pkg()
    pd, a = gc.load_set("ref_test")
    with m.Context(pkg_params(pd)) as ctx:
        ctx.load_bootstrapping_key(a["bsk"], a["ksk"])
        out = ctx.bootstrap(a["lwe_in"][:4], a["tv"])
        assert np.array_equal(ctx.decrypt_bits(a["lwe_sk"], out), np.arange(4, dtype=np.uint32))


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg5"])
def test_full_size_rows_hash_to_the_golden_digests(oracle, cfg):
    """BASELINE cfg1/2/3/5 at the full batch of 4096: the 8 recorded rows of the GPU's bootstrap,
    blind rotation and sample extract hash to the committed SHA-256 digests (the oracle is only used
    here to regenerate the SplitMix64 inputs, whose digests are checked too)"""
    m = pkg()
    d = gc.full_size_digests()
    c = d["configs"][cfg]
    p = gc.oracle_params(oracle, c["params"])
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, d["batch"], cfg_index=c["cfg_index"], lut=c["lut"])
    assert gc.sha_row(bsk) == c["bsk"] and gc.sha_row(ksk) == c["ksk"] and gc.sha_row(tv) == c["tv"]
    rows = [r["row"] for r in c["rows"]]
    with m.Context(pkg_params(c["params"])) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tv)                      # the whole 4096 batch
        acc = ctx.blind_rotate(lwe[rows], tv)
        ext = ctx.sample_extract(acc, 0)
    for j, r in enumerate(c["rows"]):
        assert gc.sha_row(lwe[r["row"]]) == r["lwe_in"]
        assert gc.sha_row(out[r["row"]]) == r["lwe_out"], (cfg, r["row"])
        assert gc.sha_row(acc[j]) == r["acc_final"], (cfg, r["row"])
        assert gc.sha_row(ext[j]) == r["extracted_lwe"], (cfg, r["row"])

"""CPU: the committed golden fixtures (tests/golden/, tools/make_golden.py) against today's oracle.

The fixtures freeze the oracle's outputs at the time they were written, so an oracle regression
fails here, and the -m gpu twin (test_gpu_golden.py) compares the HIP path with the SAME files
rather than with a fresh oracle run.  The reference crate replays the same files through
rust/reference_patch/golden_replay.rs."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import golden_common as gc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SETS = ["ref_test", "misaligned", "n1024_full_word"]


def test_fixture_files_are_well_formed():
    """magic, dims, payload size, FNV-1a checksum of the format and the manifest's SHA-256 -- read by a
    numpy reader that shares no code with the library that wrote the files"""
    m = gc.manifest()
    assert sorted(m["sets"]) == sorted(SETS)
    for name in SETS:
        p, arrays = gc.load_set(name)
        for key in m["sets"][name]["files"]:
            kind, params, flags, arr = gc.read_tfhe_file(os.path.join(gc.GOLDEN, name, key + ".tfhe"), verify_checksum=True)
            assert flags == 0  # the reference's literal decomposer
            assert params == (p["k"], p["log_n"], p["n"], p["padding_bits"], p["log_p"], 32,
                              p["ks"][0], p["ks"][1], 32, p["pbs"][0], p["pbs"][1], 32)
        rows, n, N, k = m["sets"][name]["rows"], p["n"], 1 << p["log_n"], p["k"]
        assert arrays["lwe_in"].shape == (rows, n + 1) and arrays["lwe_out"].shape == (rows, n + 1)
        assert arrays["bsk"].shape == (n, (k + 1) * p["pbs"][1], k + 1, N)
        assert arrays["acc_after_each"].shape == (rows, n, k + 1, N)
        assert arrays["extracted_lwe"].shape == (rows, k * N + 1)


@pytest.mark.parametrize("mode", [0, 1], ids=["toeplitz", "schoolbook"])
@pytest.mark.parametrize("name", SETS)
def test_oracle_reproduces_every_trace(oracle, name, mode):
    """bootstrapping.rs:58-120 step by step: mod switch, X^-b~ * tv, every CMUX, sample extract, key
    switch -- with the literal Toeplitz product (utils.rs:155-160) and its schoolbook twin (:221-236)"""
    pd, a = gc.load_set(name)
    p = gc.oracle_params(oracle, pd)
    oracle.set_poly_mul_mode(mode)
    try:
        for b in range(a["lwe_in"].shape[0]):
            out, tr = oracle.bootstrap(p, a["lwe_in"][b], a["bsk"], a["ksk"], a["tv"], trace=True, trace_each=True)
            assert np.array_equal(out, a["lwe_out"][b]), (name, b)
            for key in ("approximate_lwe", "acc_init", "acc_after_each", "acc_final", "extracted_lwe"):
                assert np.array_equal(tr[key], a[key][b]), (name, b, key)
    finally:
        oracle.set_poly_mul_mode(1)
    # the pieces on their own: the same fixtures pin switch_modulus, cmux, sample_extract, key_switch_lwe
    b = 3
    assert np.array_equal(oracle.switch_modulus(a["lwe_in"][b], 32, p.glwe_poly_degree + 1), a["approximate_lwe"][b])
    acc = a["acc_init"][b]
    for i in range(p.n):
        rotated = oracle.glwe_mul_monomial(acc, int(a["approximate_lwe"][b][i]))
        acc, _ = oracle.cmux(p, a["bsk"][i], acc, rotated)
        assert np.array_equal(acc, a["acc_after_each"][b][i]), (name, i)
    assert np.array_equal(oracle.sample_extract(p, a["acc_final"][b], 0), a["extracted_lwe"][b])
    assert np.array_equal(oracle.key_switch_lwe(a["extracted_lwe"][b], p.big_n, p.n, p.ks, a["ksk"]), a["lwe_out"][b])


def test_ref_test_rows_decrypt(oracle):
    """rows 0-3 of ref_test encrypt 0..3 under real keys with the identity LUT: the frozen outputs
    decrypt to the same messages (what bootstrapping_works asserts, bootstrapping.rs:194-230)"""
    pd, a = gc.load_set("ref_test")
    p = gc.oracle_params(oracle, pd)
    for msg in range(4):
        assert oracle.decrypt_lwe_message(p, a["lwe_sk"], a["lwe_in"][msg]) == msg
        assert oracle.decrypt_lwe_message(p, a["lwe_sk"], a["lwe_out"][msg]) == msg


@pytest.mark.parametrize("name", SETS)
def test_numpy_twin_reproduces_the_fixtures(name):
    """oracle/pyref.py shares no code with the C oracle: two rows each"""
    sys.path.insert(0, ROOT)
    from oracle import pyref
    pd, a = gc.load_set(name)
    for b in (0, 5):
        kw = dict(log_n=pd["log_n"], log_p=pd["log_p"], padding=pd["padding_bits"], pbs=tuple(pd["pbs"]), ks=tuple(pd["ks"]))
        assert np.array_equal(pyref.bootstrap(a["lwe_in"][b], a["bsk"], a["ksk"], a["tv"], return_acc=True, **kw),
                              a["acc_final"][b]), (name, b)
        assert np.array_equal(pyref.bootstrap(a["lwe_in"][b], a["bsk"], a["ksk"], a["tv"], **kw), a["lwe_out"][b]), (name, b)


@pytest.mark.parametrize("name", SETS)
def test_device_headers_reproduce_the_fixtures(emu, name):
    """the kernel bodies (csrc/*.h) through the host SIMT emulator, every field that is exact for the
    set: blind rotation + fused sample extract against the frozen acc_final / extracted_lwe"""
    import test_emu_kernels as tek
    pd, a = gc.load_set(name)
    k, logn, n, pbs = pd["k"], pd["log_n"], pd["n"], tuple(pd["pbs"])
    rows = 3
    for field in tek.FIELDS:
        if not tek.field_exact(field, k, logn, pbs):
            continue
        flat = np.ascontiguousarray(a["bsk"]).reshape(-1, 1 << logn)
        parts = emu.emu_field_parts(field)
        spec = np.zeros((flat.shape[0], parts, 1 << logn), dtype=np.uint64)
        emu.emu_set_key_k(k)   # the key's layout depends on (field, N, k)
        try:
            assert emu.emu_bsk_prepare(field, logn, 1, C.c_size_t(flat.shape[0]), tek.p32(flat), tek.p64(spec)) == 0
        finally:
            emu.emu_set_key_k(0)
        glwe = np.zeros((rows, k + 1, 1 << logn), dtype=np.uint32)
        ext = np.zeros((rows, k * (1 << logn) + 1), dtype=np.uint32)
        lwe = np.ascontiguousarray(a["lwe_in"][:rows])
        assert emu.emu_blind_rotate(field, 1, n, k, logn, pd["log_p"], pd["padding_bits"], pbs[0], pbs[1], C.c_size_t(rows),
                                    tek.p32(lwe), tek.p32(a["tv"]), C.c_size_t(0), tek.p64(spec), tek.p32(glwe), tek.p32(ext)) == 0
        assert np.array_equal(glwe, a["acc_final"][:rows]), (name, field)
        assert np.array_equal(ext, a["extracted_lwe"][:rows]), (name, field)


# rows of the full-size digests recomputed in the default CPU run (an oracle PBS costs 0.4 s at cfg1,
# 4 s at cfg2, 5 s at cfg3, 30+ s at cfg5); TFHE_GOLDEN_FULL=1 recomputes all 32
DEFAULT_FULL = {"cfg1": 8, "cfg2": 2, "cfg3": 1, "cfg5": 0}


@pytest.mark.parametrize("cfg", ["cfg1", "cfg2", "cfg3", "cfg5"])
def test_full_size_digests(oracle, cfg):
    """BASELINE cfg1/2/3/5 at full size: inputs regenerate from the SplitMix64 seed to the recorded
    digests, and the oracle's outputs for the checked rows hash to the frozen ones"""
    from concurrent.futures import ThreadPoolExecutor
    d = gc.full_size_digests()
    c = d["configs"][cfg]
    p = gc.oracle_params(oracle, {"k": c["params"]["k"], "log_n": c["params"]["log_n"], "n": c["params"]["n"],
                                  "pbs": c["params"]["pbs"], "ks": c["params"]["ks"], "log_p": c["params"]["log_p"],
                                  "padding_bits": c["params"]["padding_bits"]})
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, d["batch"], cfg_index=c["cfg_index"], lut=c["lut"])
    assert gc.sha_row(bsk) == c["bsk"] and gc.sha_row(ksk) == c["ksk"] and gc.sha_row(tv) == c["tv"]
    for r in c["rows"]:
        assert gc.sha_row(lwe[r["row"]]) == r["lwe_in"]
    count = len(c["rows"]) if os.environ.get("TFHE_GOLDEN_FULL") == "1" else DEFAULT_FULL[cfg]
    todo = c["rows"][:count]
    if not todo:
        pytest.skip("full-size oracle rows of this configuration only with TFHE_GOLDEN_FULL=1 (the -m gpu test checks all of them on the GPU)")

    def one(r):
        out, tr = oracle.bootstrap(p, lwe[r["row"]], bsk, ksk, tv, trace=True)
        return gc.sha_row(out) == r["lwe_out"] and gc.sha_row(tr["extracted_lwe"]) == r["extracted_lwe"] and \
            gc.sha_row(tr["acc_final"]) == r["acc_final"]
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as pool:
        assert all(pool.map(one, todo)), cfg

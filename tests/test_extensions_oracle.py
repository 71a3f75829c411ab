"""CPU: the two extensions beyond the reference in the oracle and in the device source
run through the emulator -- the aligned decomposer (bases with beta^l != q) and the KS-then-PBS
order.  The literal / PBS-then-KS behaviour stays the default everywhere."""
import ctypes as C

import numpy as np
import pytest

from test_emu_kernels import FIELDS, p32, p64, prepared  # noqa: F401


def signed(d):
    return d.astype(np.int64) - ((d.astype(np.int64) >> 31) << 32)


@pytest.mark.parametrize("log_base,levels", [(7, 3), (7, 4), (5, 6), (3, 10), (4, 6), (8, 4), (16, 2)])
def test_aligned_decomposer_recomposes(oracle, log_base, levels):
    dec = oracle.Decomposer(log_base, levels)
    rng = np.random.default_rng(log_base * 100 + levels)
    vals = np.concatenate([rng.integers(0, 1 << 32, size=2000, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xF8F8F8F8], dtype=np.uint32)])
    with oracle.decomposer_aligned(True):
        digits = oracle.decompose(dec, vals)          # [count][levels], MSB first
        shifts = [oracle.gadget_shift(dec, lv) for lv in range(levels)]
    assert shifts == [32 - log_base * (lv + 1) for lv in range(levels)]
    recomposed = np.zeros(vals.size, dtype=np.uint64)
    for lv in range(levels):
        recomposed += signed(digits[:, lv]).astype(np.uint64) << np.uint64(shifts[lv])
    assert np.array_equal((recomposed & 0xFFFFFFFF).astype(np.uint32), oracle.round_value(dec, vals))
    B = 1 << log_base
    assert signed(digits).min() >= -B // 2 and signed(digits).max() <= B
    literal = oracle.decompose(dec, vals)
    if 32 % log_base == 0:
        assert np.array_equal(literal, digits)        # same bits whenever log_base divides 32
    else:
        assert not np.array_equal(literal, digits)


def test_aligned_mode_makes_base_128_decrypt(oracle):
    """BASELINE cfg2's decomposer (log_base 7, 3 levels) at a small ring: in the reference's literal
    mode bootstrapping does not decrypt (the top 4 bits are never represented), in aligned mode every
    message survives an identity-LUT bootstrap."""
    p = oracle.Params(1, 9, 8, oracle.Decomposer(7, 3), oracle.Decomposer(7, 3))
    tv = oracle.construct_identity_test_vector(p)
    results = {}
    for aligned in (False, True):
        with oracle.decomposer_aligned(aligned):
            rng = oracle.Rng(99)
            lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
            got = []
            for msg in range(4):
                ct = oracle.encrypt_lwe(p, lwe_sk, msg, rng)
                got.append(oracle.decrypt_lwe_message(p, lwe_sk, oracle.bootstrap(p, ct, bsk, ksk, tv)))
            results[aligned] = got
    assert results[True] == [0, 1, 2, 3]
    assert results[False] != [0, 1, 2, 3]
    assert oracle.lib().orc_get_decomposer_aligned() == 0   # the default is restored


@pytest.mark.parametrize("field", FIELDS)
def test_emulated_external_product_aligned(emu, oracle, field):  # noqa: F811
    k, logn, pbs = 1, 10, (7, 3)
    params = oracle.Params(k, logn, 4, oracle.Decomposer(*pbs))
    rng = np.random.default_rng(77)
    ggsw = rng.integers(0, 1 << 32, size=(params.R, k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    glwe = rng.integers(0, 1 << 32, size=(k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    glwe[:, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
    spec = prepared(emu, field, params, ggsw, 1)
    out = np.zeros_like(glwe)
    emu.emu_set_aligned(1)
    try:
        assert emu.emu_external_product(field, 1, k, logn, pbs[0], pbs[1], p64(spec), p32(glwe), p32(out)) == 0
    finally:
        emu.emu_set_aligned(0)
    with oracle.decomposer_aligned(True):
        want = oracle.external_product(params, ggsw, glwe)
    assert np.array_equal(out, want)
    assert not np.array_equal(out, oracle.external_product(params, ggsw, glwe))


def test_ks_first_order_decrypts_and_keeps_combination_noise_low(oracle):
    """notes/TFHE.md:367-400 on the oracle: with the key switch first, a PBS output carries only the
    blind-rotation noise, so 4*c2 + 2*c1 + c0 of three bootstrapped bits stays far from the 3-bit
    decision boundary, while in the reference's order each term also carries the key-switch noise."""
    p = oracle.Params(2, 9, 16, oracle.Decomposer(4, 6), log_p=3)
    rng = oracle.Rng(5150)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    big_sk = glwe_sk.reshape(-1)
    tv = oracle.construct_identity_test_vector(p)
    shift = 32 - p.log_p - p.padding_bits
    errs = {"ks_first": [], "reference": []}

    def centred(raw, msg):
        e = (raw - (msg << shift)) & 0xFFFFFFFF
        return e - (1 << 32) if e >= (1 << 31) else e

    for msg in (0, 1, 5, 7):
        ct_big = oracle.encrypt_lwe(p, big_sk, msg, rng)
        out_big = oracle.bootstrap_ks_first(p, ct_big, bsk, ksk, tv)
        assert out_big.shape == (p.big_n + 1,)
        assert oracle.decrypt_lwe_message(p, big_sk, out_big) == msg
        errs["ks_first"].append(centred(oracle.decrypt_lwe_raw(big_sk, out_big), msg))
        ct = oracle.encrypt_lwe(p, lwe_sk, msg, rng)
        out = oracle.bootstrap(p, ct, bsk, ksk, tv)
        errs["reference"].append(centred(oracle.decrypt_lwe_raw(lwe_sk, out), msg))
    assert max(abs(e) for e in errs["ks_first"]) * 16 < max(abs(e) for e in errs["reference"]), errs

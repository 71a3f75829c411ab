"""CPU tests of the oracle itself: the reference's asserting tests restated (via the C selftest),
known answers, and agreement between the two independent restatements (C oracle vs numpy)."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyref

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(HERE), "oracle")


def test_selftest_binary(oracle):
    """decomposer.rs:103-115, utils.rs:265-305, lwe.rs:183-194, glwe.rs:275-294,
    key_switching.rs:118-159, bootstrapping.rs:194-230, boolean.rs:67-101 restated in C."""
    exe = os.path.join(ORACLE_DIR, "_build", "oracle_selftest")
    assert os.path.exists(exe)
    out = subprocess.run([exe, "--full"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert "OK" in out.stdout


def test_known_answers(oracle):
    dec46 = oracle.Decomposer(4, 6)
    assert oracle.decompose(dec46, [0xABCDEF12]).astype(np.int32).tolist() == [[-5, -4, -3, -2, -1, -1]]
    dec73 = oracle.Decomposer(7, 3)
    assert oracle.decompose(dec73, [0xABCDEF12]).tolist() == [[0xFFFFFFDE, 0x38, 0xFFFFFFE0]]
    got = oracle.switch_modulus([0, 1 << 21, (1 << 21) + 1, 1 << 22, 0xFFFFFFFF, 0xFFE00000, 0xFFDFFFFF], 32, 10)
    assert got.tolist() == [0, 1, 1, 1, 0, 0, 1023]
    prod = oracle.poly_mul([12, 4, 123, 43, 3, 2, 3], [12, 232, 5, 3, 2, 4, 2])
    assert prod.tolist() == [4294966139, 2387, 2353, 29088, 10647, 1354, 930]


@pytest.mark.parametrize("log_base,levels", [(4, 6), (4, 5), (7, 3), (8, 2), (8, 4), (16, 2), (5, 6), (3, 10)])
def test_decompose_c_vs_numpy(oracle, log_base, levels):
    rng = np.random.default_rng(log_base * 100 + levels)
    v = rng.integers(0, 1 << 32, size=5000, dtype=np.uint64).astype(np.uint32)
    v[:8] = [0, 1, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF, 0xFFFFFF80, 0x00000F80, 0xF8F8F8F8]
    a = oracle.decompose(oracle.Decomposer(log_base, levels), v)
    b = pyref.decompose(v, log_base, levels)
    assert np.array_equal(a, b)
    assert np.array_equal(oracle.round_value(oracle.Decomposer(log_base, levels), v[:64]),
                          pyref.round_value(v[:64], log_base, levels))


def test_digit_range_quirk(oracle):
    """D10: digits live in [-B/2, B], the value B appears (carry into a limb equal to B-1)."""
    rng = np.random.default_rng(0)
    v = rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32)
    for log_base, levels in [(4, 6), (7, 3), (8, 4)]:
        d = oracle.decompose(oracle.Decomposer(log_base, levels), v).astype(np.int32)
        assert d.min() == -(1 << (log_base - 1))
        assert d.max() == (1 << log_base)


def test_monomial_and_polymul_c_vs_numpy(oracle):
    rng = np.random.default_rng(1)
    for n in (4, 16, 64):
        a = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        b = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        assert np.array_equal(oracle.school_book_negacylic_mul(a, b), pyref.negacyclic_mul(a, b))
        oracle.set_poly_mul_mode(0)
        lit = oracle.poly_mul(a, b)
        oracle.set_poly_mul_mode(1)
        assert np.array_equal(lit, pyref.negacyclic_mul(a, b))
        for idx in (-3 * n, -n - 1, -1, 0, 1, n - 1, n, n + 1, 2 * n - 1, 2 * n, 5 * n + 3):
            assert np.array_equal(oracle.poly_mul_monomial(a, idx), pyref.mul_monomial(a, idx)), idx


SMALL = [
    # (k, logN, n, pbs, ks, log_p)
    (1, 4, 5, (4, 6), (4, 5), 2),
    (2, 3, 4, (7, 3), (4, 5), 2),
    (1, 5, 6, (8, 2), (8, 3), 2),
    (2, 4, 3, (8, 4), (4, 5), 4),
]


@pytest.mark.parametrize("k,log_n,n,pbs,ks,log_p", SMALL)
def test_bootstrap_c_vs_numpy(oracle, k, log_n, n, pbs, ks, log_p):
    p = oracle.Params(k, log_n, n, oracle.Decomposer(*pbs), oracle.Decomposer(*ks), log_p=log_p)
    assert oracle.validate(p) == 0
    rng = np.random.default_rng(7 + n)
    lut = rng.integers(0, 1 << log_p, size=1 << log_p)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 3, cfg_index=9, lut=lut)
    assert np.array_equal(tv, pyref.test_from_lut(lut, log_n, log_p))
    for b in range(3):
        out, tr = oracle.bootstrap(p, lwe[b], bsk, ksk, tv, trace=True)
        ref_acc = pyref.bootstrap(lwe[b], bsk, ksk, tv, log_n=log_n, log_p=log_p, padding=1,
                                  pbs=pbs, ks=ks, return_acc=True)
        assert np.array_equal(tr["acc_final"], ref_acc)
        assert np.array_equal(tr["extracted_lwe"], pyref.sample_extract0(ref_acc))
        ref_out = pyref.bootstrap(lwe[b], bsk, ksk, tv, log_n=log_n, log_p=log_p, padding=1,
                                  pbs=pbs, ks=ks)
        assert np.array_equal(out, ref_out)


def test_external_product_and_cmux_c_vs_numpy(oracle):
    p = oracle.Params(2, 4, 3, oracle.Decomposer(7, 3))
    rng = np.random.default_rng(3)
    ggsw = rng.integers(0, 1 << 32, size=(p.R, p.k + 1, p.N), dtype=np.uint64).astype(np.uint32)
    ct0 = rng.integers(0, 1 << 32, size=(p.k + 1, p.N), dtype=np.uint64).astype(np.uint32)
    ct1 = rng.integers(0, 1 << 32, size=(p.k + 1, p.N), dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(oracle.external_product(p, ggsw, ct0), pyref.external_product(ggsw, ct0, 7, 3))
    res, clobbered = oracle.cmux(p, ggsw, ct0, ct1)
    assert np.array_equal(res, pyref.cmux(ggsw, ct0, ct1, 7, 3))
    assert np.array_equal(clobbered, (ct1 - ct0).astype(np.uint32))  # ggsw.rs:171 mutates ct1


def test_decrypt_correct_bootstrap_and_gates(oracle):
    """bootstrapping_works / boolean_gates_work with the reference's cfg(test) parameters."""
    p = oracle.REF_TEST
    rng = oracle.Rng(1234)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    tv = oracle.construct_identity_test_vector(p)
    for m in range(4):
        ct = oracle.encrypt_lwe(p, lwe_sk, m, rng)
        out = oracle.bootstrap(p, ct, bsk, ksk, tv)
        assert oracle.decrypt_lwe_message(p, lwe_sk, out) == m
    for f in (lambda l, r: l & r, lambda l, r: l | r, lambda l, r: 1 - (l & r), lambda l, r: l ^ r):
        for i in range(4):
            lhs, rhs = (i >> 1) & 1, i & 1
            ct1 = oracle.encrypt_lwe(p, lwe_sk, lhs, rng)
            ct0 = oracle.encrypt_lwe(p, lwe_sk, rhs, rng)
            out = oracle.boolean_gate(p, f, ct0, ct1, bsk, ksk)
            assert oracle.decrypt_lwe_message(p, lwe_sk, out) == f(lhs, rhs)


def test_invalid_params_rejected(oracle):
    # levels > floor(32/log_base): the reference's truncation loop would never terminate
    assert oracle.validate(oracle.Params(1, 4, 4, oracle.Decomposer(7, 5))) != 0
    # log_base*levels > 32: usize underflow in round_value
    assert oracle.validate(oracle.Params(1, 4, 4, oracle.Decomposer(8, 5))) != 0
    for cfg in oracle.CONFIGS.values():
        assert oracle.validate(cfg) == 0

"""Unrolled (BMMP) blind rotation, notes/BMMP Bootstrapping.md:13-25.

CPU: the oracle twin (oracle.bootstrap_bmmp, composed from the reference's own external_product /
monomial product / sample_extract / key_switch_lwe) decrypts correctly under real keys, and the
device code (csrc/pbs_wave.h::blind_rotate_bmmp_team) run through the host SIMT emulator
reproduces it word for word in every exact field.  -m gpu: the HIP path through the C ABI against the
oracle twin word for word, decrypt-correctness with keys generated on the GPU at the reference's
default parameters, and the gates on top of it."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import pkg, to_pkg_params
from test_emu_kernels import FIELDS, field_exact, p32, p64, prepared

# k, n, (logB, levels), log_p: N = 512 shapes (the only ring degree the mode is offered for)
SHAPES = [(2, 4, (4, 6), 2), (1, 6, (8, 2), 2), (2, 2, (7, 3), 3)]


def params_of(oracle, k, n, pbs, log_p):
    return oracle.Params(k, 9, n, oracle.Decomposer(*pbs), log_p=log_p)


def test_identity_behind_the_unrolling():
    """X^{a s + a' s'} = s s' (X^{a+a'} - 1) + s (1 - s') (X^a - 1) + (1 - s) s' (X^{a'} - 1) + 1
    (notes/BMMP Bootstrapping.md:15) as an identity of exponents for all four (s, s')."""
    for s0 in (0, 1):
        for s1 in (0, 1):
            terms = [(s0 * s1, "a+a'"), (s0 * (1 - s1), "a"), ((1 - s0) * s1, "a'")]
            picked = [name for coef, name in terms if coef]
            want = {(0, 0): [], (1, 0): ["a"], (0, 1): ["a'"], (1, 1): ["a+a'"]}[(s0, s1)]
            assert picked == want


def test_oracle_twin_decrypts_with_real_keys(oracle):
    """every message of the plaintext space through bootstrap_bmmp with the identity LUT, and a
    non-identity LUT: the result decrypts to LUT[message] (what bootstrapping_works asserts for the
    plain loop, bootstrapping.rs:194-230)"""
    p = oracle.REF_TEST
    rng = oracle.Rng(0xB33F)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen_bmmp(p, rng)
    assert bsk.shape == oracle.bmmp_bsk_shape(p)
    assert oracle.bmmp_messages([1, 1, 1, 0, 0, 1, 0, 0]).tolist() == [1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0]
    for lut in ([0, 1, 2, 3], [3, 1, 0, 2]):
        tv = oracle.construct_test_from_lut(p, lut)
        for msg in range(4):
            ct = oracle.encrypt_lwe(p, lwe_sk, msg, rng)
            out = oracle.bootstrap_bmmp(p, ct, bsk, ksk, tv)
            assert oracle.decrypt_lwe_message(p, lwe_sk, out) == lut[msg], (lut, msg)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,n,pbs,log_p", SHAPES)
def test_device_headers_vs_oracle_twin(emu, oracle, field, k, n, pbs, log_p):
    """blind_rotate_bmmp_team + fused sample extract through the emulator, uniform synthetic words
    (the arithmetic is total), rows with a~ = 0 and b~ wrapping"""
    if not field_exact(field, k, 9, pbs):
        pytest.skip("outside this field's exactness bound")
    p = params_of(oracle, k, n, pbs, log_p)
    rng = np.random.default_rng(100 * k + n)
    bsk = rng.integers(0, 1 << 32, size=oracle.bmmp_bsk_shape(p), dtype=np.uint64).astype(np.uint32)
    lwe = rng.integers(0, 1 << 32, size=(3, n + 1), dtype=np.uint64).astype(np.uint32)
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    lwe[2, :] = 0x80000000
    tv = oracle.construct_test_from_lut(p, rng.integers(0, 1 << log_p, size=1 << log_p))
    spec = prepared(emu, field, p, bsk)
    acc = np.zeros((3, k + 1, p.N), dtype=np.uint32)
    ext = np.zeros((3, p.big_n + 1), dtype=np.uint32)
    assert emu.emu_blind_rotate_bmmp(field, n, k, log_p, 1, pbs[0], pbs[1], C.c_size_t(3), p32(lwe), p32(tv), C.c_size_t(0),
                                     p64(spec), p32(acc), p32(ext)) == 0
    for b in range(3):
        want = oracle.blind_rotate_bmmp(p, lwe[b], bsk, tv)
        assert np.array_equal(acc[b], want), (field, b)
        assert np.array_equal(ext[b], oracle.sample_extract(p, want, 0)), (field, b)


# ------------------------------------------------------------------------------------------- GPU
BACKENDS = ["auto", "fp64", "fp64-p49", "fp64-fft", "goldilocks", "goldilocks-split"]


def backend_id(name):
    m = pkg()
    return {"fp64": m.BACKEND_FP64, "goldilocks": m.BACKEND_GOLDILOCKS, "auto": m.BACKEND_AUTO,
            "goldilocks-split": m.BACKEND_GOLDILOCKS_SPLIT, "fp64-p49": m.BACKEND_FP64_P49,
            "fp64-fft": m.BACKEND_FP64_FFT}[name]


@pytest.mark.gpu
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("k,n,pbs,log_p", SHAPES)
def test_hip_path_vs_oracle_twin(oracle, k, n, pbs, log_p, backend):
    m = pkg()
    p = params_of(oracle, k, n, pbs, log_p)
    try:
        ctx = m.Context(to_pkg_params(p), backend=backend_id(backend))
    except m.TfheError as e:
        if e.status == m.TFHE_ERR_EXACTNESS:
            pytest.skip("outside this field's exactness bound")
        raise
    rng = np.random.default_rng(7 * k + n)
    bsk = rng.integers(0, 1 << 32, size=oracle.bmmp_bsk_shape(p), dtype=np.uint64).astype(np.uint32)
    ksk = rng.integers(0, 1 << 32, size=p.ksk_shape(), dtype=np.uint64).astype(np.uint32)
    lwe = rng.integers(0, 1 << 32, size=(9, n + 1), dtype=np.uint64).astype(np.uint32)
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    lwe[2, :] = 0x80000000
    tv = oracle.construct_test_from_lut(p, rng.integers(0, 1 << log_p, size=1 << log_p))
    with ctx:
        assert not ctx.uses_bmmp
        if ctx.backend not in ("goldilocks", "fp64-p49"):
            # offered only where the three accumulator sets fit the registers (tfhe_hip.h): refused with a reason elsewhere
            with pytest.raises(m.TfheError) as e:
                ctx.load_bootstrapping_key_bmmp(bsk, ksk)
            assert e.value.status == m.TFHE_ERR_UNSUPPORTED and "goldilocks" in str(e.value)
            return
        ctx.load_bootstrapping_key_bmmp(bsk, ksk)
        assert ctx.uses_bmmp
        acc = ctx.blind_rotate(lwe, tv)
        out = ctx.bootstrap(lwe, tv)
        tvs = np.stack([np.roll(tv, b) for b in range(9)])
        out2 = ctx.bootstrap(lwe, tvs)
        for b in range(9):
            assert np.array_equal(acc[b], oracle.blind_rotate_bmmp(p, lwe[b], bsk, tv)), (backend, b)
            assert np.array_equal(out[b], oracle.bootstrap_bmmp(p, lwe[b], bsk, ksk, tv)), (backend, b)
        for b in (0, 4, 8):
            assert np.array_equal(out2[b], oracle.bootstrap_bmmp(p, lwe[b], bsk, ksk, tvs[b]))
        # an ordinary key switches the context back to the reference's loop
        plain = rng.integers(0, 1 << 32, size=p.bsk_shape(), dtype=np.uint64).astype(np.uint32)
        ctx.load_bootstrapping_key(plain, ksk)
        assert not ctx.uses_bmmp
        assert np.array_equal(ctx.bootstrap(lwe[:2], tv)[1], oracle.bootstrap(p, lwe[1], plain, ksk, tv))


@pytest.mark.gpu
def test_bmmp_refused_where_it_is_not_offered(oracle):
    m = pkg()
    for params in (m.TfheParams(1, 10, 4, m.DecomposerParams(2, 8)), m.TfheParams(2, 9, 5, m.DecomposerParams(4, 6))):
        with m.Context(params, backend=m.BACKEND_FP64_P49) as ctx:   # N = 1024; odd n
            with pytest.raises(m.TfheError) as e:
                ctx.load_bootstrapping_key_bmmp(np.zeros(params.bsk_bmmp_shape(), dtype=np.uint32),
                                                np.zeros(params.ksk_shape(), dtype=np.uint32))
            assert e.value.status == m.TFHE_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_bmmp_with_gpu_generated_keys_at_the_reference_default_parameters(oracle):
    """lib.rs:101-123 (N=512, k=2, n=722, l=6, logB=4): BMMP key made on the GPU, NAND gates and
    identity bootstraps over a batch decrypt correctly; two outputs bit-exact against the oracle twin"""
    m = pkg()
    p = oracle.CFG3
    rng = np.random.default_rng(12)
    with m.Context(to_pkg_params(p), backend=m.BACKEND_FP64_P49) as ctx:
        lwe_sk, glwe_sk, bsk, ksk = ctx.generate_keys(rng, bmmp=True)
        assert ctx.uses_bmmp and bsk.shape == oracle.bmmp_bsk_shape(p)
        bits = rng.integers(0, 2, size=(256, 2))
        c1 = ctx.encrypt_bits(lwe_sk, bits[:, 0], rng)
        c0 = ctx.encrypt_bits(lwe_sk, bits[:, 1], rng)
        out = ctx.gate(m.GATE_NAND, c0, c1)
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, out), 1 - (bits[:, 0] & bits[:, 1]))
        msgs = rng.integers(0, 4, size=64)
        cts = ctx.encrypt_bits(lwe_sk, msgs, rng)
        tv = m.construct_identity_test_vector(to_pkg_params(p))
        ref = ctx.bootstrap(cts, tv)
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, ref), msgs)
    for b in (0, 63):
        assert np.array_equal(ref[b], oracle.bootstrap_bmmp(p, cts[b], bsk, ksk, tv)), b

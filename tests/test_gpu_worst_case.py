"""GPU (-m gpu): worst-case operand magnitudes through the C ABI on gfx950.

The exact-NTT fields are exact only below a bound on |integer convolution| (tfhe_hip.h, capi.cpp
`convolution_bits`): fp64-p49 has 0.08 bit of room at the reference's default parameters
(18 rows x 512 x 2^4 x 2^31 = 2^48.17 against p/2 = 2^48.254).  Random data never comes near those
bounds, so these tests drive the bound itself on the hardware whose v_fma_f64 / v_rndne_f64 the
argument is about: every gadget digit at +B or -B/2, every key word at the extremes of its signed
halves (0x7FFF7FFF, 0x80008000, 0x7FFF8000) or of the whole signed word (0x80000000, 0x7FFFFFFF),
and the negacyclic sign pattern arranged so that nothing cancels at coefficient 0.  The CPU twin of
the external-product case is tests/test_emu_kernels.py::test_external_product_worst_case_magnitudes
(same operands through the device headers on x86)."""
import numpy as np
import pytest

from gpu_common import pkg, to_pkg_params

pytestmark = pytest.mark.gpu

# k, logN, (logB, levels): the four BASELINE shapes, the 49-bit field's row and growth limits
# (20 rows / 10 stages; 11 stages over four waves), a wide base that only Goldilocks lifts
SHAPES = [
    ("cfg1", 1, 9, (8, 2)),
    ("cfg2", 1, 10, (7, 3)),
    ("cfg3", 2, 9, (4, 6)),
    ("cfg5", 2, 11, (8, 4)),
    ("p49_rows20_n1024", 1, 10, (2, 10)),
    ("p49_n2048", 2, 11, (2, 5)),
    ("wide_base", 1, 9, (16, 2)),
    # the admission edge of fp64-fft: the sets with the largest proven rounding bound below 1/4 that AUTO hands to it
    # (no prime field admits log2 B > 9): 0.209, 0.174 and 0.160 (csrc/field_fft.h::error_bound)
    ("fft_edge_n1024_k2_b11", 2, 10, (11, 2)),
    ("fft_edge_n512_k1_b13", 1, 9, (13, 2)),
    ("fft_edge_n1024_k2_b10_l3", 2, 10, (10, 3)),
]
BACKENDS = ["fp64", "fp64-p49", "fp64-fft", "goldilocks", "goldilocks-split"]
KEY_WORDS = [0x7FFF7FFF, 0x80008000, 0x7FFF8000, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF]


def backend_id(name):
    m = pkg()
    return {"fp64": m.BACKEND_FP64, "goldilocks": m.BACKEND_GOLDILOCKS, "goldilocks-split": m.BACKEND_GOLDILOCKS_SPLIT,
            "fp64-p49": m.BACKEND_FP64_P49, "fp64-fft": m.BACKEND_FP64_FFT}[name]


def extreme_words(oracle, pbs):
    """(word whose digits sum to the most positive value, word with the most negative one): every kept
    limb at B (carry chain) resp. -B/2, found by scoring candidates with the oracle's decomposer"""
    rng = np.random.default_rng(7)
    cand = np.concatenate([rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32),
                           np.array([0xFFFFFFFF, 0x7FFFFFFF, 0xF8F8F8F8, 0xFFFFFF80, 0x80000000, 0x88888888, 0x77777777],
                                    dtype=np.uint32)])
    d = oracle.decompose(oracle.Decomposer(*pbs), cand).astype(np.int32).astype(np.int64)
    score = d.sum(axis=1)
    return int(cand[int(score.argmax())]), int(cand[int(score.argmin())])


def open_context(oracle, k, logn, pbs, backend, n=1):
    m = pkg()
    p = oracle.Params(k, logn, n, oracle.Decomposer(*pbs))
    try:
        return p, m.Context(to_pkg_params(p), backend=backend_id(backend))
    except m.TfheError as e:
        if e.status == m.TFHE_ERR_EXACTNESS:
            pytest.skip("outside this field's exactness bound: tfhe_context_create refuses it")
        if e.status == m.TFHE_ERR_UNSUPPORTED and backend == "fp64-fft":
            pytest.skip("the complex-FFT backend has kernels at N = 512 and 1024")
        raise


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,k,logn,pbs", SHAPES, ids=[s[0] for s in SHAPES])
def test_external_product_at_the_exactness_bound(oracle, name, k, logn, pbs, backend):
    """tfhe_external_product_batch (ggsw.rs:132-161): 12 (key word, digit pattern) pairs in one batch,
    one GGSW per sample, bit-exact against the oracle"""
    p, ctx = open_context(oracle, k, logn, pbs, backend)
    wpos, wneg = extreme_words(oracle, pbs)
    N = p.N
    ggsw, glwe = [], []
    for key_word in KEY_WORDS:
        for glwe_word in (wpos, wneg):
            g = np.full((p.R, k + 1, N), key_word, dtype=np.uint32)
            # X^N = -1: negating every key coefficient but the first makes all N terms of coefficient 0
            # of the negacyclic product carry the same sign
            g[:, :, 1:] = (np.uint32(0) - g[:, :, 1:]).astype(np.uint32)
            ggsw.append(g)
            glwe.append(np.full((k + 1, N), glwe_word, dtype=np.uint32))
    ggsw, glwe = np.stack(ggsw), np.stack(glwe)
    with ctx:
        got = ctx.external_product(ggsw, glwe)
        shared = ctx.external_product(ggsw[0], glwe)   # the blind-rotation shape: one GGSW for the batch
    for b in range(ggsw.shape[0]):
        assert np.array_equal(got[b], oracle.external_product(p, ggsw[b], glwe[b])), (name, backend, b)
    for b in (0, 1):
        assert np.array_equal(shared[b], oracle.external_product(p, ggsw[0], glwe[b])), (name, backend, b)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,k,logn,pbs", SHAPES, ids=[s[0] for s in SHAPES])
def test_external_product_at_the_bound_with_random_signs(oracle, name, k, logn, pbs, backend):
    """The same magnitudes -- every digit +B or -B/2, every signed 16-bit key half +-2^15 -- with an independent
    random choice per coefficient.  Constant polynomials maximise |z| (the exact fields' bound); for the complex
    transform's ROUNDING error the adversarial operands are dense ones with full-magnitude spectra everywhere, which
    random signs give.  32 products per shape: 8 of them against the oracle word for word, all 32 against the
    Goldilocks-split backend (exact integer arithmetic, lifts every base the reference can express), and the
    shared-GGSW launch shape against the per-sample one."""
    p, ctx = open_context(oracle, k, logn, pbs, backend)
    wpos, wneg = extreme_words(oracle, pbs)
    N = p.N
    rng = np.random.default_rng(logn * 131 + k * 17 + pbs[0])
    key_words = np.array([0x7FFF7FFF, 0x80008000, 0x7FFF8000, 0x80007FFF], dtype=np.uint32)
    batch = 32
    ggsw = rng.choice(key_words, size=(batch, p.R, k + 1, N))
    glwe = rng.choice(np.array([wpos, wneg], dtype=np.uint32), size=(batch, k + 1, N))
    with ctx:
        got = ctx.external_product(ggsw, glwe)
        shared = ctx.external_product(ggsw[0], glwe)
        aut = None
        if ctx.backend != "goldilocks-split":
            # an independent exact route for the rows the oracle does not redo: Goldilocks-split lifts every base
            m = pkg()
            with m.Context(to_pkg_params(p), backend=m.BACKEND_GOLDILOCKS_SPLIT) as ref:
                aut = ref.external_product(ggsw, glwe)
    for b in range(8):
        assert np.array_equal(got[b], oracle.external_product(p, ggsw[b], glwe[b])), (name, backend, b)
    assert np.array_equal(shared[0], got[0]), (name, backend)
    assert np.array_equal(shared[1], oracle.external_product(p, ggsw[0], glwe[1])), (name, backend)
    if aut is not None:
        assert np.array_equal(got, aut), (name, backend)


def test_auto_picks_fft_at_the_admission_edge(oracle):
    """AUTO hands the edge sets to fp64-fft (the proof carries the weight there), and refuses one more bit of base"""
    m = pkg()
    for name, k, logn, pbs in SHAPES:
        if not name.startswith("fft_edge"):
            continue
        p = oracle.Params(k, logn, 4, oracle.Decomposer(*pbs))
        with m.Context(to_pkg_params(p)) as ctx:
            assert ctx.backend == "fp64-fft", (name, ctx.backend)
    p = oracle.Params(2, 10, 4, oracle.Decomposer(12, 2))
    with pytest.raises(m.TfheError) as e:
        m.Context(to_pkg_params(p), backend=m.BACKEND_FP64_FFT)
    assert e.value.status == m.TFHE_ERR_EXACTNESS
    with m.Context(to_pkg_params(p)) as ctx:     # AUTO falls through to an exact field that lifts it
        assert ctx.backend in ("goldilocks", "goldilocks-split")


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,k,logn,pbs", SHAPES, ids=[s[0] for s in SHAPES])
def test_blind_rotation_with_extreme_keys(oracle, name, k, logn, pbs, backend):
    """tfhe_blind_rotate_batch (bootstrapping.rs:79-105) with a bootstrapping key made of extreme words
    only: the blind-rotation kernel (a separate instantiation of the same team code, accumulator in
    LDS, digits taken from the rotated difference) sees key spectra of maximal magnitude while the
    accumulator runs through whatever the products leave; 4 CMUXes, per-sample test vectors of
    maximal plaintext words, bit-exact against the oracle"""
    n = 4
    p, ctx = open_context(oracle, k, logn, pbs, backend, n=n)
    N = p.N
    rng = np.random.default_rng(len(name))
    bsk = rng.choice(np.array(KEY_WORDS, dtype=np.uint32), size=p.bsk_shape())
    bsk[0] = 0x80000000
    bsk[1, :, :, 0::2] = 0x7FFFFFFF
    bsk[1, :, :, 1::2] = 0x80000001
    ksk = rng.integers(0, 1 << 32, size=p.ksk_shape(), dtype=np.uint64).astype(np.uint32)
    lwe = rng.integers(0, 1 << 32, size=(6, n + 1), dtype=np.uint64).astype(np.uint32)
    lwe[0, :n] = 1 << (32 - logn - 1)          # a~ = 1: X * acc - acc, dense differences
    lwe[1, :n] = 0x80000000                    # a~ = N: -2 acc
    tvs = np.full((6, N), (1 << p.log_p) - 1, dtype=np.uint32)
    tvs[3:] = rng.integers(0, 1 << p.log_p, size=(3, N))
    with ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        acc = ctx.blind_rotate(lwe, tvs)
        out = ctx.bootstrap(lwe, tvs)
    for b in range(lwe.shape[0]):
        want, tr = oracle.bootstrap(p, lwe[b], bsk, ksk, tvs[b], trace=True)
        assert np.array_equal(acc[b], tr["acc_final"]), (name, backend, b)
        assert np.array_equal(out[b], want), (name, backend, b)

"""CPU parity tests of the *device source* (tfhe-research_amd/csrc/*.h) through the host SIMT
emulator (tests/emu): field arithmetic vs Python integers, the per-wave NTT vs the executable
model in tools/ntt_model.py, and external product / blind rotation / sample extract vs the oracle.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ntt_model  # noqa: E402

P = ntt_model.P
EMU_DIR = os.path.join(HERE, "emu")
CSRC = os.path.join(ROOT, "tfhe-research_amd", "csrc")


# Goldilocks, fp64 42-bit prime, Goldilocks with the key split in 16-bit halves, fp64 49-bit prime,
# complex FFT in fp64 (exact by its rounding-error bound)
GL, FP, GLS, FP49, FFT = 1, 2, 3, 4, 5
FIELDS = [GL, FP, GLS, FP49, FFT]


def fft_error_bound(logn, rows, log_base):
    """field_fft.h::FftField::error_bound"""
    import math
    u = 2.0 ** -53
    m, n = float(1 << (logn - 1)), float(logn - 1)
    # three transforms of n stages (eta = 7.1 u each) + the FMA chain accumulating `rows` products (quadratic in rows)
    per_unit = 1.001 * (3.0 * n * 7.1 * u + 1.42 * (rows + 1.0) * u)
    return per_unit * rows * m * math.sqrt(m) * math.sqrt(2) * (1 << log_base) * math.sqrt(2) * 32768.0


# the parameter sets with the largest proven bound below 1/4 that AUTO gives to fp64-fft (no prime field admits
# log2 B > 9 and the 49-bit one stops at N B rows < 2^17.25): N = 1024, k = 2, l = 2, B = 2^11 (0.209) and
# N = 512, k = 1, l = 2, B = 2^13 (0.174); next to them N = 1024, k = 2, l = 3, B = 2^10 (0.160)
FFT_EDGE_SHAPES = [(2, 10, (11, 2), 1), (1, 9, (13, 2), 1), (2, 10, (10, 3), 1)]


def extreme_operands(oracle, params, pbs, rng, random_signs):
    """GGSW and GLWE at the magnitude bound of the exactness argument: every digit +B or -B/2, every signed 16-bit
    key half at +-2^15.  random_signs=False: constant polynomials with the negacyclic sign pattern that makes all N
    terms of coefficient 0 add up (maximal |z|).  random_signs=True: the same magnitudes with an independent random
    choice per coefficient -- the adversarial case for the FFT's ROUNDING error, which a constant polynomial (one
    non-zero spectral pattern) does not exercise."""
    k, N = params.k, params.N
    cand = np.concatenate([rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32),
                           np.array([0xFFFFFFFF, 0x7FFFFFFF, 0xF8F8F8F8, 0xFFFFFF80, 0x80000000], dtype=np.uint32)])
    score = oracle.decompose(oracle.Decomposer(*pbs), cand).astype(np.int32).astype(np.int64).sum(axis=1)
    wpos, wneg = cand[int(score.argmax())], cand[int(score.argmin())]
    key_words = np.array([0x7FFF7FFF, 0x80008000, 0x7FFF8000, 0x80007FFF], dtype=np.uint32)  # halves (+,+) (-,-) (-,-) (+,-)
    if random_signs:
        ggsw = rng.choice(key_words, size=(params.R, k + 1, N))
        glwe = rng.choice(np.array([wpos, wneg], dtype=np.uint32), size=(k + 1, N))
        return [(ggsw, glwe)]
    out = []
    for kw in key_words[:3]:
        for gw in (wpos, wneg):
            ggsw = np.full((params.R, k + 1, N), kw, dtype=np.uint32)
            ggsw[:, :, 1:] = (np.uint32(0) - ggsw[:, :, 1:]).astype(np.uint32)
            out.append((ggsw, np.full((k + 1, N), gw, dtype=np.uint32)))
    return out


def field_exact(field, k, logn, pbs, g=1):
    """the bounds tfhe_context_create_with_backend checks (capi.cpp): worst-case |convolution| of
    the (k+1)*l digit rows against the field's capacity (for the complex transform: the proven rounding
    error of an output coefficient against 1/4, and the shapes the emulator can run it in)"""
    import math
    rows = (k + 1) * pbs[1]
    bits = math.log2(rows) + logn + pbs[0]
    if field == FFT:
        shape = (logn in (9, 10) and g == 1) or logn == 11
        return shape and pbs[0] <= 16 and fft_error_bound(logn, rows, pbs[0]) < 0.25
    if field == FP:
        return bits + 15 < 40.9 and pbs[0] <= 9
    if field == FP49:
        return bits + 31 < 48.25 and rows <= 20
    if field == GL:
        return bits + 32 < 62
    return bits + 15 < 62


def pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def p32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def test_field_arithmetic(emu):
    rng = np.random.default_rng(0)
    edge = [0, 1, 2, P - 1, P - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63), (1 << 63) - 1,
            P - (1 << 32), 0xFFFFFFFF00000000, 0xFFFFFFFE00000001, 0x00000000FFFFFFFF]
    vals = edge + [int(x) % P for x in rng.integers(0, 1 << 63, size=300, dtype=np.uint64) * 2 + 1]
    for a in vals[:40]:
        for b in vals[:40]:
            assert emu.emu_gl_mul(C.c_uint64(a), C.c_uint64(b)) == a * b % P
            assert emu.emu_gl_add(C.c_uint64(a), C.c_uint64(b)) == (a + b) % P
            assert emu.emu_gl_sub(C.c_uint64(a), C.c_uint64(b)) == (a - b) % P
    a = np.array([int(x) % P for x in rng.integers(0, 1 << 64, size=200000, dtype=np.uint64)], dtype=np.uint64)
    b = np.array([int(x) % P for x in rng.integers(0, 1 << 64, size=200000, dtype=np.uint64)], dtype=np.uint64)
    # force the reduction corner cases: operands near p and products with extreme limbs
    a[:1000] = P - 1 - np.arange(1000, dtype=np.uint64)
    b[:1000] = P - 1 - np.arange(1000, dtype=np.uint64)[::-1]
    a[1000:2000] = np.uint64(0xFFFFFFFF) << np.uint64(32)
    out = np.zeros_like(a)
    emu.emu_gl_mul_many(p64(a), p64(b), p64(out), C.c_size_t(a.size))
    want = np.array([int(x) * int(y) % P for x, y in zip(a.tolist(), b.tolist())], dtype=np.uint64)
    assert np.array_equal(out, want)
    for d in (0, 1, 127, 128, 256, 65536, -1, -64, -128, -32768):
        assert emu.emu_gl_from_i32(C.c_uint32(d & 0xFFFFFFFF)) == d % P
    for x in (0, 1, 12345, (1 << 55), -1, -(1 << 55), -77, (1 << 32) - 1, -(1 << 32)):
        assert emu.emu_gl_lift(C.c_uint64(x % P)) == x % (1 << 32)
    # a weakly reduced representative of a small positive value lifts correctly too
    assert emu.emu_gl_lift(C.c_uint64(P + 5)) == 5


@pytest.mark.parametrize("logn", [9, 10, 11])
def test_wave_ntt_matches_model(emu, logn):
    n = 1 << logn
    fwd, inv = ntt_model.tables(logn)
    # + psi_rev[1]*psi_rev[2], psi_rev[1]*psi_rev[3] and 16 fused-stage constants (the 42-bit field's)
    tw = np.zeros(n + 18, dtype=np.uint64)
    emu.emu_twiddles(GL, logn, p64(tw))
    assert tw[:n].tolist() == fwd
    assert tw[n:n + 2].tolist() == [fwd[1] * fwd[2] % P, fwd[1] * fwd[3] % P]
    rng = np.random.default_rng(logn)
    a = np.array([int(x) % P for x in rng.integers(0, 1 << 64, size=n, dtype=np.uint64)], dtype=np.uint64)
    out = np.zeros_like(a)
    assert emu.emu_poly_ntt(GL, logn, 1, p64(a), p64(out), 0) == 0
    ref = ntt_model.ntt_ref(a.tolist(), logn, fwd)
    assert out.tolist() == ref
    back = np.zeros_like(a)
    assert emu.emu_poly_ntt(GL, logn, 1, p64(out), p64(back), 1) == 0
    assert back.tolist() == [x * n % P for x in a.tolist()]  # unscaled inverse


def test_fp_field_arithmetic(emu):
    """fp64 field: mul/reduce/to_u32 against Python integers, including the magnitudes the
    transforms can reach (|a| up to 2^52.9, |w| <= p/2)."""
    p = int(emu.emu_fp_p())
    assert p == (1 << 42) - 24575
    rng = np.random.default_rng(1)
    n = 200000
    a = rng.integers(-(1 << 52), 1 << 52, size=n).astype(np.float64)
    a[:8] = [0, 1, -1, (1 << 53) - 2, -((1 << 53) - 2), p, -p, (p - 1) // 2]
    a[8:1000] = rng.integers(-(1 << 53) + 2, (1 << 53) - 2, size=992).astype(np.float64)
    w = rng.integers(-(p // 2), p // 2 + 1, size=n).astype(np.float64)
    w[:4] = [p // 2, -(p // 2), 1, -1]
    out = np.zeros(n)
    emu.emu_fp_mul_many(pd(a), pd(w), pd(out), C.c_size_t(n))
    ai, wi, oi = [int(x) for x in a], [int(x) for x in w], [int(x) for x in out]
    assert all(float(o) == fo for o, fo in zip(oi, out))               # integer valued
    assert all((o - x * y) % p == 0 for o, x, y in zip(oi, ai, wi))    # congruent
    bound = [p / 2 + 1.5 * abs(x) * p / 2 ** 53 + 2 for x in ai]
    assert all(abs(o) <= b for o, b in zip(oi, bound))                 # growth bound used in field_fp.h
    red = np.zeros(n)
    emu.emu_fp_reduce_many(pd(a), pd(red), C.c_size_t(n))
    assert all((int(r) - x) % p == 0 and abs(int(r)) <= p // 2 + 1 for r, x in zip(red, ai))
    t = rng.integers(-(1 << 51) + 1, 1 << 51, size=n).astype(np.float64)   # to_u32's domain: |t| < 2^51
    t[:8] = [0, -1, (1 << 32), -(1 << 32), (1 << 32) - 1, -(1 << 31), (1 << 51) - 1, -(1 << 51) + 1]
    u = np.zeros(n, dtype=np.uint32)
    emu.emu_fp_to_u32_many(pd(t), p32(u), C.c_size_t(n))
    assert u.tolist() == [int(x) % (1 << 32) for x in t]
    for word in (0, 1, 0x7FFF, 0x8000, 0xFFFF, 0x10000, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFF, 0x8000FFFF, 0x12348765):
        lo = emu.emu_fp_from_key_word(C.c_uint32(word), 0)
        hi = emu.emu_fp_from_key_word(C.c_uint32(word), 1)
        assert abs(lo) <= 1 << 15 and abs(hi) <= 1 << 15
        assert (int(lo) + (int(hi) << 16) - word) % (1 << 32) == 0


@pytest.mark.parametrize("logn,g", [(9, 1), (10, 1), (11, 1), (11, 2), (11, 4)])
def test_fp_wave_ntt_roundtrip_and_convolution(emu, logn, g):
    """fp64 field transform: forward of two small polynomials, pointwise product, inverse = exact
    negacyclic convolution (checked with numpy integers)."""
    n = 1 << logn
    p = int(emu.emu_fp_p())
    rng = np.random.default_rng(logn)
    a = rng.integers(-256, 257, size=n).astype(np.float64)
    b = rng.integers(-(1 << 15), (1 << 15) + 1, size=n).astype(np.float64)
    fa, fb = np.zeros(n), np.zeros(n)
    assert emu.emu_poly_ntt(FP, logn, g, pd(a), pd(fa), 0) == 0
    assert emu.emu_poly_ntt(FP, logn, g, pd(b), pd(fb), 0) == 0
    assert np.abs(fa).max() <= 6.2 * p and np.abs(fb).max() <= 6.2 * p
    ninv = pow(n, p - 2, p)
    prod = np.array([float(((int(x) * int(y) % p) * ninv) % p) for x, y in zip(fa, fb)])
    prod = np.where(prod > p // 2, prod - p, prod)
    back = np.zeros(n)
    assert emu.emu_poly_ntt(FP, logn, g, pd(prod), pd(back), 1) == 0
    assert np.abs(back).max() < 2 ** 53
    ai, bi = a.astype(np.int64), b.astype(np.int64)
    full = np.convolve(ai, bi)
    want = full[:n].copy()
    want[: n - 1] -= full[n:]
    got = np.array([int(x) % p for x in back])
    assert np.array_equal(got, want % p)


@pytest.mark.parametrize("logn", [9, 10, 11])
def test_fp_fused_stage_constants(emu, logn):
    """entries [n .. n+18) of the 42-bit field's table: w1*w2a, w1*w2b and, for the fused third stage,
    w3[q] * (1, +-w2, +-w1, +-w1*w2) with the signs of radix-4 output q (field_fp.h::radix8_small_v)"""
    n = 1 << logn
    p = int(emu.emu_fp_p())
    tw = np.zeros(n + 18)
    emu.emu_twiddles(FP, logn, pd(tw))
    t = [int(x) % p for x in tw]
    w1, w2a, w2b = t[1], t[2], t[3]
    assert t[n] == w1 * w2a % p and t[n + 1] == w1 * w2b % p
    signs = {0: (1, 1, 1), 1: (-1, 1, -1), 2: (1, -1, -1), 3: (-1, -1, 1)}  # (b, c, d) of output q
    for q in range(4):
        w3, w2 = t[4 + q], (w2a if q < 2 else w2b)
        sb, sc, sd = signs[q]
        want = [w3, sb * w3 * w2 % p, sc * w3 * w1 % p, sd * w3 * w1 * w2 % p]
        assert t[n + 2 + 4 * q: n + 6 + 4 * q] == [x % p for x in want], q
    assert all(abs(x) <= p // 2 for x in tw)


def prepared(emu, field, params, bsk, g=1):
    flat = np.ascontiguousarray(bsk, dtype=np.uint32).reshape(-1, params.N)
    parts = emu.emu_field_parts(field)
    out = np.zeros((flat.shape[0], parts, params.N), dtype=np.uint64)
    emu.emu_set_key_k(params.k)   # the key's layout depends on (field, N, k): pbs_wave.h::key_layout_e
    try:
        assert emu.emu_bsk_prepare(field, params.glwe_poly_degree, g, C.c_size_t(flat.shape[0]), p32(flat), p64(out)) == 0
    finally:
        emu.emu_set_key_k(0)
    return out


CASES = [
    # k, logN, n, (logB, levels), log_p, waves per polynomial
    (1, 9, 3, (8, 2), 2, 1),
    (1, 10, 3, (7, 3), 2, 1),   # BASELINE cfg2 shape, misaligned base (bits 28..31 dropped)
    (2, 9, 2, (4, 6), 2, 1),    # reference default shape
    (2, 11, 1, (8, 4), 4, 1),   # BASELINE cfg5 shape, one wave per polynomial
    (2, 11, 1, (8, 4), 4, 2),   # BASELINE cfg5 shape, two waves per polynomial
    (2, 11, 1, (8, 4), 4, 4),   # BASELINE cfg5 shape, four waves per polynomial (the shipped mapping)
    (1, 11, 2, (4, 7), 2, 2),
]

# shapes inside the 49-bit field's bounds beyond the reference default (which is in CASES)
CASES_P49 = [
    (1, 10, 2, (2, 10), 2, 1, 1),  # 20 digit rows, 10 stages: the field's row and growth limits
    (2, 11, 1, (2, 5), 2, 4, 2),   # 11 stages over four waves, two exchange buffers (as shipped)
    (1, 9, 3, (4, 6), 2, 1, 1),
]


@pytest.fixture(params=[1, 2], ids=["one-exchange-buffer", "two-exchange-buffers"])
def exchange_buffers(emu, request):
    """both spectrum-exchange schemes of external_product_team (kernels.hip picks per ring degree)"""
    emu.emu_set_exchange_buffers(request.param)
    yield request.param
    emu.emu_set_exchange_buffers(1)


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,logn,n,pbs,log_p,g", CASES)
def test_external_product_vs_oracle(emu, oracle, exchange_buffers, field, k, logn, n, pbs, log_p, g):
    if not field_exact(field, k, logn, pbs, g):
        pytest.skip("outside this field's exactness bound")
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    rng = np.random.default_rng(11 * logn + k)
    ggsw = rng.integers(0, 1 << 32, size=(params.R, k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    glwe = rng.integers(0, 1 << 32, size=(k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    # hit the digit == B and digit == -B/2 paths in every polynomial
    glwe[:, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
    spec = prepared(emu, field, params, ggsw, g)
    out = np.zeros_like(glwe)
    assert emu.emu_external_product(field, g, k, logn, pbs[0], pbs[1], p64(spec), p32(glwe), p32(out)) == 0
    assert np.array_equal(out, oracle.external_product(params, ggsw, glwe))


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,logn,pbs,g", [(1, 10, (7, 3), 1), (2, 11, (8, 4), 2), (2, 11, (8, 4), 4), (2, 9, (4, 6), 1), (1, 9, (16, 2), 1),
                                          (1, 10, (2, 10), 1), (2, 11, (2, 5), 4)] + FFT_EDGE_SHAPES)
def test_external_product_worst_case_magnitudes(emu, oracle, field, k, logn, pbs, g):
    """Adversarial inputs that drive the integer convolution to its bound: every digit at +B or
    -B/2 and every key word at 0x7FFF8000-type extremes (both 16-bit halves maximal), aligned so
    that the negacyclic sums do not cancel.  The exact-NTT bound must hold, not just typical inputs."""
    params = oracle.Params(k, logn, 1, oracle.Decomposer(*pbs))
    log_base, levels = pbs
    N = params.N
    if not field_exact(field, k, logn, pbs, g):
        pytest.skip("outside this field's exactness bound: the context selects another field here")
    if field == FFT and logn == 11 and g != 4:
        pytest.skip("the complex transform at N = 2048 (not shipped on the GPU) keeps its five-pass shape only: run time")
    first_shift = log_base * (32 // log_base - levels)
    # word whose every kept limb is B-1 with an incoming carry -> digits B ... (top ones), built by brute force
    def word_with_digits(target):
        rng = np.random.default_rng(7)
        best, best_score = 0, -1
        cand = np.concatenate([rng.integers(0, 1 << 32, size=200000, dtype=np.uint64).astype(np.uint32),
                               np.array([0xFFFFFFFF, 0x7FFFFFFF, 0xF8F8F8F8, 0xFFFFFF80], dtype=np.uint32)])
        d = oracle.decompose(oracle.Decomposer(*pbs), cand).astype(np.int32).astype(np.int64)
        score = (d * target).sum(axis=1)
        return int(cand[int(score.argmax())])
    wpos = word_with_digits(+1)
    wneg = word_with_digits(-1)
    # 0x80000000 / 0x7FFFFFFF: the extremes of the whole signed word (the 49-bit field's key operand)
    for key_word, glwe_word in ((0x7FFF7FFF, wpos), (0x80008000, wpos), (0x7FFF8000, wneg), (0xFFFFFFFF, wneg),
                                (0x80000000, wpos), (0x80000000, wneg), (0x7FFFFFFF, wpos), (0x7FFFFFFF, wneg)):
        ggsw = np.full((params.R, k + 1, N), key_word, dtype=np.uint32)
        glwe = np.full((k + 1, N), glwe_word, dtype=np.uint32)
        # negacyclic sign pattern: make the key alternate sign across the wrap so sums add up at coefficient 0
        ggsw[:, :, 1:] = (np.uint32(0) - ggsw[:, :, 1:]).astype(np.uint32)
        spec = prepared(emu, field, params, ggsw, g)
        out = np.zeros_like(glwe)
        assert emu.emu_external_product(field, g, k, logn, log_base, levels, p64(spec), p32(glwe), p32(out)) == 0
        assert np.array_equal(out, oracle.external_product(params, ggsw, glwe)), hex(key_word)


@pytest.mark.parametrize("k,logn,pbs,g", [(1, 10, (7, 3), 1), (1, 9, (8, 2), 1), (2, 9, (4, 6), 1), (2, 11, (8, 4), 2)] + FFT_EDGE_SHAPES)
def test_fft_rounding_margin(emu, oracle, k, logn, pbs, g):
    """The complex-FFT backend is exact because every lifted value is within 1/2 of the integer it stands for.
    field_fft.h proves a bound (FftField::error_bound, admitted below 1/4); this measures the distance the emulator
    actually sees -- on operands at the magnitude bound (constant-sign AND random-sign ones) and on random ones -- and
    holds it against that bound, and the Python restatement of the bound against the C++ one.  The shapes include the
    admission edge: the parameter sets with the largest bound below 1/4 that AUTO hands to this backend."""
    emu.emu_fft_error_bound.restype = C.c_double
    emu.emu_fft_error_max.restype = C.c_double
    log_base, levels = pbs
    rows = (k + 1) * levels
    bound = emu.emu_fft_error_bound(logn, rows, log_base)
    assert abs(bound - fft_error_bound(logn, rows, log_base)) < 1e-12 * bound
    assert bound < 0.25
    params = oracle.Params(k, logn, 1, oracle.Decomposer(*pbs))
    N = params.N
    rng = np.random.default_rng(logn + 7 * k)
    worst = 0.0
    cases = extreme_operands(oracle, params, pbs, rng, False)
    for _ in range(4):
        cases += extreme_operands(oracle, params, pbs, rng, True)
    for _ in range(2):
        cases.append((rng.integers(0, 1 << 32, size=(params.R, k + 1, N), dtype=np.uint64).astype(np.uint32),
                      rng.integers(0, 1 << 32, size=(k + 1, N), dtype=np.uint64).astype(np.uint32)))
    for ggsw, glwe in cases:
        spec = prepared(emu, FFT, params, ggsw, g)
        out = np.zeros_like(glwe)
        emu.emu_fft_error_reset()
        assert emu.emu_external_product(FFT, g, k, logn, log_base, levels, p64(spec), p32(glwe), p32(out)) == 0
        worst = max(worst, emu.emu_fft_error_max())
        assert np.array_equal(out, oracle.external_product(params, ggsw, glwe))
    # measured distance from the integers: orders of magnitude inside the proven bound, itself inside 1/4
    assert worst < bound / 100, (worst, bound)
    print(f"fft rounding margin N=2^{logn} k={k} rows={rows} B=2^{log_base}: measured {worst:.3g}, proven bound {bound:.3g}")


def test_hoisted_rounding_equals_the_literal_one(emu):
    """round_value_fast ((v + half) & keep, the hot loop's two-instruction form) against round_value (decomposer.rs:27-40
    restated literally) for every number of ignored bits: the words around every wrap and carry, and a strided sweep of
    the whole u32 range"""
    emu.emu_round_value_mismatches.restype = C.c_ulonglong
    for ig in range(0, 32):
        assert emu.emu_round_value_mismatches(ig, 0, 1, C.c_ulonglong(1 << 16)) == 0
        assert emu.emu_round_value_mismatches(ig, 0xFFFF0000, 1, C.c_ulonglong(1 << 16)) == 0
        assert emu.emu_round_value_mismatches(ig, 0x7FFF8000, 1, C.c_ulonglong(1 << 16)) == 0
        assert emu.emu_round_value_mismatches(ig, 12345, 65521, C.c_ulonglong(1 << 18)) == 0


def test_fft_error_bound_is_quadratic_in_the_rows():
    """The accumulation term of the bound: R products summed by an FMA chain round the running sum R times, so the
    rounding part grows like R (R + 1), not like R.  Pinned numbers of the corrected formula."""
    assert abs(fft_error_bound(10, 6, 7) - 0.013067) < 2e-6      # cfg2
    assert abs(fft_error_bound(9, 18, 4) - 0.0016958) < 2e-7     # the reference's default parameters
    assert abs(fft_error_bound(10, 6, 11) - 0.20907) < 2e-5      # the largest admitted set at N = 1024
    assert fft_error_bound(10, 6, 12) > 0.25                     # one more bit of base: refused
    r1, r2 = fft_error_bound(10, 10, 2), fft_error_bound(10, 20, 2)
    assert r2 / r1 > 2.05                                        # more than linear in the rows


@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,logn,n,pbs,log_p,g", CASES)
def test_blind_rotate_and_extract_vs_oracle(emu, oracle, exchange_buffers, field, k, logn, n, pbs, log_p, g):
    if not field_exact(field, k, logn, pbs, g):
        pytest.skip("outside this field's exactness bound")
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 2
    lut = np.random.default_rng(5).integers(0, 1 << log_p, size=1 << log_p)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(params, batch, cfg_index=40 + logn, lut=lut)
    lwe = lwe.copy()
    lwe[0, 0] = 0            # a~ = 0: the skipped iteration
    lwe[1, n] = 0xFFFFFFFF   # b~ rounds up to 2N and wraps to 0
    spec = prepared(emu, field, params, bsk, g)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    rc = emu.emu_blind_rotate(field, g, n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tv),
                              C.c_size_t(0), p64(spec), p32(glwe), p32(ext))
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tv, trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"


@pytest.mark.parametrize("k,logn,n,pbs,log_p,g", [(1, 9, 5, (8, 2), 2, 1), (2, 9, 4, (4, 6), 2, 1), (1, 10, 3, (8, 4), 2, 1),
                                                  (2, 11, 3, (8, 4), 4, 4), (1, 11, 3, (8, 3), 2, 4)])
def test_two_samples_per_team_vs_oracle(emu, oracle, k, logn, n, pbs, log_p, g):
    """pbs_wave.h::blind_rotate_team_multi with NS = 2 (the complex transform's kernels at N = 512, k = 2 and N = 2048):
    a team rotates two samples at once -- one key fetch, one set of barriers, the inverse transforms in lockstep -- and
    every sample must come out as if it had been alone.  Odd batch (3): the last team redoes its last sample in the
    free slot and writes it once; per-sample test vectors; a~ = 0 and b~ -> 2N rows included."""
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 3
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(params, batch, cfg_index=70 + logn)
    rng = np.random.default_rng(logn * 3 + k)
    tvs = rng.integers(0, 1 << log_p, size=(batch, params.N)).astype(np.uint32)
    lwe = lwe.copy()
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    spec = prepared(emu, FFT, params, bsk, g)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    emu.emu_set_samples_per_team(2)
    try:
        rc = emu.emu_blind_rotate(FFT, g, n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tvs),
                                  C.c_size_t(params.N), p64(spec), p32(glwe), p32(ext))
    finally:
        emu.emu_set_samples_per_team(1)
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tvs[b], trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"


WIDE_CASES = [
    (1, 10, 4, (7, 3), 2),    # cfg2's shape: levels 2 + 1 over the two halves (the literal, misaligned decomposer)
    (1, 10, 3, (8, 4), 2),    # every CMUX depends on the key; levels 2 + 2
    (2, 9, 4, (4, 6), 2),     # the reference's parameters: six waves, 18 digit rows, odd chunk count per level
    (1, 9, 5, (8, 2), 2),     # cfg1's shape: one level per half
    (2, 10, 3, (10, 3), 2),   # an admission-edge shape of the rounding bound
    (1, 10, 3, (9, 1), 2),    # ONE level: the second half transforms nothing, more waves than digit rows
    (1, 9, 3, (5, 5), 4),     # five levels (3 + 2), log_p = 4
]


@pytest.mark.parametrize("k,logn,n,pbs,log_p", WIDE_CASES)
@pytest.mark.parametrize("segments,key_ring", [(1, 1), (2, 1), (1, 0)])
def test_wide_team_blind_rotation_vs_oracle(emu, oracle, k, logn, n, pbs, log_p, segments, key_ring):
    """pbs_wave.h::blind_rotate_team_wide -- the latency shape for small batches: 2 (k+1) waves per sample, wave (c, q)
    transforms half of polynomial c's digit levels, accumulates key part q of column c over all rows, inverse-transforms
    that one accumulator and adds lift << 16 q into the accumulator polynomial.  Same words as the reference's loop
    (oracle trace): a~ = 0 and b~ -> 2N rows, per-sample test vectors, whole and segmented rotations; with the KEY RING
    (pbs_wave.h::WideKeyRing: the instantiations for 2, 3, 4 and 6 levels keep whole rows of key tiles in registers and refill
    a slot with the next row -- of the next CMUX's GGSW at the end of a product) and with the generic kernel's chunked loads."""
    if not field_exact(FFT, k, logn, pbs):
        pytest.skip("outside the rounding bound")
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 3
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(params, batch, cfg_index=130 + logn + k)
    rng = np.random.default_rng(logn * 5 + k)
    tvs = rng.integers(0, 1 << log_p, size=(batch, params.N)).astype(np.uint32)
    lwe = lwe.copy()
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    spec = prepared(emu, FFT, params, bsk, 1)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    emu.emu_set_segments(segments)
    emu.emu_set_wide_key_ring(key_ring)
    try:
        rc = emu.emu_blind_rotate_wide(n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tvs),
                                       C.c_size_t(params.N), p64(spec), p32(glwe), p32(ext))
    finally:
        emu.emu_set_segments(1)
        emu.emu_set_wide_key_ring(1)
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tvs[b], trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"


@pytest.mark.parametrize("n,pbs,log_p,segments", [(5, (8, 2), 2, 1), (4, (8, 2), 2, 2), (3, (4, 6), 2, 1), (3, (13, 2), 2, 1), (3, (9, 1), 2, 1),
                                                  (3, (5, 5), 4, 3)])
def test_pair_kernel_blind_rotation_vs_oracle(emu, oracle, n, pbs, log_p, segments):
    """pbs_wave.h::blind_rotate_pair -- ONE wave per sample at N = 512, k = 1: polynomial c in lanes 32 c .. 32 c + 31, 8
    transform elements per lane (NttShape<8, 0>: three passes, two transposes serving both polynomials), spectra handed
    between the halves through LDS under a wave-level fence, no barrier.  Same words as the reference's loop: cfg1's shape,
    six levels, the rounding bound's admission edge (log2 B = 13), one level, five levels; a~ = 0 and b~ -> 2N rows, per-sample
    test vectors, segmented rotations."""
    k, logn = 1, 9
    if not field_exact(FFT, k, logn, pbs):
        pytest.skip("outside the rounding bound")
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 3
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(params, batch, cfg_index=160 + n)
    rng = np.random.default_rng(n * 11 + pbs[0])
    tvs = rng.integers(0, 1 << log_p, size=(batch, params.N)).astype(np.uint32)
    lwe = lwe.copy()
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    spec = prepared(emu, FFT, params, bsk, 1)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    emu.emu_set_segments(segments)
    try:
        rc = emu.emu_blind_rotate_pair(n, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tvs), C.c_size_t(params.N),
                                       p64(spec), p32(glwe), p32(ext))
    finally:
        emu.emu_set_segments(1)
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tvs[b], trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"


def test_wide_team_with_the_aligned_decomposer(emu, oracle):
    """cfg2's base (log2 B = 7) with the aligned decomposer: the data-dependent case of the headline parameters"""
    params = oracle.Params(1, 10, 4, oracle.Decomposer(7, 3), log_p=2)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(params, 2, cfg_index=141)
    spec = prepared(emu, FFT, params, bsk, 1)
    glwe = np.zeros((2, 2, params.N), dtype=np.uint32)
    emu.emu_set_aligned(1)
    try:
        with oracle.decomposer_aligned(True):
            rc = emu.emu_blind_rotate_wide(4, 1, 10, 2, 1, 7, 3, C.c_size_t(2), p32(lwe), p32(tv), C.c_size_t(0), p64(spec),
                                           p32(glwe), None)
            assert rc == 0
            for b in range(2):
                _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tv, trace=True)
                assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
    finally:
        emu.emu_set_aligned(0)


@pytest.mark.parametrize("field,k,logn,n,pbs,log_p,g,ns,segments", [(FFT, 2, 9, 5, (4, 6), 2, 1, 2, 2), (FFT, 1, 10, 5, (8, 4), 2, 1, 1, 3),
                                                                     (FFT, 2, 11, 3, (8, 4), 4, 4, 2, 3), (FP, 1, 9, 4, (8, 2), 2, 1, 1, 4)])
def test_segmented_blind_rotation_vs_oracle(emu, oracle, field, k, logn, n, pbs, log_p, g, ns, segments):
    """pbs_wave.h::blind_rotate_team_multi over iteration ranges: a rotation cut into several launches (kernels.hip blocks
    the key for the caches that way) parks the accumulators in global memory and resumes from them -- the same words as
    the uninterrupted rotation, with one and with two samples per team, segments that do not divide n, odd batch"""
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 3
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(params, batch, cfg_index=90 + logn)
    spec = prepared(emu, field, params, bsk, g)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    emu.emu_set_samples_per_team(ns)
    emu.emu_set_segments(segments)
    try:
        rc = emu.emu_blind_rotate(field, g, n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tv),
                                  C.c_size_t(0), p64(spec), p32(glwe), p32(ext))
    finally:
        emu.emu_set_samples_per_team(1)
        emu.emu_set_segments(1)
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tv, trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"


@pytest.mark.parametrize("k,logn,n,pbs,log_p,g,exb", CASES_P49)
def test_fp49_field_more_shapes(emu, oracle, k, logn, n, pbs, log_p, g, exb):
    """external product and blind rotation + sample extract in the 49-bit single-spectrum field"""
    assert field_exact(FP49, k, logn, pbs)
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(params, 2, cfg_index=60 + logn)
    spec = prepared(emu, FP49, params, bsk, g)
    emu.emu_set_exchange_buffers(exb)
    try:
        glwe = np.random.default_rng(logn).integers(0, 1 << 32, size=(k + 1, params.N), dtype=np.uint64).astype(np.uint32)
        glwe[:, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
        out = np.zeros_like(glwe)
        assert emu.emu_external_product(FP49, g, k, logn, pbs[0], pbs[1], p64(spec[: params.R * (k + 1)]), p32(glwe), p32(out)) == 0
        assert np.array_equal(out, oracle.external_product(params, bsk[0], glwe))
        acc = np.zeros((2, k + 1, params.N), dtype=np.uint32)
        ext = np.zeros((2, params.big_n + 1), dtype=np.uint32)
        assert emu.emu_blind_rotate(FP49, g, n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(2), p32(lwe), p32(tv),
                                    C.c_size_t(0), p64(spec), p32(acc), p32(ext)) == 0
    finally:
        emu.emu_set_exchange_buffers(1)
    for b in range(2):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tv, trace=True)
        assert np.array_equal(acc[b], tr["acc_final"]) and np.array_equal(ext[b], tr["extracted_lwe"])


def test_wide_base_needs_split_goldilocks(emu, oracle):
    """log_base = 23, l = 1 at N = 2048 (the shape of modern one-level parameter sets): the plain
    convolution bound 2 * 2048 * 2^23 * 2^32 = 2^67 exceeds Goldilocks, the split field lifts it."""
    params = oracle.Params(1, 11, 1, oracle.Decomposer(23, 1))
    rng = np.random.default_rng(23)
    ggsw = rng.integers(0, 1 << 32, size=(params.R, 2, params.N), dtype=np.uint64).astype(np.uint32)
    glwe = rng.integers(0, 1 << 32, size=(2, params.N), dtype=np.uint64).astype(np.uint32)
    # adversarial: every digit at the extreme, key halves at their bounds, signs aligned
    glwe[0, :] = 0x7FFFFE00
    ggsw[:, :, 0] = 0x7FFF7FFF
    ggsw[:, :, 1:] = np.uint32(0x80008001)  # = -0x7FFF7FFF mod 2^32
    spec = prepared(emu, GLS, params, ggsw, 2)
    out = np.zeros_like(glwe)
    assert emu.emu_external_product(GLS, 2, 1, 11, 23, 1, p64(spec), p32(glwe), p32(out)) == 0
    assert np.array_equal(out, oracle.external_product(params, ggsw, glwe))


# ---------------------------------------------------------------- encryption side
@pytest.mark.parametrize("field", FIELDS)
@pytest.mark.parametrize("k,logn,g", [(1, 9, 1), (2, 9, 1), (1, 10, 1), (2, 11, 1), (2, 11, 2), (2, 11, 4)])
def test_glwe_mask_dot_key_vs_oracle(emu, oracle, field, k, logn, g):
    # binary key x whole mask word: k N 2^31 against the field capacity (always inside for k <= 2)
    if not emu.emu_field_shape_ok(field, logn, g):
        pytest.skip("the complex transform has no kernel shape here")
    """glwe_mask_dot_key (the a*s of glwe.rs:197/:252) against the oracle's encrypt_glwe_zero /
    decrypt_glwe_ciphertext with the same pre-drawn samples; includes the all-ones key and
    all-0xFFFFFFFF / 0x8000 / 0x7FFF mask halves that maximise the convolution."""
    params = oracle.Params(k, logn, 16, oracle.Decomposer(4, 3))
    N = params.N
    rng = np.random.default_rng(5 * logn + k + 100 * g)
    rows = 3
    samples = rng.integers(0, 1 << 32, size=(rows, k + 1, N), dtype=np.uint64).astype(np.uint32)
    samples[1, :k] = 0xFFFFFFFF
    samples[2, :k, ::2] = 0x80008000
    samples[2, :k, 1::2] = 0x7FFF7FFF
    for sk in (rng.integers(0, 2, size=(k, N), dtype=np.uint64).astype(np.uint32),
               np.ones((k, N), dtype=np.uint32)):
        out = np.zeros((rows, N), dtype=np.uint32)
        assert emu.emu_glwe_body(field, logn, g, k, rows, p32(samples), p32(sk), p32(out), 0) == 0
        expect = oracle.encrypt_glwe_zero_from_samples(params, sk, samples)
        assert np.array_equal(out, expect[:, k])
        assert np.array_equal(expect[:, :k], samples[:, :k])   # masks untouched
        dec = np.zeros((rows, N), dtype=np.uint32)
        assert emu.emu_glwe_body(field, logn, g, k, rows, p32(expect), p32(sk), p32(dec), 1) == 0
        assert np.array_equal(dec, samples[:, k])              # decrypt gives the error samples back
        for r in range(rows):
            assert np.array_equal(dec[r], oracle.decrypt_glwe_raw(params, sk, expect[r]))


def test_complex_transform_lds_layouts_in_the_bank_model():
    """tools/ntt_model.py::conflicts_b128: the swizzles wave_ntt.h uses for 16-byte elements under gfx950's banking
    of ds_read_b128 / ds_write_b128: N = 1024 (512 points) conflict free in every window; N = 512 (256 points) and
    N = 2048 over four waves conflict free except two-way on the last window's stores"""
    assert all(v == (0, 0) for v in ntt_model.conflicts_b128(9).values())
    c8 = ntt_model.conflicts_b128(8)
    assert all(v == (0, 0) for lo, v in c8.items() if lo != 0) and c8[0][0] == 0 and c8[0][1] <= 32
    c10 = ntt_model.conflicts_b128(10, 4)
    assert all(v == (0, 0) for lo, v in c10.items() if lo != 0) and c10[0][0] == 0 and c10[0][1] <= 4 * 32
    # the pair kernel's half-wave shape (32 lanes x 8 elements): reads and two windows' stores conflict free, the third
    # window's stores two-way in every group: 8 extra LDS-array cycles per store instruction = 16 against the ~14 cycles
    # the store's register transfer takes anyway (profiles/r04_lds_rates_gfx950.txt)
    half, worst_store = ntt_model.conflicts_b128_half()
    assert half[5] == (0, 0) and half[2] == (0, 0) and half[0] == (0, 32) and worst_store == 8


def test_key_word_split_has_no_signed_overflow(emu):
    """w = 0x7FFFxxxx with a negative low half: hi must come out as -32768 (w - lo = 2^31 taken in
    wrapping u32), found by the sanitizer run below."""
    for w, lo, hi in ((0x7FFFD5ED, -10771, -32768), (0x7FFF8000, -32768, -32768), (0x80007FFF, 32767, -32768),
                      (0xFFFFFFFF, -1, 0), (0x00010000, 0, 1), (0x7FFF7FFF, 32767, 32767)):
        assert emu.emu_fp_from_key_word(C.c_uint32(w), 0) == float(lo)
        assert emu.emu_fp_from_key_word(C.c_uint32(w), 1) == float(hi)
        assert (lo + (hi << 16)) & 0xFFFFFFFF == w


def test_device_headers_under_address_and_ub_sanitizers():
    """tests/emu/sanitize_main.cpp: every shipped shape, field and exchange-buffer scheme through
    bsk_prepare / external product / blind rotation / keygen with ASan + UBSan (CPU only: the GPU
    pool has no sanitizer).  LDS is an exact-size heap buffer there, so a bad slot or twiddle index
    is an error, and so is signed overflow or an out-of-range shift."""
    import conftest
    exe = conftest.sanitizer_binary() or conftest.sanitizer_binary()  # first call may only start the build; the second joins it
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "sanitized run clean" in res.stdout

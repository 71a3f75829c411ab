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


@pytest.fixture(scope="session")
def emu():
    so = os.path.join(EMU_DIR, "libtfhe_emu.so")
    srcs = [os.path.join(EMU_DIR, "emu.cpp")] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I", CSRC,
                        "-o", so, os.path.join(EMU_DIR, "emu.cpp")], check=True)
    lib = C.CDLL(so)
    for f in ("emu_gl_mul", "emu_gl_add", "emu_gl_sub", "emu_gl_from_i32"):
        getattr(lib, f).restype = C.c_uint64
    lib.emu_gl_lift.restype = C.c_uint32
    return lib


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
    tw = np.zeros(n, dtype=np.uint64)
    emu.emu_twiddles(logn, p64(tw))
    assert tw.tolist() == fwd
    rng = np.random.default_rng(logn)
    a = np.array([int(x) % P for x in rng.integers(0, 1 << 64, size=n, dtype=np.uint64)], dtype=np.uint64)
    out = np.zeros_like(a)
    assert emu.emu_poly_ntt(logn, p64(a), p64(out), 0) == 0
    ref = ntt_model.ntt_ref(a.tolist(), logn, fwd)
    assert out.tolist() == ref
    back = np.zeros_like(a)
    assert emu.emu_poly_ntt(logn, p64(out), p64(back), 1) == 0
    assert back.tolist() == [x * n % P for x in a.tolist()]  # unscaled inverse


def prepared(emu, params, bsk):
    flat = np.ascontiguousarray(bsk, dtype=np.uint32).reshape(-1, params.N)
    out = np.zeros(flat.shape, dtype=np.uint64)
    assert emu.emu_bsk_prepare(params.glwe_poly_degree, C.c_size_t(flat.shape[0]), p32(flat), p64(out)) == 0
    return out


CASES = [
    # k, logN, n, (logB, levels), log_p
    (1, 9, 3, (8, 2), 2),
    (1, 10, 3, (7, 3), 2),   # BASELINE cfg2 shape, misaligned base (bits 28..31 dropped)
    (2, 9, 2, (4, 6), 2),    # reference default shape
    (2, 11, 1, (8, 4), 4),   # BASELINE cfg5 shape
]


@pytest.mark.parametrize("k,logn,n,pbs,log_p", CASES)
def test_external_product_vs_oracle(emu, oracle, k, logn, n, pbs, log_p):
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    rng = np.random.default_rng(11 * logn + k)
    ggsw = rng.integers(0, 1 << 32, size=(params.R, k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    glwe = rng.integers(0, 1 << 32, size=(k + 1, params.N), dtype=np.uint64).astype(np.uint32)
    # hit the digit == B and digit == -B/2 paths in every polynomial
    glwe[:, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
    spec = prepared(emu, params, ggsw)
    out = np.zeros_like(glwe)
    assert emu.emu_external_product(k, logn, pbs[0], pbs[1], p64(spec), p32(glwe), p32(out)) == 0
    assert np.array_equal(out, oracle.external_product(params, ggsw, glwe))


@pytest.mark.parametrize("k,logn,n,pbs,log_p", CASES)
def test_blind_rotate_and_extract_vs_oracle(emu, oracle, k, logn, n, pbs, log_p):
    params = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    batch = 2
    lut = np.random.default_rng(5).integers(0, 1 << log_p, size=1 << log_p)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(params, batch, cfg_index=40 + logn, lut=lut)
    lwe = lwe.copy()
    lwe[0, 0] = 0            # a~ = 0: the skipped iteration
    lwe[1, n] = 0xFFFFFFFF   # b~ rounds up to 2N and wraps to 0
    spec = prepared(emu, params, bsk)
    glwe = np.zeros((batch, k + 1, params.N), dtype=np.uint32)
    ext = np.zeros((batch, params.big_n + 1), dtype=np.uint32)
    rc = emu.emu_blind_rotate(n, k, logn, log_p, 1, pbs[0], pbs[1], C.c_size_t(batch), p32(lwe), p32(tv),
                              C.c_size_t(0), p64(spec), p32(glwe), p32(ext))
    assert rc == 0
    for b in range(batch):
        _, tr = oracle.bootstrap(params, lwe[b], bsk, ksk, tv, trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"sample {b}"
        assert np.array_equal(ext[b], tr["extracted_lwe"]), f"sample {b}"

"""ctypes/numpy front-end of the CPU ORACLE (oracle/tfhe_oracle.{h,c}).

TEST INFRASTRUCTURE ONLY.  May be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the shipped package (tfhe-research_amd/).  See the
header of tfhe_oracle.h for what the oracle restates (reference file:line per function) and for
its pinning status ("parity unpinned" by reference vectors: the reference holds none).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field, replace

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libtfhe_oracle.so")

u32p = C.POINTER(C.c_uint32)


class CDecomposer(C.Structure):
    _fields_ = [("log_base", C.c_uint32), ("levels", C.c_uint32), ("log_q", C.c_uint32)]


class CParams(C.Structure):
    _fields_ = [
        ("glwe_dimension", C.c_uint32),
        ("glwe_poly_degree", C.c_uint32),
        ("lwe_dimension", C.c_uint32),
        ("padding_bits", C.c_uint32),
        ("log_p", C.c_uint32),
        ("log_q", C.c_uint32),
        ("ks_decomposer", CDecomposer),
        ("pbs_decomposer", CDecomposer),
        ("lwe_std_dev", C.c_double),
        ("glwe_std_dev", C.c_double),
    ]


class CTrace(C.Structure):
    _fields_ = [
        ("approximate_lwe", u32p),
        ("acc_init", u32p),
        ("acc_after_each", u32p),
        ("acc_final", u32p),
        ("extracted_lwe", u32p),
    ]


class CRng(C.Structure):
    _fields_ = [("state", C.c_uint64), ("literal_noise", C.c_int), ("have_spare", C.c_int),
                ("spare", C.c_double)]


@dataclass(frozen=True)
class Decomposer:
    """decomposer.rs:2-6"""
    log_base: int
    levels: int
    log_q: int = 32

    def to_c(self) -> CDecomposer:
        return CDecomposer(self.log_base, self.levels, self.log_q)


@dataclass(frozen=True)
class Params:
    """lib.rs:23-34.  glwe_poly_degree is log2(N), as in the reference (lib.rs:40,60)."""
    glwe_dimension: int
    glwe_poly_degree: int
    lwe_dimension: int
    pbs: Decomposer
    ks: Decomposer = field(default_factory=lambda: Decomposer(4, 5))
    log_p: int = 2
    padding_bits: int = 1
    log_q: int = 32
    lwe_std_dev: float = 0.000013071021089943935
    glwe_std_dev: float = 0.00000004990272175010415

    @property
    def N(self) -> int:
        return 1 << self.glwe_poly_degree

    @property
    def k(self) -> int:
        return self.glwe_dimension

    @property
    def n(self) -> int:
        return self.lwe_dimension

    @property
    def R(self) -> int:
        return (self.k + 1) * self.pbs.levels

    @property
    def big_n(self) -> int:
        """post-PBS LWE dimension, lib.rs:60"""
        return self.N * self.k

    def bsk_shape(self):
        return (self.n, self.R, self.k + 1, self.N)

    def ksk_shape(self):
        return (self.big_n * self.ks.levels, self.n + 1)

    def to_c(self) -> CParams:
        return CParams(self.glwe_dimension, self.glwe_poly_degree, self.lwe_dimension,
                       self.padding_bits, self.log_p, self.log_q, self.ks.to_c(), self.pbs.to_c(),
                       self.lwe_std_dev, self.glwe_std_dev)

    def with_n(self, n: int) -> "Params":
        return replace(self, lwe_dimension=n)


# The reference's two parameter sets (lib.rs:77-99 cfg(test), lib.rs:101-123).
REF_TEST = Params(2, 9, 4, Decomposer(4, 6))
REF_DEFAULT = Params(2, 9, 722, Decomposer(4, 6))
# BASELINE.json configs with the unspecified fields fixed as in SURVEY.md section 8(d).
CFG1 = Params(1, 9, 500, Decomposer(8, 2))
CFG2 = Params(1, 10, 630, Decomposer(7, 3))
CFG3 = REF_DEFAULT
CFG4 = CFG2
CFG5 = Params(2, 11, 630, Decomposer(8, 4), log_p=4)
CONFIGS = {"cfg1": CFG1, "cfg2": CFG2, "cfg3": CFG3, "cfg4": CFG4, "cfg5": CFG5}


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    srcs = [os.path.join(_HERE, f) for f in ("tfhe_oracle.c", "tfhe_oracle_crypto.c", "tfhe_oracle.h")]
    stale = force or not os.path.exists(_SO) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "all"], check=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_round_value.restype = C.c_uint32
        _lib.orc_recompose.restype = C.c_uint32
        _lib.orc_integer_division.restype = C.c_uint32
        _lib.orc_decrypt_lwe.restype = C.c_uint32
        _lib.orc_sample_gaussian.restype = C.c_uint32
        _lib.orc_gadget_shift.restype = C.c_uint32
        _lib.orc_lwe_decode.restype = C.c_uint32
        _lib.orc_rng_next_u64.restype = C.c_uint64
        _lib.orc_rng_next_u32.restype = C.c_uint32
        for name in ("orc_ggsw_words", "orc_bsk_words", "orc_ksk_words"):
            getattr(_lib, name).restype = C.c_size_t
    return _lib


def _a(x) -> np.ndarray:
    return np.ascontiguousarray(x, dtype=np.uint32)


def _p(x: np.ndarray):
    return x.ctypes.data_as(u32p)


def set_poly_mul_mode(mode: int) -> None:
    """0 = literal Toeplitz matrix + mat-vec (utils.rs:155-160); 1 = schoolbook (utils.rs:221-236)."""
    lib().orc_set_poly_mul_mode(C.c_int(mode))


def set_decomposer_aligned(aligned: bool) -> None:
    """False = the reference's literal decomposer; True = the aligned extension (tfhe_oracle.h)."""
    lib().orc_set_decomposer_aligned(C.c_int(int(aligned)))


class decomposer_aligned:
    """with oracle.decomposer_aligned(True): ...  (restores the previous mode)"""

    def __init__(self, aligned: bool):
        self.aligned = aligned

    def __enter__(self):
        self.prev = lib().orc_get_decomposer_aligned()
        set_decomposer_aligned(self.aligned)

    def __exit__(self, *exc):
        set_decomposer_aligned(bool(self.prev))


def gadget_shift(dec: Decomposer, level: int) -> int:
    cd = dec.to_c()
    return int(lib().orc_gadget_shift(C.byref(cd), C.c_uint32(level)))


def bootstrap_ks_first(params: Params, lwe_big, bsk, ksk, test_vector_poly) -> np.ndarray:
    """The swapped order of notes/TFHE.md:367-400 composed from the reference's own steps:
    key_switch_lwe (key_switching.rs:63-103) -> blind rotation (bootstrapping.rs:67-105) ->
    sample_extract (bootstrapping.rs:122-156).  [k*N+1] -> [k*N+1]."""
    small = key_switch_lwe(lwe_big, params.big_n, params.n, params.ks, ksk)
    return sample_extract(params, blind_rotate(params, small, bsk, test_vector_poly), 0)


def validate(params: Params) -> int:
    cp = params.to_c()
    return lib().orc_params_validate(C.byref(cp))


# ---------------------------------------------------------------- decomposer.rs
def round_value(dec: Decomposer, values) -> np.ndarray:
    v = _a(values).ravel()
    d = dec.to_c()
    return np.array([lib().orc_round_value(C.byref(d), C.c_uint32(int(x))) for x in v], dtype=np.uint32)


def decompose(dec: Decomposer, values) -> np.ndarray:
    """-> (len(values), levels), MSB first (decomposer.rs:42-80)."""
    v = _a(values).ravel()
    out = np.zeros((v.size, dec.levels), dtype=np.uint32)
    # decompose_poly writes a (levels x n) matrix; use it for speed, then transpose
    tmp = np.zeros((dec.levels, v.size), dtype=np.uint32)
    d = dec.to_c()
    rc = lib().orc_decompose_poly(_p(v), C.c_size_t(v.size), C.byref(d), _p(tmp))
    if rc:
        raise ValueError("invalid decomposer parameters")
    out[:] = tmp.T
    return out


def recompose(dec: Decomposer, legs) -> int:
    legs = _a(legs)
    d = dec.to_c()
    return int(lib().orc_recompose(C.byref(d), _p(legs)))


# ---------------------------------------------------------------- utils.rs
def switch_modulus(values, log_from: int, log_to: int) -> np.ndarray:
    v = _a(values).ravel()
    out = np.zeros_like(v)
    lib().orc_switch_modulus(_p(v), C.c_size_t(v.size), C.c_uint32(log_from), C.c_uint32(log_to), _p(out))
    return out


def poly_mul(p0, p1) -> np.ndarray:
    p0, p1 = _a(p0), _a(p1)
    out = np.zeros_like(p0)
    lib().orc_poly_mul(_p(p0), _p(p1), C.c_size_t(p0.size), _p(out))
    return out


def school_book_negacylic_mul(p0, p1) -> np.ndarray:
    p0, p1 = _a(p0), _a(p1)
    out = np.zeros_like(p0)
    lib().orc_school_book_negacylic_mul(_p(p0), _p(p1), C.c_size_t(p0.size), _p(out))
    return out


def poly_mul_monomial(p0, monomial_index: int) -> np.ndarray:
    p0 = _a(p0)
    out = np.zeros_like(p0)
    lib().orc_poly_mul_monomial(_p(p0), C.c_size_t(p0.size), C.c_int64(monomial_index), _p(out))
    return out


# ---------------------------------------------------------------- glwe.rs / ggsw.rs
def glwe_mul_monomial(glwe, monomial_index: int) -> np.ndarray:
    g = _a(glwe)
    out = np.zeros_like(g)
    lib().orc_glwe_mul_monomial(_p(g), C.c_size_t(g.shape[0]), C.c_size_t(g.shape[1]),
                                C.c_int64(monomial_index), _p(out))
    return out


def decompose_glwe_ciphertext(glwe, dec: Decomposer) -> np.ndarray:
    g = _a(glwe)
    out = np.zeros((g.shape[0] * dec.levels, g.shape[1]), dtype=np.uint32)
    d = dec.to_c()
    rc = lib().orc_decompose_glwe_ciphertext(_p(g), C.c_size_t(g.shape[0]), C.c_size_t(g.shape[1]),
                                             C.byref(d), _p(out))
    if rc:
        raise ValueError("invalid decomposer parameters")
    return out


def external_product(params: Params, ggsw, glwe) -> np.ndarray:
    ggsw, glwe = _a(ggsw), _a(glwe)
    assert ggsw.shape == (params.R, params.k + 1, params.N) and glwe.shape == (params.k + 1, params.N)
    out = np.zeros_like(glwe)
    cp = params.to_c()
    rc = lib().orc_external_product(C.byref(cp), _p(ggsw), _p(glwe), _p(out))
    if rc:
        raise ValueError("external_product failed")
    return out


def cmux(params: Params, ggsw, ct0, ct1):
    """-> (result, ct1 clobbered with ct1 - ct0), as ggsw.rs:164-178."""
    ggsw, ct0 = _a(ggsw), _a(ct0)
    ct1 = _a(ct1).copy()
    out = np.zeros_like(ct0)
    cp = params.to_c()
    rc = lib().orc_cmux(C.byref(cp), _p(ggsw), _p(ct0), _p(ct1), _p(out))
    if rc:
        raise ValueError("cmux failed")
    return out, ct1


# ---------------------------------------------------------------- bootstrapping.rs / key_switching.rs
def sample_extract(params: Params, glwe, sample_index: int = 0) -> np.ndarray:
    g = _a(glwe)
    out = np.zeros(params.big_n + 1, dtype=np.uint32)
    cp = params.to_c()
    rc = lib().orc_sample_extract(C.byref(cp), _p(g), C.c_size_t(sample_index), _p(out))
    if rc:
        raise ValueError("sample_index out of range")
    return out


def key_switch_lwe(lwe, from_n: int, to_n: int, dec: Decomposer, ksk) -> np.ndarray:
    lwe, ksk = _a(lwe), _a(ksk)
    assert lwe.size == from_n + 1 and ksk.shape == (from_n * dec.levels, to_n + 1)
    out = np.zeros(to_n + 1, dtype=np.uint32)
    d = dec.to_c()
    rc = lib().orc_key_switch_lwe(_p(lwe), C.c_size_t(from_n), C.c_size_t(to_n), C.byref(d), _p(ksk), _p(out))
    if rc:
        raise ValueError("key_switch_lwe failed")
    return out


def _trace_buffers(params: Params, each: bool):
    bufs = {
        "approximate_lwe": np.zeros(params.n + 1, dtype=np.uint32),
        "acc_init": np.zeros((params.k + 1, params.N), dtype=np.uint32),
        "acc_final": np.zeros((params.k + 1, params.N), dtype=np.uint32),
        "extracted_lwe": np.zeros(params.big_n + 1, dtype=np.uint32),
    }
    if each:
        bufs["acc_after_each"] = np.zeros((params.n, params.k + 1, params.N), dtype=np.uint32)
    t = CTrace(_p(bufs["approximate_lwe"]), _p(bufs["acc_init"]),
               _p(bufs["acc_after_each"]) if each else None, _p(bufs["acc_final"]),
               _p(bufs["extracted_lwe"]))
    return bufs, t


def bootstrap(params: Params, lwe_ct, bsk, ksk, test_vector_poly, trace: bool = False,
              trace_each: bool = False):
    """bootstrapping.rs:58-120 for ONE ciphertext.  -> out or (out, trace dict)."""
    lwe_ct, bsk, ksk, tv = _a(lwe_ct), _a(bsk), _a(ksk), _a(test_vector_poly)
    assert lwe_ct.size == params.n + 1 and bsk.shape == params.bsk_shape()
    assert ksk.shape == params.ksk_shape() and tv.size == params.N
    out = np.zeros(params.n + 1, dtype=np.uint32)
    cp = params.to_c()
    bufs, t = (None, None)
    if trace:
        bufs, t = _trace_buffers(params, trace_each)
    rc = lib().orc_bootstrap(C.byref(cp), _p(lwe_ct), _p(bsk), _p(ksk), _p(tv), _p(out),
                             C.byref(t) if t is not None else None)
    if rc:
        raise ValueError(f"bootstrap failed rc={rc}")
    return (out, bufs) if trace else out


def blind_rotate(params: Params, lwe_ct, bsk, test_vector_poly) -> np.ndarray:
    lwe_ct, bsk, tv = _a(lwe_ct), _a(bsk), _a(test_vector_poly)
    assert bsk.shape == params.bsk_shape()
    acc = np.zeros((params.k + 1, params.N), dtype=np.uint32)
    cp = params.to_c()
    rc = lib().orc_blind_rotate(C.byref(cp), _p(lwe_ct), _p(bsk), _p(tv), _p(acc), None)
    if rc:
        raise ValueError("blind_rotate failed")
    return acc


# ---------------------------------------------------------------- notes/BMMP Bootstrapping.md
def bmmp_messages(lwe_sk) -> np.ndarray:
    """The GGSW messages of the unrolled blind rotation, three per pair of key bits
    (notes/BMMP Bootstrapping.md:22-24): s s', s (1 - s'), s' (1 - s)."""
    sk = _a(lwe_sk).astype(np.int64)
    assert sk.size % 2 == 0
    s0, s1 = sk[0::2], sk[1::2]
    return np.stack([s0 * s1, s0 * (1 - s1), s1 * (1 - s0)], axis=1).reshape(-1).astype(np.uint32)


def bmmp_bsk_shape(params: Params):
    return (params.n // 2 * 3, params.R, params.k + 1, params.N)


def blind_rotate_bmmp(params: Params, lwe_ct, bsk_bmmp, test_vector_poly) -> np.ndarray:
    """notes/BMMP Bootstrapping.md:13-25 composed from the reference's own steps: with
    X^{a s + a' s'} = s s' (X^{a+a'} - 1) + s (1 - s') (X^a - 1) + (1 - s) s' (X^{a'} - 1) + 1
    one step is acc += sum_m (X^{e_m} - 1) * external_product(bk_{3j+m}, acc), e = (a+a', a, a');
    external_product = ggsw.rs:132-161, the monomial products = glwe.rs:20-34, all sums wrapping u32.
    bsk_bmmp [n/2*3][R][k+1][N]."""
    lwe_ct, bsk, tv = _a(lwe_ct), _a(bsk_bmmp), _a(test_vector_poly)
    assert params.n % 2 == 0 and bsk.shape == bmmp_bsk_shape(params)
    two_n = 2 * params.N
    a = switch_modulus(lwe_ct, 32, params.glwe_poly_degree + 1)
    acc = np.zeros((params.k + 1, params.N), dtype=np.uint32)
    acc[params.k] = (tv.astype(np.uint64) << np.uint64(32 - params.log_p - params.padding_bits)).astype(np.uint32)
    acc = glwe_mul_monomial(acc, -int(a[params.n]))
    for j in range(params.n // 2):
        e = ((int(a[2 * j]) + int(a[2 * j + 1])) % two_n, int(a[2 * j]), int(a[2 * j + 1]))
        prods = [external_product(params, bsk[3 * j + m], acc) for m in range(3)]
        for m in range(3):
            acc = (acc + glwe_mul_monomial(prods[m], e[m]) - prods[m]).astype(np.uint32)
    return acc


def bootstrap_bmmp(params: Params, lwe_ct, bsk_bmmp, ksk, test_vector_poly) -> np.ndarray:
    """bootstrap() (bootstrapping.rs:58-120) with the unrolled blind rotation in place of the loop
    :79-105: mod switch -> blind_rotate_bmmp -> sample_extract -> key_switch_lwe."""
    acc = blind_rotate_bmmp(params, lwe_ct, bsk_bmmp, test_vector_poly)
    return key_switch_lwe(sample_extract(params, acc, 0), params.big_n, params.n, params.ks, _a(ksk))


def keygen_bmmp(params: Params, rng: "Rng"):
    """-> (lwe_sk, glwe_sk, bsk_bmmp [n/2*3][R][k+1][N], ksk): bootstrapping_key_gen
    (bootstrapping.rs:23-56) with the three product messages per key-bit pair in place of the bits."""
    cp = params.to_c()
    lwe_sk = np.zeros(params.n, dtype=np.uint32)
    glwe_sk = np.zeros((params.k, params.N), dtype=np.uint32)
    lib().orc_lwe_secret_key_random(C.byref(cp), C.byref(rng.c), _p(lwe_sk))
    lib().orc_glwe_secret_key_random(C.byref(cp), C.byref(rng.c), _p(glwe_sk))
    bsk = np.stack([encrypt_ggsw(params, glwe_sk, int(m), rng) for m in bmmp_messages(lwe_sk)])
    # the key switching key is the ordinary one: any valid KSK from the flattened GLWE key to lwe_sk
    samples = rng.uniform_u32(params.ksk_shape())
    samples[:, params.n] = sample_gaussian(rng, params.lwe_std_dev, samples.shape[0])
    ksk = generate_ksk_from_samples(glwe_sk.reshape(-1), lwe_sk, params.ks, samples)
    return lwe_sk, glwe_sk, bsk, ksk


# ---------------------------------------------------------------- test_vector.rs / boolean.rs
def construct_test_from_lut(params: Params, lut) -> np.ndarray:
    lut = _a(lut)
    out = np.zeros(params.N, dtype=np.uint32)
    cp = params.to_c()
    rc = lib().orc_construct_test_from_lut(C.byref(cp), _p(lut), C.c_size_t(lut.size), _p(out))
    if rc:
        raise ValueError("lut must hold 2^log_p entries")
    return out


def construct_identity_test_vector(params: Params) -> np.ndarray:
    return construct_test_from_lut(params, np.arange(1 << params.log_p, dtype=np.uint32))


def construct_test_vector_boolean(params: Params, f) -> np.ndarray:
    """f(lhs, rhs) -> bit, the closure of test_vector.rs:5-20."""
    lut = [f((i >> 1) & 1, i & 1) for i in range(1 << params.log_p)]
    return construct_test_from_lut(params, lut)


def boolean_gate(params: Params, f, ct0, ct1, bsk, ksk) -> np.ndarray:
    """boolean.rs:9-30 with an arbitrary closure: bootstrap(2*ct1 + ct0)."""
    tv = construct_test_vector_boolean(params, f)
    ct_in = (_a(ct1) * np.uint32(2) + _a(ct0)).astype(np.uint32)
    return bootstrap(params, ct_in, bsk, ksk, tv)


# ---------------------------------------------------------------- host-side crypto
class Rng:
    def __init__(self, seed: int, literal_noise: bool = False):
        self.c = CRng()
        lib().orc_rng_seed(C.byref(self.c), C.c_uint64(seed))
        self.c.literal_noise = int(literal_noise)

    def uniform_u32(self, shape) -> np.ndarray:
        out = np.zeros(shape, dtype=np.uint32)
        lib().orc_fill_uniform_u32(C.byref(self.c), _p(out), C.c_size_t(out.size))
        return out


def keygen(params: Params, rng: Rng):
    """-> (lwe_sk[n], glwe_sk[k][N], bsk, ksk)  (bootstrapping.rs:23-56)."""
    cp = params.to_c()
    lwe_sk = np.zeros(params.n, dtype=np.uint32)
    glwe_sk = np.zeros((params.k, params.N), dtype=np.uint32)
    lib().orc_lwe_secret_key_random(C.byref(cp), C.byref(rng.c), _p(lwe_sk))
    lib().orc_glwe_secret_key_random(C.byref(cp), C.byref(rng.c), _p(glwe_sk))
    bsk = np.zeros(params.bsk_shape(), dtype=np.uint32)
    ksk = np.zeros(params.ksk_shape(), dtype=np.uint32)
    prev = lib().orc_get_poly_mul_mode()
    set_poly_mul_mode(1)
    lib().orc_bootstrapping_key_gen(C.byref(cp), _p(lwe_sk), _p(glwe_sk), C.byref(rng.c), _p(bsk), _p(ksk))
    set_poly_mul_mode(prev)
    return lwe_sk, glwe_sk, bsk, ksk


def encrypt_lwe(params: Params, sk, message: int, rng: Rng) -> np.ndarray:
    sk = _a(sk)
    cp = params.to_c()
    pt = C.c_uint32()
    if lib().orc_lwe_encode(C.byref(cp), C.c_uint32(message), C.byref(pt)):
        raise ValueError("message out of range")
    ct = np.zeros(sk.size + 1, dtype=np.uint32)
    lib().orc_encrypt_lwe_plaintext(C.c_size_t(sk.size), C.c_double(params.lwe_std_dev), _p(sk),
                                    pt, C.byref(rng.c), _p(ct))
    return ct


def decrypt_lwe_raw(sk, ct) -> int:
    sk, ct = _a(sk), _a(ct)
    return int(lib().orc_decrypt_lwe(C.c_size_t(sk.size), _p(sk), _p(ct)))


def decrypt_lwe_message(params: Params, sk, ct) -> int:
    """decrypt, round to the nearest message slot, reduce mod 2^log_p."""
    raw = decrypt_lwe_raw(sk, ct)
    shift = params.log_q - (params.log_p + params.padding_bits)
    return (((raw + (1 << (shift - 1))) & 0xFFFFFFFF) >> shift) & ((1 << params.log_p) - 1)


def encrypt_glwe(params: Params, sk, message, rng: Rng) -> np.ndarray:
    sk, msg = _a(sk), _a(message)
    cp = params.to_c()
    pt = np.zeros(params.N, dtype=np.uint32)
    if lib().orc_glwe_encode_message(C.byref(cp), _p(msg), C.c_size_t(msg.size), _p(pt)):
        raise ValueError("message out of range")
    ct = np.zeros((params.k + 1, params.N), dtype=np.uint32)
    prev = lib().orc_get_poly_mul_mode()
    set_poly_mul_mode(1)
    lib().orc_encrypt_glwe_plaintext(C.byref(cp), _p(pt), _p(sk), C.byref(rng.c), _p(ct))
    set_poly_mul_mode(prev)
    return ct


def decrypt_glwe_raw(params: Params, sk, ct) -> np.ndarray:
    sk, ct = _a(sk), _a(ct)
    cp = params.to_c()
    pt = np.zeros(params.N, dtype=np.uint32)
    prev = lib().orc_get_poly_mul_mode()
    set_poly_mul_mode(1)
    lib().orc_decrypt_glwe_ciphertext(C.byref(cp), _p(sk), _p(ct), _p(pt))
    set_poly_mul_mode(prev)
    return pt


def encrypt_ggsw(params: Params, glwe_sk, message: int, rng: Rng) -> np.ndarray:
    sk = _a(glwe_sk)
    cp = params.to_c()
    out = np.zeros((params.R, params.k + 1, params.N), dtype=np.uint32)
    prev = lib().orc_get_poly_mul_mode()
    set_poly_mul_mode(1)
    lib().orc_encrypt_ggsw_plaintext(C.byref(cp), C.c_uint32(message), _p(sk), C.byref(rng.c), _p(out))
    set_poly_mul_mode(prev)
    return out


# ---- the same with the random draws hoisted out (buffers arrive pre-filled with masks + errors)
def _schoolbook(fn):
    prev = lib().orc_get_poly_mul_mode()
    set_poly_mul_mode(1)
    try:
        fn()
    finally:
        set_poly_mul_mode(prev)


def encrypt_lwe_from_samples(sk, samples, plaintexts=None) -> np.ndarray:
    """samples [B][n+1]: uniform masks + error in the b slot -> ciphertexts (lwe.rs:138-160)."""
    sk = _a(sk)
    out = _a(samples).copy().reshape(-1, sk.size + 1)
    for i in range(out.shape[0]):
        pt = 0 if plaintexts is None else int(plaintexts[i])
        lib().orc_encrypt_lwe_from_samples(C.c_size_t(sk.size), _p(sk), C.c_uint32(pt), _p(out[i]))
    return out


def encrypt_glwe_zero_from_samples(params: Params, sk, samples) -> np.ndarray:
    """samples [count][k+1][N]: uniform masks + errors in the body (glwe.rs:190-209)."""
    sk = _a(sk)
    cp = params.to_c()
    out = _a(samples).copy().reshape(-1, params.k + 1, params.N)
    _schoolbook(lambda: [lib().orc_encrypt_glwe_zero_from_samples(C.byref(cp), _p(sk), _p(out[i]))
                         for i in range(out.shape[0])])
    return out


def encrypt_ggsw_from_samples(params: Params, glwe_sk, messages, samples) -> np.ndarray:
    """samples [count][R][k+1][N] pre-filled row by row (ggsw.rs:76-130)."""
    sk = _a(glwe_sk)
    cp = params.to_c()
    out = _a(samples).copy().reshape(-1, params.R, params.k + 1, params.N)
    _schoolbook(lambda: [lib().orc_encrypt_ggsw_from_samples(C.byref(cp), C.c_uint32(int(messages[i])),
                                                             _p(sk), _p(out[i]))
                         for i in range(out.shape[0])])
    return out


def generate_ksk_from_samples(from_sk, to_sk, dec: Decomposer, samples) -> np.ndarray:
    """samples [from_n*levels][to_n+1] pre-filled (key_switching.rs:20-60)."""
    from_sk, to_sk = _a(from_sk).reshape(-1), _a(to_sk)
    out = _a(samples).copy()
    cd = dec.to_c()
    lib().orc_generate_ksk_from_samples(_p(from_sk), C.c_size_t(from_sk.size), _p(to_sk),
                                        C.c_size_t(to_sk.size), C.byref(cd), _p(out))
    return out


def bootstrapping_key_gen_from_samples(params: Params, lwe_sk, glwe_sk, bsk_samples, ksk_samples):
    """bootstrapping.rs:23-56 on pre-filled bsk / ksk buffers -> (bsk, ksk)."""
    lwe_sk, glwe_sk = _a(lwe_sk), _a(glwe_sk)
    cp = params.to_c()
    bsk, ksk = _a(bsk_samples).copy(), _a(ksk_samples).copy()
    _schoolbook(lambda: lib().orc_bootstrapping_key_gen_from_samples(C.byref(cp), _p(lwe_sk), _p(glwe_sk),
                                                                    _p(bsk), _p(ksk)))
    return bsk, ksk


def sample_binary(rng: Rng, shape) -> np.ndarray:
    out = np.zeros(shape, dtype=np.uint32)
    lib().orc_sample_binary(C.byref(rng.c), _p(out), C.c_size_t(out.size))
    return out


def sample_gaussian(rng: Rng, std_dev: float, shape) -> np.ndarray:
    out = np.zeros(shape, dtype=np.uint32)
    flat = out.reshape(-1)
    fn = lib().orc_sample_gaussian
    for i in range(flat.size):
        flat[i] = fn(C.byref(rng.c), C.c_double(std_dev))
    return out


# ---------------------------------------------------------------- synthetic inputs
def splitmix64_u32(seed: int, count: int) -> np.ndarray:
    """`count` uniform u32 words: the high halves of a SplitMix64 stream (vectorised)."""
    idx = np.arange(1, count + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(32)).astype(np.uint32)


SYNTH_SEED = 0x7466686500000000


def synthetic_inputs(params: Params, batch: int, cfg_index: int = 2, lut=None):
    """Deterministic uniform-u32 LWE / BSK / KSK words (the arithmetic is total, so parity and
    throughput do not need valid encryptions) and a test vector built from `lut` (identity LUT
    by default) through construct_test_from_lut."""
    seed = SYNTH_SEED + cfg_index
    n_lwe = batch * (params.n + 1)
    n_bsk = int(np.prod(params.bsk_shape()))
    n_ksk = int(np.prod(params.ksk_shape()))
    words = splitmix64_u32(seed, n_lwe + n_bsk + n_ksk)
    lwe = words[:n_lwe].reshape(batch, params.n + 1)
    bsk = words[n_lwe:n_lwe + n_bsk].reshape(params.bsk_shape())
    ksk = words[n_lwe + n_bsk:].reshape(params.ksk_shape())
    tv = (construct_identity_test_vector(params) if lut is None
          else construct_test_from_lut(params, lut))
    return lwe, bsk, ksk, tv

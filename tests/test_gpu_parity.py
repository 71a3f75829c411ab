"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI, against
the CPU oracle on the same seeded inputs -- bit-exact, all arithmetic is wrapping u32."""
import numpy as np
import pytest

from gpu_common import pkg, rand_u32, to_pkg_params

pytestmark = pytest.mark.gpu

SHAPES = [
    # name, k, logN, n, pbs(logB, l), ks(logB, l), log_p
    ("ref_test", 2, 9, 4, (4, 6), (4, 5), 2),      # reference cfg(test) parameters (lib.rs:77-99)
    ("cfg1_small", 1, 9, 6, (8, 2), (4, 5), 2),
    ("cfg2_small", 1, 10, 5, (7, 3), (4, 5), 2),   # misaligned base: literal decomposer quirk
    ("cfg5_small", 2, 11, 2, (8, 4), (4, 5), 4),
    ("k2_n1024", 2, 10, 3, (4, 7), (8, 3), 3),
    ("k1_n2048", 1, 11, 2, (16, 2), (2, 9), 2),
    ("k1_n2048_b8", 1, 11, 3, (8, 3), (4, 5), 2),  # N = 2048 with k = 1 inside the complex transform's bound: 8-wave teams, two samples each
]


def make(oracle, k, logn, n, pbs, ks, log_p):
    return oracle.Params(k, logn, n, oracle.Decomposer(*pbs), oracle.Decomposer(*ks), log_p=log_p)


BACKENDS = ["fp64", "goldilocks", "goldilocks-split", "fp64-p49", "fp64-fft"]


def backend_id(name):
    m = pkg()
    return {"fp64": m.BACKEND_FP64, "goldilocks": m.BACKEND_GOLDILOCKS, "auto": m.BACKEND_AUTO,
            "goldilocks-split": m.BACKEND_GOLDILOCKS_SPLIT, "fp64-p49": m.BACKEND_FP64_P49,
            "fp64-fft": m.BACKEND_FP64_FFT}[name]


def fp64_exact(p):
    """the fp64 backend's exactness bound (tfhe_hip.h): (k+1) l N B 2^15 < 2^40.9"""
    return np.log2(p.R) + p.glwe_poly_degree + p.pbs.log_base + 15 < 40.9 and p.pbs.log_base <= 9


def fp49_exact(p):
    """the 49-bit backend's bound: (k+1) l N B 2^31 < 2^48.25 and at most 20 digit rows"""
    return np.log2(p.R) + p.glwe_poly_degree + p.pbs.log_base + 31 < 48.25 and p.R <= 20


@pytest.fixture(scope="module")
def contexts(oracle):
    made = {}

    def get(name, backend="auto"):
        key = (name, backend)
        if key not in made:
            spec = next(s for s in SHAPES if s[0] == name)
            p = make(oracle, *spec[1:])
            batch = 9  # not a multiple of the waves per workgroup: exercises the ragged tail
            lut = np.random.default_rng(len(name)).integers(0, 1 << p.log_p, size=1 << p.log_p)
            lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, batch, cfg_index=70 + len(name), lut=lut)
            lwe = lwe.copy()
            lwe[0, 0] = 0             # a~ = 0 -> skipped CMUX
            lwe[1, p.n] = 0xFFFFFFFF  # b~ rounds to 2N and wraps to 0
            lwe[2, :] = 0x80000000
            if backend == "fp64" and not fp64_exact(p):
                pytest.skip("outside the fp64 backend's exactness bound")
            if backend == "fp64-p49" and not fp49_exact(p):
                pytest.skip("outside the 49-bit backend's exactness bound")
            try:
                ctx = pkg().Context(to_pkg_params(p), backend=backend_id(backend))
            except pkg().TfheError as e:
                # the complex-FFT backend: kernels at N = 512 and 1024, admitted below its rounding-error bound
                if backend == "fp64-fft" and e.status in (pkg().TFHE_ERR_UNSUPPORTED, pkg().TFHE_ERR_EXACTNESS):
                    pytest.skip("the fp64-fft backend does not cover this shape")
                raise
            ctx.load_bootstrapping_key(bsk, ksk)
            made[key] = (p, ctx, lwe, bsk, ksk, tv)
        return made[key]

    yield get
    for _, ctx, *_ in made.values():
        ctx.close()


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", [s[0] for s in SHAPES])
def test_bootstrap_matches_oracle(oracle, contexts, name, backend):
    p, ctx, lwe, bsk, ksk, tv = contexts(name, backend)
    assert ctx.backend == {"fp64": "fp64-p42"}.get(backend, backend)
    got = ctx.bootstrap(lwe, tv)
    glwe = ctx.blind_rotate(lwe, tv)
    for b in range(lwe.shape[0]):
        want, tr = oracle.bootstrap(p, lwe[b], bsk, ksk, tv, trace=True)
        assert np.array_equal(glwe[b], tr["acc_final"]), f"{name}: blind rotation, sample {b}"
        assert np.array_equal(got[b], want), f"{name}: bootstrap, sample {b}"
    # per-sample test vectors
    tvs = np.stack([np.roll(tv, b) for b in range(lwe.shape[0])])
    got2 = ctx.bootstrap(lwe, tvs)
    for b in (0, 3, 8):
        assert np.array_equal(got2[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tvs[b]))
    # single ciphertext (the reference's call shape)
    assert np.array_equal(ctx.bootstrap(lwe[4], tv), got[4])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name", [s[0] for s in SHAPES])
def test_external_product_and_cmux(oracle, contexts, name, backend):
    p, ctx, *_ = contexts(name, backend)
    rng = np.random.default_rng(3)
    batch = 5
    ggsw = rand_u32(rng, (batch, p.R, p.k + 1, p.N))
    ct0 = rand_u32(rng, (batch, p.k + 1, p.N))
    ct1 = rand_u32(rng, (batch, p.k + 1, p.N))
    ct0[:, :, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
    # extreme key words: both signed 16-bit halves at their bounds
    ggsw[0, :, :, :6] = [0x7FFF7FFF, 0x80008000, 0x7FFF8000, 0xFFFFFFFF, 0x00008000, 0x80000000]
    # one GGSW per sample
    got = ctx.external_product(ggsw, ct0)
    for b in range(batch):
        assert np.array_equal(got[b], oracle.external_product(p, ggsw[b], ct0[b])), (name, b)
    # one GGSW shared by the batch (the blind-rotation shape)
    got = ctx.external_product(ggsw[0], ct0)
    for b in range(batch):
        assert np.array_equal(got[b], oracle.external_product(p, ggsw[0], ct0[b])), (name, b)
    res, clob = ctx.cmux(ggsw, ct0, ct1)
    for b in range(batch):
        want, want_clob = oracle.cmux(p, ggsw[b], ct0[b], ct1[b])
        assert np.array_equal(res[b], want) and np.array_equal(clob[b], want_clob), (name, b)


@pytest.mark.parametrize("name", [s[0] for s in SHAPES])
def test_key_switch_and_sample_extract(oracle, contexts, name):
    p, ctx, lwe, bsk, ksk, tv = contexts(name)
    rng = np.random.default_rng(4)
    batch = 37
    big = rand_u32(rng, (batch, p.big_n + 1))
    got = ctx.key_switch(big)
    for b in (0, 1, 17, 36):
        assert np.array_equal(got[b], oracle.key_switch_lwe(big[b], p.big_n, p.n, p.ks, ksk)), (name, b)
    glwe = rand_u32(rng, (3, p.k + 1, p.N))
    for idx in (0, 1, p.N // 2, p.N - 1):
        got = ctx.sample_extract(glwe, idx)
        for b in range(3):
            assert np.array_equal(got[b], oracle.sample_extract(p, glwe[b], idx)), (name, idx, b)


def test_small_ops(oracle, contexts):
    p, ctx, *_ = contexts("cfg2_small")
    rng = np.random.default_rng(5)
    v = rand_u32(rng, 10000)
    v[:6] = [0, 1, 0xFFFFFFFF, 0x80000000, 0xF8F8F8F8, 0x00000F80]
    m = pkg()
    assert np.array_equal(ctx.decompose(v, m.DECOMPOSER_PBS), oracle.decompose(p.pbs, v))
    assert np.array_equal(ctx.decompose(v, m.DECOMPOSER_KS), oracle.decompose(p.ks, v))
    for log_to in (1, 10, 11, 12, 31):
        assert np.array_equal(ctx.switch_modulus(v, 32, log_to), oracle.switch_modulus(v, 32, log_to))
    glwe = rand_u32(rng, (4, p.k + 1, p.N))
    assert np.array_equal(ctx.decompose_glwe(glwe)[2], oracle.decompose_glwe_ciphertext(glwe[2], p.pbs))
    idx = np.array([-1, 0, p.N + 3, -(5 * p.N) - 7], dtype=np.int64)
    got = ctx.glwe_mul_monomial(glwe, idx)
    for b in range(4):
        assert np.array_equal(got[b], oracle.glwe_mul_monomial(glwe[b], int(idx[b])))
    # LweCiphertext Add / Mul<u32> (lwe.rs:9-23) and the gate input 2*ct1 + ct0 (boolean.rs:18)
    a, b = rand_u32(rng, (7, p.n + 1)), rand_u32(rng, (7, p.n + 1))
    assert np.array_equal(ctx.lwe_linear(1, a, 1, b), (a + b).astype(np.uint32))
    assert np.array_equal(ctx.lwe_linear(0xFFFFFFFD, a), (a * np.uint32(0xFFFFFFFD)).astype(np.uint32))
    assert np.array_equal(ctx.lwe_linear(1, a, 2, b), (b * np.uint32(2) + a).astype(np.uint32))
    for lut in ([0, 1, 2, 3], [3, 1, 0, 2]):
        assert np.array_equal(m.construct_test_from_lut(to_pkg_params(p), lut), oracle.construct_test_from_lut(p, lut))


def test_gates_bit_exact_and_decrypt(oracle):
    """boolean.rs:67-101: truth table through real keys, plus NAND/OR/XOR via the closure hook;
    outputs bit-exact vs the oracle AND decrypting to the right bit."""
    p = oracle.REF_TEST
    rng = oracle.Rng(2024)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        gates = {"and": (m.GATE_AND, lambda l, r: l & r), "or": (m.GATE_OR, lambda l, r: l | r),
                 "nand": (m.GATE_NAND, lambda l, r: 1 - (l & r)), "xor": (m.GATE_XOR, lambda l, r: l ^ r)}
        ct1 = np.stack([oracle.encrypt_lwe(p, lwe_sk, (i >> 1) & 1, rng) for i in range(4)])
        ct0 = np.stack([oracle.encrypt_lwe(p, lwe_sk, i & 1, rng) for i in range(4)])
        for name, (truth, f) in gates.items():
            out = ctx.gate(truth, ct0, ct1)
            for i in range(4):
                assert np.array_equal(out[i], oracle.boolean_gate(p, f, ct0[i], ct1[i], bsk, ksk)), name
                assert oracle.decrypt_lwe_message(p, lwe_sk, out[i]) == f((i >> 1) & 1, i & 1), name
        # identity-LUT PBS refreshes every message (bootstrapping.rs:194-230)
        tv = m.construct_identity_test_vector(to_pkg_params(p))
        cts = np.stack([oracle.encrypt_lwe(p, lwe_sk, msg, rng) for msg in range(4)])
        out = ctx.bootstrap(cts, tv)
        for msg in range(4):
            assert oracle.decrypt_lwe_message(p, lwe_sk, out[msg]) == msg


def test_error_behaviour(oracle):
    """The reference panics (assert!/unwrap); across the C ABI those become status codes."""
    m = pkg()
    p = oracle.REF_TEST
    with pytest.raises(m.TfheError) as e:  # levels > floor(32/log_base): endless loop in the reference
        m.Context(m.TfheParams(1, 10, 8, m.DecomposerParams(7, 5)))
    assert e.value.status == 1
    with pytest.raises(m.TfheError) as e:  # N = 256 has no kernel
        m.Context(m.TfheParams(1, 8, 8, m.DecomposerParams(4, 6)))
    assert e.value.status == 2
    with m.Context(to_pkg_params(p)) as ctx:
        lwe = np.zeros((2, p.n + 1), dtype=np.uint32)
        tv = np.zeros(p.N, dtype=np.uint32)
        with pytest.raises(m.TfheError) as e:  # no key loaded
            ctx.bootstrap(lwe, tv)
        assert e.value.status == 3
        _, bsk, ksk, _ = oracle.synthetic_inputs(p, 1, cfg_index=1)
        ctx.load_bootstrapping_key(bsk, ksk)
        tv[5] = 4  # glwe.rs:144 assert!(m < 2^log_p)
        with pytest.raises(m.TfheError) as e:
            ctx.bootstrap(lwe, tv)
        assert e.value.status == 5
        with pytest.raises(m.TfheError) as e:  # bootstrapping.rs:127 assert!(sample_index < N)
            ctx.sample_extract(np.zeros((1, p.k + 1, p.N), dtype=np.uint32), p.N)
        assert e.value.status == 5


def test_full_size_cfg2_sample_parity(oracle):
    """BASELINE cfg2 at full size (N=1024, k=1, n=630, l=3, logB=7): a batch of 64 on the GPU with
    both backends, three samples checked against the oracle end to end (each oracle PBS is ~8 G u32
    MACs), and the two backends against each other on the whole batch."""
    p = oracle.CFG2
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 64, cfg_index=2)
    m = pkg()
    outs = {}
    for backend in BACKENDS:
        if backend == "fp64-p49":
            continue  # cfg2's base 2^7 with 3 levels is outside the 49-bit field's bound (next test)
        with m.Context(to_pkg_params(p), backend=backend_id(backend)) as ctx:
            ctx.load_bootstrapping_key(bsk, ksk)
            outs[backend] = ctx.bootstrap(lwe, tv)
            # determinism: same inputs, same bits
            assert np.array_equal(outs[backend], ctx.bootstrap(lwe, tv))
    assert np.array_equal(outs["fp64"], outs["goldilocks"])
    assert np.array_equal(outs["fp64"], outs["goldilocks-split"])
    for b in (0, 31, 63):
        assert np.array_equal(outs["fp64"][b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


def test_backend_selection(oracle):
    """AUTO picks fp64 when (k+1) l N B 2^15 < 2^40.9 and Goldilocks otherwise; forcing fp64 outside
    its bound is refused (TFHE_ERR_EXACTNESS)."""
    m = pkg()
    with m.Context(to_pkg_params(oracle.CFG2)) as ctx:  # N = 1024, rounding-error bound 0.011 < 1/4
        assert ctx.backend == "fp64-fft"
    with m.Context(to_pkg_params(oracle.CFG5)) as ctx:  # N = 2048: bound 0.17; two samples per team (round 3)
        assert ctx.backend == "fp64-fft"
    with pytest.raises(m.TfheError) as e:   # N = 1024, 2 levels of 2^16: error bound 2.5 > 1/4
        m.Context(m.TfheParams(1, 10, 2, m.DecomposerParams(16, 2)), backend=m.BACKEND_FP64_FFT)
    assert e.value.status == m.TFHE_ERR_EXACTNESS
    with pytest.raises(m.TfheError) as e:   # 6 * 1024 * 2^7 * 2^31 = 2^50.6 > 2^48.25
        m.Context(to_pkg_params(oracle.CFG2), backend=m.BACKEND_FP64_P49)
    assert e.value.status == 7
    # the reference's default parameters (18 * 512 * 2^4 * 2^31 = 2^48.17) fit the 49-bit field, but AUTO takes the
    # complex transform wherever its rounding bound holds (since round 3 it is ahead with many digit rows too).  All
    # fields agree bit for bit on a batch
    p3 = oracle.CFG3
    lwe3, bsk3, ksk3, tv3 = oracle.synthetic_inputs(p3, 32, cfg_index=3)
    outs = {}
    for name, b in (("auto", m.BACKEND_AUTO), ("fp64", m.BACKEND_FP64), ("gl", m.BACKEND_GOLDILOCKS),
                    ("gls", m.BACKEND_GOLDILOCKS_SPLIT), ("p49", m.BACKEND_FP64_P49)):
        with m.Context(to_pkg_params(p3), backend=b) as ctx:
            if name == "auto":
                assert ctx.backend == "fp64-fft"
            ctx.load_bootstrapping_key(bsk3, ksk3)
            outs[name] = ctx.bootstrap(lwe3, tv3)
    for name in ("fp64", "gl", "gls", "p49"):
        assert np.array_equal(outs[name], outs["auto"]), name
    with m.Context(m.TfheParams(1, 10, 8, m.DecomposerParams(2, 10))) as ctx:   # 20 digit rows: still the complex transform
        assert ctx.backend == "fp64-fft"
    wide = m.TfheParams(1, 11, 2, m.DecomposerParams(16, 2))  # 2 * 2048 * 2^16 * 2^15 = 2^43
    with m.Context(wide) as ctx:
        assert ctx.backend == "goldilocks"
    with pytest.raises(m.TfheError) as e:
        m.Context(wide, backend=m.BACKEND_FP64)
    assert e.value.status == 7
    # one-level decomposition with a 23-bit base at N = 2048: 2 * 2048 * 2^23 * 2^32 = 2^67 is beyond
    # plain Goldilocks, so AUTO falls through to the split-key field -- and it is still bit-exact
    p = oracle.Params(1, 11, 3, oracle.Decomposer(23, 1))
    with m.Context(to_pkg_params(p)) as ctx:
        assert ctx.backend == "goldilocks-split"
        lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 3, cfg_index=23)
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tv)
    for b in range(3):
        assert np.array_equal(out[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b
    with pytest.raises(m.TfheError) as e:
        m.Context(to_pkg_params(p), backend=m.BACKEND_GOLDILOCKS)
    assert e.value.status == 7


@pytest.mark.parametrize("log_base", [24, 27, 31])
def test_one_level_decompositions_with_bases_above_two_to_the_23(oracle, log_base):
    """decomposer.rs:42-80 accepts any log_base < 32 with log_base * levels <= 32: one level of 24, 27 or 31
    bits (digits up to 2^31).  The kernels' digit chain uses a 24-bit multiply-add only below 2^23 and plain
    shifts above; these shapes land in the split-key Goldilocks field.  Rows with every limb pattern that
    matters: all ones (digit = B after the carry), the half bit alone, random words."""
    m = pkg()
    p = oracle.Params(1, 9, 3, oracle.Decomposer(log_base, 1))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 6, cfg_index=100 + log_base)
    rng = np.random.default_rng(log_base)
    ggsw = rng.integers(0, 1 << 32, size=(p.R, 2, p.N), dtype=np.uint64).astype(np.uint32)
    glwe = rng.integers(0, 1 << 32, size=(4, 2, p.N), dtype=np.uint64).astype(np.uint32)
    glwe[0, :, :] = 0xFFFFFFFF
    glwe[1, :, ::2] = np.uint32(1) << np.uint32(31)
    glwe[1, :, 1::2] = (np.uint32(1) << np.uint32(32 - log_base + log_base - 1)) - np.uint32(1)
    with m.Context(to_pkg_params(p)) as ctx:
        assert ctx.backend == "goldilocks-split"
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tv)
        got = ctx.external_product(ggsw, glwe)
        digits = ctx.decompose(glwe[:2].reshape(-1)[:4096])
    for b in range(lwe.shape[0]):
        assert np.array_equal(out[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), (log_base, b)
    for b in range(4):
        assert np.array_equal(got[b], oracle.external_product(p, ggsw, glwe[b])), (log_base, b)
    assert np.array_equal(digits, oracle.decompose(p.pbs, glwe[:2].reshape(-1)[:4096]))


def test_full_size_cfg3_nand_gate_stream(oracle):
    """BASELINE cfg3 = the reference's default parameters (lib.rs:101-123: N=512, k=2, n=722, l=6,
    logB=4) with REAL keys: a stream of NAND gates through the closure hook (test_vector.rs:5),
    bit-exact vs the oracle on sampled gates, and every gate decrypts to NAND of its inputs."""
    p = oracle.CFG3
    rng = oracle.Rng(31337)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    batch = 256
    bits = np.random.default_rng(9).integers(0, 2, size=(batch, 2))
    ct1 = np.stack([oracle.encrypt_lwe(p, lwe_sk, int(b[0]), rng) for b in bits])
    ct0 = np.stack([oracle.encrypt_lwe(p, lwe_sk, int(b[1]), rng) for b in bits])
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.gate(m.GATE_NAND, ct0, ct1)
    for i in range(batch):
        assert oracle.decrypt_lwe_message(p, lwe_sk, out[i]) == 1 - (bits[i, 0] & bits[i, 1]), i
    nand = lambda l, r: 1 - (l & r)
    for i in (0, 100, 255):
        assert np.array_equal(out[i], oracle.boolean_gate(p, nand, ct0[i], ct1[i], bsk, ksk)), i


def test_cfg3_nand_gate_stream_65536_scrambled_copies(oracle):
    """BASELINE configs[2] as stated: a stream of 65,536 homomorphic NAND gates at the reference's default parameters
    (boolean.rs:9-53, lib.rs:101-123) in ONE tfhe_gate_batch_device call with real keys.  The stream is assembled on the
    device from 256 distinct encrypted input pairs in a scrambled order (as test_cfg4_per_gpu_share_scrambled_copies
    does for cfg4): every one of the 65,536 outputs must equal the output of its source pair in the 256-gate run --
    whichever workgroup, launch of the grid or key-switch tile it lands in -- and the 256-gate run itself decrypts to
    NAND of its inputs and matches the oracle's boolean gate word for word on sampled gates."""
    import torch
    p = oracle.CFG3
    rng = oracle.Rng(4242)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    dev = torch.device("cuda", 0)
    distinct, stream = 256, 65536
    bits = np.random.default_rng(19).integers(0, 2, size=(distinct, 2))
    ct1 = np.stack([oracle.encrypt_lwe(p, lwe_sk, int(b[0]), rng) for b in bits])
    ct0 = np.stack([oracle.encrypt_lwe(p, lwe_sk, int(b[1]), rng) for b in bits])
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        base = ctx.gate(m.GATE_NAND, ct0, ct1)
        order = torch.from_numpy(np.random.default_rng(23).integers(0, distinct, size=stream)).to(dev)
        ct0_d = torch.from_numpy(ct0.view(np.int32)).to(dev).index_select(0, order).contiguous()
        ct1_d = torch.from_numpy(ct1.view(np.int32)).to(dev).index_select(0, order).contiguous()
        out_d = ctx.gate(m.GATE_NAND, ct0_d, ct1_d)
        want_d = torch.from_numpy(base.view(np.int32)).to(dev).index_select(0, order)
        assert tuple(out_d.shape) == (stream, p.n + 1)
        assert torch.equal(out_d, want_d)
        ctx.set_stream(None)
    for i in range(distinct):
        assert oracle.decrypt_lwe_message(p, lwe_sk, base[i]) == 1 - (bits[i, 0] & bits[i, 1]), i
    nand = lambda l, r: 1 - (l & r)
    for i in (0, 77, 255):
        assert np.array_equal(base[i], oracle.boolean_gate(p, nand, ct0[i], ct1[i], bsk, ksk)), i


def test_full_size_cfg1_and_cfg5_single_sample(oracle):
    """BASELINE cfg1 (N=512, k=1, n=500, l=2, logB=8) and cfg5 (N=2048, k=2, n=630, l=4, logB=8,
    log_p=4 LUT of 16 entries) at full size: one sample each against the oracle (cfg5 costs the
    oracle ~95 G u32 MACs, so one sample only) plus batch-internal consistency."""
    m = pkg()
    for name, p, lut in (("cfg1", oracle.CFG1, None),
                         ("cfg5", oracle.CFG5, np.random.default_rng(5).integers(0, 16, size=16))):
        lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 4, cfg_index=1 if name == "cfg1" else 5, lut=lut)
        lwe = lwe.copy()
        lwe[3] = lwe[0]  # identical inputs must give identical outputs wherever they sit in the batch
        with m.Context(to_pkg_params(p)) as ctx:
            ctx.load_bootstrapping_key(bsk, ksk)
            out = ctx.bootstrap(lwe, tv)
            # the full BASELINE batch (4096) built from these 4 rows in a scrambled order: every copy
            # must reproduce the 4-row run, whichever workgroup / round / key-switch split it lands in
            order = np.random.default_rng(17).integers(0, 4, size=4096)
            big = ctx.bootstrap(lwe[order], tv)
        assert np.array_equal(out[3], out[0]), name
        assert np.array_equal(out[0], oracle.bootstrap(p, lwe[0], bsk, ksk, tv)), name
        assert np.array_equal(big, out[order]), name


def test_batch_4096_properties_cfg2(oracle):
    """BASELINE cfg2 at the full batch of 4096 through the device entry points: size-independent
    properties instead of 4096 oracle runs -- (1) a batch assembled from 64 distinct ciphertexts
    repeated 64 times returns 64 identical groups that equal the 64-ciphertext run, (2) per-sample
    test vectors rotated by r equal the shared-vector result of ... a different LUT (checked on
    three samples against the oracle)."""
    import torch
    p = oracle.CFG2
    m = pkg()
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 64, cfg_index=2)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        base = ctx.bootstrap(lwe, tv)
        big = np.tile(lwe, (64, 1))
        dev = torch.device("cuda", 0)
        lwe_d = torch.from_numpy(big.view(np.int32)).to(dev)
        tv_d = torch.from_numpy(tv.view(np.int32)).to(dev)
        ctx.use_torch_stream()
        out_d = ctx.bootstrap(lwe_d, tv_d)
        torch.cuda.synchronize()
        out = out_d.cpu().numpy().view(np.uint32)
        ctx.set_stream(None)
    assert out.shape == (4096, p.n + 1)
    assert np.array_equal(out.reshape(64, 64, -1), np.broadcast_to(base, (64, 64, p.n + 1)))
    for b in (5, 40):
        assert np.array_equal(base[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


@pytest.mark.parametrize("batch,per_sample", [(5000, False), (5000, True), (20000, False), (17000, True)],
                         ids=["stride-shared-ggsw", "stride-ggsw-per-sample", "queue-shared-ggsw", "queue-ggsw-per-sample"])
def test_external_product_batches_beyond_the_resident_teams(oracle, batch, per_sample):
    """The standalone external-product kernel is a persistent grid: with more samples than resident teams
    (1,024 at this shape) every team walks several samples -- by stride up to 15 per team, through a
    device work queue from 16.  Batches that are not multiples of the grid, through the device entry point,
    against the same products launched in chunks of 256 (one sample per team) and against the oracle on rows
    of the first, a middle and the last pass; the queue's counters must be back at zero after every launch
    (three launches in a row, then three replays of a captured graph)."""
    import torch
    p = oracle.Params(1, 10, 4, oracle.Decomposer(7, 3))
    m = pkg()
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(77)
    rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
    with m.Context(to_pkg_params(p)) as ctx:
        ggsw = rw(batch if per_sample else 1, p.R, p.k + 1, p.N)
        prep = ctx.prepare_ggsw_device(ggsw)
        glwe = rw(batch, p.k + 1, p.N)
        full = ctx.external_product_prepared(prep, glwe)
        parts = torch.cat([ctx.external_product_prepared(prep[i:i + 256] if per_sample else prep, glwe[i:i + 256].contiguous())
                           for i in range(0, batch, 256)])
        assert torch.equal(full, parts)
        for _ in range(2):
            assert torch.equal(ctx.external_product_prepared(prep, glwe), parts)
        side = torch.cuda.Stream()
        out = torch.empty_like(glwe)
        with torch.cuda.stream(side):
            ctx.external_product_prepared(prep, glwe, out=out)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                ctx.external_product_prepared(prep, glwe, out=out)
            for _ in range(3):
                out.zero_()
                graph.replay()
                side.synchronize()
                assert torch.equal(out, parts)
        gg, gl, fu = (t.cpu().numpy().view(np.uint32) for t in (ggsw, glwe, full))
        ctx.set_stream(None)
    for b in (0, 1023, 1024, batch // 2, batch - 2, batch - 1):
        assert np.array_equal(fu[b], oracle.external_product(p, gg[b if per_sample else 0], gl[b])), b


def test_cfg4_per_gpu_share_scrambled_copies(oracle):
    """BASELINE configs[3] shards 2^20 bootstraps over 8 GPUs: ONE GPU's share is a batch of 2^17 at
    cfg2 parameters.  The batch is assembled on the device from 256 distinct ciphertexts in a
    scrambled order; every one of the 131,072 outputs must equal the output of its source row in the
    256-row run (whose rows 0, 100, 255 are checked against the oracle), whichever workgroup, wave of
    the grid or key-switch tile it lands in."""
    import torch
    p = oracle.CFG4
    m = pkg()
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 256, cfg_index=4)
    dev = torch.device("cuda", 0)
    batch = 1 << 17
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        base = ctx.bootstrap(lwe, tv)
        order = torch.from_numpy(np.random.default_rng(17).integers(0, 256, size=batch)).to(dev)
        lwe_d = torch.from_numpy(lwe.view(np.int32)).to(dev).index_select(0, order).contiguous()
        tv_d = torch.from_numpy(tv.view(np.int32)).to(dev)
        out_d = ctx.bootstrap(lwe_d, tv_d)
        want_d = torch.from_numpy(base.view(np.int32)).to(dev).index_select(0, order)
        assert tuple(out_d.shape) == (batch, p.n + 1)
        assert torch.equal(out_d, want_d)
        ctx.set_stream(None)
    for b in (0, 100, 255):
        assert np.array_equal(base[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


@pytest.mark.parametrize("n,batch,streams", [(6, 512, 1), (16, 2304, 2)])
def test_torch_entry_points_follow_the_current_stream(oracle, n, batch, streams):
    """The torch-tensor entry points are stream-ordered like torch ops: inputs produced on a side
    stream right before the call, output consumed on it right after, no host synchronisation and no
    use_torch_stream() by the caller; then the same on the default stream.  The second case is a batch
    larger than the chip over a key of two slices: its blind rotation forks onto the context's second
    stream and joins again (kernels.hip::blind_rotate_plan) -- the producer must be waited for by both
    streams and the consumer must wait for both."""
    import torch
    p = oracle.Params(1, 10, n, oracle.Decomposer(7, 3))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, batch, cfg_index=12)
    m = pkg()
    dev = torch.device("cuda", 0)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        plan = ctx.blind_rotate_plan(batch)
        assert plan["streams"] == streams and (plan["segments"] > 1) == (streams == 2), plan
        want = ctx.bootstrap(lwe, tv)
        for b in (0, batch // 2 + 1, batch - 1):  # both halves of the batch against the oracle
            assert np.array_equal(want[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b
        src = torch.from_numpy(lwe.view(np.int32)).to(dev)
        tv_d = torch.from_numpy(tv.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        for stream in (torch.cuda.Stream(), torch.cuda.current_stream(), torch.cuda.Stream()):
            with torch.cuda.stream(stream):
                # a long producer chain on this stream: the input exists only once it has run
                x = src.clone()
                for _ in range(200):
                    x = x + 1
                x = x - 200
                y = ctx.bootstrap(x, tv_d)
                z = y.clone() ^ 0          # consumer on the same stream
            stream.synchronize()
            assert np.array_equal(z.cpu().numpy().view(np.uint32), want)
        ctx.set_stream(None)


@pytest.mark.parametrize("n,batch", [(6, 16), (16, 2304)])
def test_device_entry_points_are_graph_capturable(oracle, n, batch):
    """The `_device` entry points promise no allocation and no synchronisation once the workspace is
    reserved: capture a whole bootstrap (blind rotation + key switch) into a HIP graph on a torch
    stream, replay it on new inputs, and compare with the eager result and the oracle.  Second case: a
    batch whose eager plan uses the context's second stream -- inside a capture the launches stay on the
    captured stream (segments, one stream), same bits."""
    import torch
    p = oracle.Params(1, 10, n, oracle.Decomposer(7, 3))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, batch, cfg_index=11)
    m = pkg()
    dev = torch.device("cuda", 0)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.reserve(batch)
        lwe_d = torch.from_numpy(lwe.view(np.int32).copy()).to(dev)
        tv_d = torch.from_numpy(tv.view(np.int32).copy()).to(dev)
        out_d = torch.empty_like(lwe_d)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ctx.use_torch_stream()
            ctx.bootstrap(lwe_d, tv_d, out=out_d)  # warm-up outside the capture (one-time kernel attributes)
            side.synchronize()
            eager = out_d.cpu().numpy().view(np.uint32).copy()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                ctx.bootstrap(lwe_d, tv_d, out=out_d)
            # replay on different inputs written into the captured buffers
            lwe2 = np.roll(lwe, 3, axis=0)
            lwe_d.copy_(torch.from_numpy(lwe2.view(np.int32).copy()))
            graph.replay()
            side.synchronize()
            replayed = out_d.cpu().numpy().view(np.uint32).copy()
        ctx.set_stream(None)
    assert np.array_equal(replayed, np.roll(eager, 3, axis=0))
    assert np.array_equal(eager[2], oracle.bootstrap(p, lwe[2], bsk, ksk, tv))


@pytest.mark.parametrize("name", ["cfg2_small", "ref_test", "cfg5_small"])
def test_batch_edge_cases(oracle, contexts, name):
    """batch of 1, odd batches, and the empty batch (refused with a status, never a crash).  ref_test and cfg5_small
    are shapes whose teams rotate TWO samples at once (N = 512 with k = 2, N = 2048): a batch of 1 and every odd batch
    leave the last team one sample short -- it redoes its last sample in the free slot and must write it once; per-sample
    test vectors and the blind-rotation output ([batch][k+1][N], written per sample) go through the same slots."""
    p, ctx, lwe, bsk, ksk, tv = contexts(name, "auto")
    m = pkg()
    full = ctx.bootstrap(lwe, tv)
    acc = ctx.blind_rotate(lwe, tv)
    tvs = np.stack([np.roll(tv, 3 * b) for b in range(lwe.shape[0])])
    per = ctx.bootstrap(lwe, tvs)
    for b in (0, lwe.shape[0] - 1):
        assert np.array_equal(per[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tvs[b])), b
    for size in (1, 2, 3, 7):
        assert np.array_equal(ctx.bootstrap(lwe[:size], tv), full[:size])
        assert np.array_equal(ctx.blind_rotate(lwe[:size], tv), acc[:size])
        assert np.array_equal(ctx.bootstrap(lwe[:size], tvs[:size]), per[:size])
    with pytest.raises(m.TfheError) as e:
        ctx.bootstrap(np.zeros((0, p.n + 1), dtype=np.uint32), tv)
    assert e.value.status == 5


@pytest.mark.parametrize("chunk,segments,streams", [("3", None, None), ("2", None, None), ("5", None, None), ("3", "2", None),
                                                    ("8", "3", "2"), ("5", "4", "2"), ("7", "2", "2")])
def test_short_launches_with_two_samples_per_team(chunk, segments, streams):
    """kernels.hip::launch_blind_rotate cuts a batch into groups of `chunk` samples, the n CMUX iterations into `segments`
    launches whose accumulators wait in global memory in between, and lets the two halves of a group alternate on two
    streams (blind_rotate_plan); TFHE_BR_CHUNK, TFHE_BR_SEGMENTS and TFHE_BR_STREAMS override the plan.  With groups of 3, 2,
    5, 7 and 8 samples a batch of 8 exercises the offsets of inputs, per-sample test vectors, parked accumulators and both
    outputs, and -- at the shapes whose teams rotate two samples -- a last team that is one sample short in every odd launch
    (with two streams: halves of 4 + 4, 3 + 2 and 4 + 3 samples; at two samples per team the first half is rounded up to
    whole teams); with 2, 3 and 4 segments (n = 4, 2, 3: more segments than iterations are clamped) every launch but the
    first resumes.  Three shapes against the oracle, in a child process (the variables are read once)."""
    import subprocess
    import sys
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, TFHE_BR_CHUNK=chunk)
    for name, value in (("TFHE_BR_SEGMENTS", segments), ("TFHE_BR_STREAMS", streams)):
        env.pop(name, None)
        if value:
            env[name] = value
    run = subprocess.run([sys.executable, os.path.join(here, "br_chunk_probe.py")], env=env, capture_output=True, text=True,
                         timeout=600)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-2000:]
    assert "short launches parity True" in run.stdout


def test_blind_rotate_plan_matches_its_description(oracle):
    """tfhe_debug_blind_rotate_plan: a batch the chip rotates at once goes out as one launch; a larger one at N <= 1024 as
    key slices of 2 MiB on two streams (kernels.hip::blind_rotate_plan).  The full-size parity tests run under exactly
    these plans; this pins the numbers the bench line and DESIGN.md quote."""
    m = pkg()
    p = oracle.Params(1, 10, 630, oracle.Decomposer(7, 3), oracle.Decomposer(2, 8), log_p=2)  # cfg2
    with m.Context(to_pkg_params(p)) as ctx:
        small = ctx.blind_rotate_plan(8)
        assert small["segments"] == 1 and small["streams"] == 1 and small["launches"] == 1
        resident = small["resident_samples"]
        assert resident >= 256  # at least one team per CU
        at = ctx.blind_rotate_plan(resident)
        assert at["segments"] == 1 and at["streams"] == 1
        big = ctx.blind_rotate_plan(4096)
        assert 4096 > resident and big["streams"] == 2
        key_bytes = 630 * 2 * 3 * 2 * 2 * 1024 * 8  # n (k+1) l (k+1) parts N 8
        per = -(-630 // -(-key_bytes // (2 << 20)))  # iterations per launch: slices of at most 2 MiB -> 60 -> 11 iterations
        assert per == 11 and big["segments"] == -(-630 // per) == 58
        assert big["groups"] == 1 and big["launches"] == 116
        assert ctx.blind_rotate_plan(1 << 20)["groups"] == 8  # groups of 131,072


def test_encrypted_adder_gate_graph(oracle):
    """a graph of AND/OR/XOR gates evaluated with all ciphertexts resident on the device
    (tfhe_gate_batch_device, gates of one level batched): 32 independent 4-bit ripple-carry adders
    under real keys with the reference's cfg(test)-sized LWE key.  Every decrypted sum must equal
    a + b, and sampled wires must be bit-exact with the oracle gate by gate."""
    import importlib
    import torch
    p = oracle.Params(2, 9, 16, oracle.Decomposer(4, 6))  # reference defaults with a short LWE key
    rng = oracle.Rng(777)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    gates = importlib.import_module("tfhe_research_amd.gates")
    circuit, out_wires = gates.ripple_carry_adder(4)
    inst = 32
    nprng = np.random.default_rng(3)
    a, b = nprng.integers(0, 16, size=inst), nprng.integers(0, 16, size=inst)
    bits = np.array([[(a[i] >> j) & 1 for j in range(4)] + [(b[i] >> j) & 1 for j in range(4)] for i in range(inst)])
    cts = np.stack([np.stack([oracle.encrypt_lwe(p, lwe_sk, int(bit), rng) for bit in row]) for row in bits])
    dev = torch.device("cuda", 0)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.use_torch_stream()
        wires = gates.evaluate(ctx, circuit, torch.from_numpy(cts.view(np.int32)).to(dev))
        torch.cuda.synchronize()
        wires = wires.cpu().numpy().view(np.uint32)
        ctx.set_stream(None)
    for i in range(inst):
        clear = circuit.evaluate_clear(bits[i].tolist())
        got = [oracle.decrypt_lwe_message(p, lwe_sk, wires[i, w]) for w in range(circuit.n_wires)]
        assert got == clear, i
        total = sum(got[w] << j for j, w in enumerate(out_wires))
        assert total == a[i] + b[i]
    # gate-by-gate bit-exactness on a few wires of instance 0 (inputs of each gate taken from the GPU run)
    fns = {"and": lambda l, r: l & r, "or": lambda l, r: l | r, "xor": lambda l, r: l ^ r}
    for g in (0, 5, len(circuit.gates) - 1):
        kind, lhs, rhs = circuit.gates[g]
        want = oracle.boolean_gate(p, fns[kind], wires[0, rhs], wires[0, lhs], bsk, ksk)
        assert np.array_equal(wires[0, circuit.n_inputs + g], want), g


@pytest.mark.parametrize("cfg,samples", [("cfg2", 384), ("cfg3", 128), ("cfg1", 256), ("cfg5", 16)])
def test_many_full_size_samples_against_the_oracle_on_all_host_cores(oracle, cfg, samples):
    """Every BASELINE configuration with a full 4096 batch on the GPU and `samples` of its rows --
    spread over the batch, edge rows included -- recomputed by the oracle on the host cores in
    parallel (ctypes releases the GIL; 0.4 s (cfg1) to 30 s (cfg5) per bootstrap per core).  Every one
    of them must match bit for bit."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    p = oracle.CONFIGS[cfg]
    lut = np.random.default_rng(5).integers(0, 1 << p.log_p, size=1 << p.log_p) if cfg == "cfg5" else None
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 4096, cfg_index=int(cfg[3:]), lut=lut)
    lwe = lwe.copy()
    lwe[0, 0] = 0                 # a~ = 0
    lwe[1, p.n] = 0xFFFFFFFF      # b~ rounds to 2N and wraps
    lwe[4095, :] = 0x80000000
    with pkg().Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tv)
    rows = sorted(set([0, 1, 4095] + list(np.random.default_rng(7).choice(4096, size=samples - 3, replace=False))))
    workers = max(1, min(16, os.cpu_count() or 1))
    with ThreadPoolExecutor(workers) as pool:
        want = list(pool.map(lambda b: oracle.bootstrap(p, lwe[b], bsk, ksk, tv), rows))
    bad = [int(b) for b, w in zip(rows, want) if not np.array_equal(out[b], w)]
    assert not bad, f"{cfg}: rows {bad} differ from the oracle"


@pytest.mark.parametrize("cfg,fast,count,slow_count", [("cfg3", "fp64-fft", 16384, 2048), ("cfg2", "fp64-fft", 8192, 2048),
                                                       ("cfg5", "fp64-fft", 1024, 256), ("cfg1", "fp64-fft", 8192, 2048)])
def test_fields_agree_on_large_batches(oracle, cfg, fast, count, slow_count):
    """Independent arithmetic, same bits: `count` random full-size bootstraps in the field AUTO
    picks against the 42-bit fp64 field where AUTO picks another one, and the first `slow_count` of them against
    Goldilocks and Goldilocks-split -- every output word equal."""
    import torch
    p = oracle.CONFIGS[cfg]
    m = pkg()
    lut = np.random.default_rng(5).integers(0, 1 << p.log_p, size=1 << p.log_p)
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=int(cfg[3:]), lut=lut)
    gen = torch.Generator(device="cuda:0").manual_seed(99)
    lwe = torch.randint(-(1 << 31), (1 << 31) - 1, (count, p.n + 1), dtype=torch.int32, device="cuda:0", generator=gen)
    tv_d = torch.from_numpy(tv.view(np.int32)).to("cuda:0")
    runs = [("auto", m.BACKEND_AUTO, count), ("gl", m.BACKEND_GOLDILOCKS, slow_count), ("gls", m.BACKEND_GOLDILOCKS_SPLIT, slow_count)]
    if fast in ("fp64-p49", "fp64-fft"):
        runs.append(("p42", m.BACKEND_FP64, count))
    outs = {}
    for name, b, c in runs:
        with m.Context(to_pkg_params(p), backend=b) as ctx:
            if name == "auto":
                assert ctx.backend == fast
            ctx.load_bootstrapping_key(bsk, ksk)
            ctx.use_torch_stream()
            outs[name] = ctx.bootstrap(lwe[:c].contiguous(), tv_d)
            torch.cuda.synchronize()
            ctx.set_stream(None)
    for name, _, c in runs[1:]:
        assert torch.equal(outs["auto"][:c], outs[name]), name


def test_backends_agree_at_full_size_cfg2_with_a_key_dependent_rotation(oracle):
    """BASELINE cfg2 (N=1024, k=1, n=630, l=3, log2 B=7) with the ALIGNED decomposer: in the literal mode a trivial
    accumulator has no bit below 2^29 while bits 28..31 are never decomposed, so every digit is zero and the blind
    rotation ignores the key; aligned, all 630 CMUXes do real work.  4,096 random ciphertexts and a random
    key through the complex-FFT backend and the 42-bit prime field, the first 512 also through both Goldilocks fields:
    every output word equal, and 96 rows spread over the batch against the oracle in the same mode (host cores in
    parallel) -- the full-size key-dependent cfg2 check against the CPU restatement itself, not only between backends."""
    import torch
    p = oracle.CFG2
    m = pkg()
    lut = np.random.default_rng(7).integers(0, 1 << p.log_p, size=1 << p.log_p)
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=2, lut=lut)
    gen = torch.Generator(device="cuda:0").manual_seed(2024)
    count, slow = 4096, 512
    lwe = torch.randint(-(1 << 31), (1 << 31) - 1, (count, p.n + 1), dtype=torch.int32, device="cuda:0", generator=gen)
    tv_d = torch.from_numpy(tv.view(np.int32)).to("cuda:0")
    outs = {}
    for name, b, c in (("fft", m.BACKEND_FP64_FFT, count), ("p42", m.BACKEND_FP64, count), ("gl", m.BACKEND_GOLDILOCKS, slow),
                       ("gls", m.BACKEND_GOLDILOCKS_SPLIT, slow)):
        with m.Context(to_pkg_params(p), backend=b) as ctx:
            ctx.set_decomposer_alignment(True)
            ctx.load_bootstrapping_key(bsk, ksk)
            ctx.use_torch_stream()
            outs[name] = ctx.bootstrap(lwe[:c].contiguous(), tv_d)
            torch.cuda.synchronize()
            ctx.set_stream(None)
    assert torch.unique(outs["fft"]).numel() > 100000  # the rotation did depend on the data
    assert torch.equal(outs["fft"], outs["p42"])
    assert torch.equal(outs["fft"][:slow], outs["gl"]) and torch.equal(outs["fft"][:slow], outs["gls"])
    import os
    from concurrent.futures import ThreadPoolExecutor
    host = outs["fft"].cpu().numpy().view(np.uint32)
    rows = lwe.cpu().numpy().view(np.uint32)
    picks = sorted(set([0, 1, 2, count - 1] + list(np.random.default_rng(11).choice(count, size=92, replace=False))))
    with oracle.decomposer_aligned(True):   # a process-wide switch the worker threads only read
        oracle.set_poly_mul_mode(1)
        with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as pool:
            want = list(pool.map(lambda b: oracle.bootstrap(p, rows[b], bsk, ksk, tv), picks))
    bad = [int(b) for b, w in zip(picks, want) if not np.array_equal(host[b], w)]
    assert len(picks) >= 64 and not bad, f"rows {bad} differ from the aligned-mode oracle"


def test_timing_ring_reads_back_every_step_of_a_loop(oracle):
    """tfhe_kernel_ms_ago: K bootstraps enqueued back to back with timing on, their kernel durations read afterwards
    (no host synchronisation inside the loop); asking further back than what was timed is refused."""
    import torch
    m = pkg()
    p = oracle.CFG1
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=1)
    lwe = torch.randint(-(1 << 31), (1 << 31) - 1, (64, p.n + 1), dtype=torch.int32, device="cuda:0")
    tv_d = torch.from_numpy(tv.view(np.int32)).to("cuda:0")
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.use_torch_stream()
        ctx.set_timing(True)
        out = torch.empty_like(lwe)
        for _ in range(5):
            ctx.bootstrap(lwe, tv_d, out=out)
        times = [ctx.kernel_ms_ago(i) for i in range(5)]
        assert all(br > 0 and ks > 0 for br, ks in times)
        assert times[0] == ctx.last_kernel_ms()
        with pytest.raises(m.TfheError) as e:
            ctx.kernel_ms_ago(5)
        assert e.value.status == m.TFHE_ERR_INVALID_ARGUMENT
        torch.cuda.synchronize()
        ctx.set_stream(None)

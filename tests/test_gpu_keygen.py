"""GPU parity tests of the encryption side: keygen / encrypt / decrypt through the
C ABI against the oracle's `_from_samples` functions on the same pre-drawn randomness (bit-exact),
and end-to-end runs at full scale (the reference's default parameters and BASELINE cfg2) whose keys
never exist on the host side of the ABI except as the random samples."""
import numpy as np
import pytest

from gpu_common import pkg, rand_u32, to_pkg_params

pytestmark = pytest.mark.gpu

SHAPES = [
    # name, k, logN, n, pbs(logB, l), ks(logB, l), log_p
    ("ref_test", 2, 9, 4, (4, 6), (4, 5), 2),
    ("cfg2_small", 1, 10, 5, (7, 3), (4, 5), 2),
    ("cfg5_small", 2, 11, 2, (8, 4), (4, 5), 4),   # two waves per polynomial
    ("k1_n2048", 1, 11, 3, (16, 2), (2, 9), 2),
]
BACKENDS = ["auto", "goldilocks", "goldilocks-split", "fp64", "fp64-fft"]


def backend_id(name):
    m = pkg()
    return {"auto": m.BACKEND_AUTO, "goldilocks": m.BACKEND_GOLDILOCKS, "fp64": m.BACKEND_FP64,
            "goldilocks-split": m.BACKEND_GOLDILOCKS_SPLIT, "fp64-fft": m.BACKEND_FP64_FFT}[name]


def noise_u32(rng, std_dev, shape):
    """two-sided rounded Gaussian on the torus (what sample_gaussian_array is meant to draw)"""
    return (np.rint(rng.normal(0.0, std_dev * 2.0 ** 32, size=shape)).astype(np.int64) & 0xFFFFFFFF).astype(np.uint32)


def glwe_samples(rng, p, lead):
    s = rand_u32(rng, tuple(lead) + (p.k + 1, p.N))
    s[..., p.k, :] = noise_u32(rng, p.glwe_std_dev, tuple(lead) + (p.N,))
    return s


def lwe_samples(rng, std_dev, rows, dim):
    s = rand_u32(rng, (rows, dim + 1))
    s[:, dim] = noise_u32(rng, std_dev, rows)
    return s


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
def test_encryption_entry_points_match_oracle(oracle, shape, backend):
    _, k, logn, n, pbs, ks, log_p = shape
    p = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), oracle.Decomposer(*ks), log_p=log_p)
    if backend == "fp64" and not (np.log2(p.R) + logn + pbs[0] + 15 < 40.9 and pbs[0] <= 9):
        pytest.skip("outside the fp64 backend's exactness bound")  # "auto" is the 49-bit field at ref_test
    if backend == "fp64-fft" and pbs[0] > 13:
        pytest.skip("outside the fp64-fft backend's rounding-error bound (tfhe_context_create refuses it)")
    rng = np.random.default_rng(1000 * logn + 10 * k + n)
    glwe_sk = rng.integers(0, 2, size=(k, p.N)).astype(np.uint32)
    lwe_sk = rng.integers(0, 2, size=n).astype(np.uint32)
    with pkg().Context(to_pkg_params(p), backend=backend_id(backend)) as ctx:
        # encrypt_glwe_zero / decrypt_glwe_ciphertext (5 rows: ragged against 4 rows per workgroup)
        s = glwe_samples(rng, p, (5,))
        s[1, :k] = 0xFFFFFFFF
        ct = ctx.glwe_encrypt_zero(glwe_sk, s)
        assert np.array_equal(ct, oracle.encrypt_glwe_zero_from_samples(p, glwe_sk, s))
        assert np.array_equal(ctx.glwe_decrypt(glwe_sk, ct), s[:, k])
        assert np.array_equal(ctx.glwe_decrypt(glwe_sk, ct)[2], oracle.decrypt_glwe_raw(p, glwe_sk, ct[2]))
        # encrypt_ggsw_plaintext, messages 0 / 1 / a non-bit value
        msgs = np.array([0, 1, 3], dtype=np.uint32)
        gs = glwe_samples(rng, p, (3, p.R))
        ggsw = ctx.ggsw_encrypt(glwe_sk, msgs, gs)
        assert np.array_equal(ggsw, oracle.encrypt_ggsw_from_samples(p, glwe_sk, msgs, gs))
        # LWE encrypt / decrypt at the small and the post-PBS dimension
        for sk in (lwe_sk, glwe_sk.reshape(-1)):
            ls = lwe_samples(rng, p.lwe_std_dev, 7, sk.size)
            pts = rand_u32(rng, 7)
            enc = ctx.lwe_encrypt(sk, ls, pts)
            assert np.array_equal(enc, oracle.encrypt_lwe_from_samples(sk, ls, pts))
            assert np.array_equal(ctx.lwe_encrypt(sk, ls), oracle.encrypt_lwe_from_samples(sk, ls))
            dec = ctx.lwe_decrypt(sk, enc)
            assert np.array_equal(dec, (pts + ls[:, sk.size]).astype(np.uint32))
            assert int(dec[3]) == oracle.decrypt_lwe_raw(sk, enc[3])
        # generate_ksk and the whole bootstrapping_key_gen
        ks_s = lwe_samples(rng, p.lwe_std_dev, p.big_n * p.ks.levels, n)
        assert np.array_equal(ctx.generate_ksk(glwe_sk, lwe_sk, ks_s),
                              oracle.generate_ksk_from_samples(glwe_sk, lwe_sk, p.ks, ks_s))
        bs = glwe_samples(rng, p, (n, p.R))
        bsk, ksk = ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, bs, ks_s, load=True)
        ebsk, eksk = oracle.bootstrapping_key_gen_from_samples(p, lwe_sk, glwe_sk, bs, ks_s)
        assert np.array_equal(bsk, ebsk) and np.array_equal(ksk, eksk)
        # the key the call installed is the key it returned: bootstrap agrees with the oracle
        lwe = rand_u32(rng, (3, n + 1))
        tv = rng.integers(0, 1 << log_p, size=p.N).astype(np.uint32)
        out = ctx.bootstrap(lwe, tv)
        for i in range(3):
            assert np.array_equal(out[i], oracle.bootstrap(p, lwe[i], ebsk, eksk, tv))


def test_encryption_refuses_non_binary_keys(oracle):
    p = oracle.REF_TEST
    m = pkg()
    rng = np.random.default_rng(3)
    with m.Context(to_pkg_params(p)) as ctx:
        sk = np.zeros((p.k, p.N), dtype=np.uint32)
        sk[1, 7] = 2
        with pytest.raises(m.TfheError) as e:
            ctx.glwe_encrypt_zero(sk, glwe_samples(rng, p, (1,)))
        assert e.value.status == m.TFHE_ERR_INVALID_ARGUMENT and "binary" in str(e.value)
        with pytest.raises(m.TfheError):
            ctx.lwe_encrypt(np.array([0, 1, 5, 0], dtype=np.uint32), lwe_samples(rng, 1e-5, 2, 4))


@pytest.mark.parametrize("cfg,valid_crypto", [("cfg3", True), ("cfg2", False)])
def test_full_scale_keygen_encrypt_bootstrap_decrypt(oracle, cfg, valid_crypto):
    """The reference's bootstrapping_works flow (bootstrapping.rs:194-230) at full scale with
    keygen, encryption, PBS and decryption all through the ABI's device entry points: the keys never
    exist on the host side except as the random samples.  A slice of the generated key, the
    ciphertexts and one bootstrap are checked bit for bit against the oracle.  cfg3 (the reference's
    default parameters) must also decrypt every message; cfg2's base 2^7 does not divide 2^32, so
    the reference's literal decomposer drops the top 4 bits there and only bit parity
    is meaningful."""
    import torch
    p = oracle.CONFIGS[cfg]
    m = pkg()
    rng = np.random.default_rng(20261003)
    lwe_sk = rng.integers(0, 2, size=p.n).astype(np.uint32)
    glwe_sk = rng.integers(0, 2, size=(p.k, p.N)).astype(np.uint32)
    bs = glwe_samples(rng, p, (p.n, p.R))
    ks_s = lwe_samples(rng, p.lwe_std_dev, p.big_n * p.ks.levels, p.n)
    batch = 256
    msgs = rng.integers(0, 1 << p.log_p, size=batch).astype(np.uint32)
    pts = (msgs << (32 - p.log_p - p.padding_bits)).astype(np.uint32)
    ls = lwe_samples(rng, p.lwe_std_dev, batch, p.n)
    dev = torch.device("cuda:0")
    as_dev = lambda a: torch.from_numpy(a.view(np.int32)).to(dev)  # noqa: E731
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.use_torch_stream()
        d_bsk, d_ksk = ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, as_dev(bs), as_dev(ks_s), load=True)
        d_ct = ctx.lwe_encrypt(lwe_sk, as_dev(ls), as_dev(pts))
        tv = as_dev(m.construct_identity_test_vector(to_pkg_params(p)))
        d_out = ctx.bootstrap(d_ct, tv)
        d_dec = ctx.lwe_decrypt(lwe_sk, d_out)
        torch.cuda.synchronize()
        bsk = d_bsk.cpu().numpy().view(np.uint32)
        ksk = d_ksk.cpu().numpy().view(np.uint32)
        dec = d_dec.cpu().numpy().view(np.uint32)
        out = d_out.cpu().numpy().view(np.uint32)
        ct = d_ct.cpu().numpy().view(np.uint32)
    # key parity on a slice (first, a middle and the last GGSW) and the whole KSK
    for i in (0, 317, p.n - 1):
        assert np.array_equal(bsk[i], oracle.encrypt_ggsw_from_samples(p, glwe_sk, lwe_sk[i:i + 1], bs[i:i + 1])[0])
    assert np.array_equal(ksk, oracle.generate_ksk_from_samples(glwe_sk, lwe_sk, p.ks, ks_s))
    assert np.array_equal(ct, oracle.encrypt_lwe_from_samples(lwe_sk, ls, pts))
    # one bootstrap under the generated key, bit for bit, and the decryption of all of them
    tv_host = m.construct_identity_test_vector(to_pkg_params(p))
    assert np.array_equal(out[5], oracle.bootstrap(p, ct[5], bsk, ksk, tv_host))
    for i in (0, batch - 1):
        assert int(dec[i]) == oracle.decrypt_lwe_raw(lwe_sk, out[i])
    if valid_crypto:
        shift = 32 - p.log_p - p.padding_bits
        decoded = ((dec.astype(np.uint64) + (1 << (shift - 1))) >> shift) & ((1 << p.log_p) - 1)
        assert np.array_equal(decoded.astype(np.uint32), msgs)


def test_convenience_keygen_encrypt_gate_decrypt(oracle):
    """Context.generate_keys / encrypt_bits / decrypt_bits: the whole client + server flow of the
    reference's boolean_gates_work (boolean.rs:67-101) at its default parameters through the package
    alone -- no oracle involved in producing keys or ciphertexts, only in the final spot check."""
    m = pkg()
    p = oracle.REF_DEFAULT
    rng = np.random.default_rng(4)
    with m.Context(to_pkg_params(p)) as ctx:
        lwe_sk, glwe_sk, bsk, ksk = ctx.generate_keys(rng)
        a = rng.integers(0, 2, size=512).astype(np.uint32)
        b = rng.integers(0, 2, size=512).astype(np.uint32)
        ca, cb = ctx.encrypt_bits(lwe_sk, a, rng), ctx.encrypt_bits(lwe_sk, b, rng)
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, ca), a)
        nand = ctx.gate(m.GATE_NAND, cb, ca)       # truth[(lhs << 1) | rhs], lhs = ct1
        xor = ctx.gate(m.GATE_XOR, cb, ca)
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, nand), 1 - (a & b))
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, xor), a ^ b)
        assert np.array_equal(ctx.decrypt_bits(lwe_sk, ctx.lwe_not(nand)), a & b)
    assert np.array_equal(nand[7], oracle.boolean_gate(p, lambda l, r: 1 - (l & r), cb[7], ca[7], bsk, ksk))
    with pytest.raises(m.TfheError):
        with m.Context(to_pkg_params(oracle.REF_TEST)) as ctx:
            ctx.encrypt_bits(np.zeros(4, dtype=np.uint32), [5])

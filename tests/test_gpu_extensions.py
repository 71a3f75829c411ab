"""GPU parity tests of the two extensions beyond the reference: the aligned
decomposer (tfhe_context_set_decomposer_alignment) and the KS-then-PBS order
(tfhe_context_set_bootstrap_order), both against the oracle in the same mode, bit-exact, plus
decryption under real keys -- including BASELINE cfg2 at full scale, which only decrypts in aligned
mode."""
import importlib

import numpy as np
import pytest

from gpu_common import pkg, rand_u32, to_pkg_params
from test_gpu_keygen import glwe_samples, lwe_samples

pytestmark = pytest.mark.gpu

BACKENDS = {"fp64": 2, "goldilocks": 1, "goldilocks-split": 3, "fp64-fft": 5}


@pytest.mark.parametrize("backend", list(BACKENDS))
@pytest.mark.parametrize("k,logn,n,pbs,ks", [(1, 10, 5, (7, 3), (7, 3)), (2, 9, 4, (5, 5), (3, 9)), (1, 11, 2, (4, 6), (4, 5))])
def test_aligned_decomposer_matches_oracle(oracle, k, logn, n, pbs, ks, backend):
    p = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), oracle.Decomposer(*ks))
    m = pkg()
    rng = np.random.default_rng(logn * 10 + k)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 4, cfg_index=40 + logn)
    glwe = rand_u32(rng, (3, k + 1, p.N))
    glwe[0, :, :4] = [0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 0xF8F8F8F8]
    vals = rand_u32(rng, 300)
    # (with the aligned decomposer the blind rotation at cfg2's shape really depends on the key: in the literal mode
    # the top four bits of a word are never decomposed, a trivial accumulator has nothing below them, every digit is
    # zero and the CMUXes leave it alone -- the external-product and keygen tests carry the arithmetic there)
    with m.Context(to_pkg_params(p), backend=BACKENDS[backend]) as ctx:
        ctx.set_decomposer_alignment(True)
        ctx.load_bootstrapping_key(bsk, ksk)
        out = ctx.bootstrap(lwe, tv)
        ep = ctx.external_product(bsk[0], glwe)
        dg = ctx.decompose_glwe(glwe)
        dp = ctx.decompose(vals, m.DECOMPOSER_PBS)
        dk = ctx.decompose(vals, m.DECOMPOSER_KS)
        big = rand_u32(rng, (3, p.big_n + 1))
        ksw = ctx.key_switch(big)
        # keygen in aligned mode: gadget factors 2^{32 - log_base*(level+1)}
        glwe_sk = rng.integers(0, 2, size=(k, p.N)).astype(np.uint32)
        lwe_sk = rng.integers(0, 2, size=n).astype(np.uint32)
        bs, kss = glwe_samples(rng, p, (n, p.R)), lwe_samples(rng, p.lwe_std_dev, p.big_n * p.ks.levels, n)
        gbsk, gksk = ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, bs, kss, load=False)
        ctx.set_decomposer_alignment(False)
        literal = ctx.bootstrap(lwe, tv)
    with oracle.decomposer_aligned(True):
        for i in range(lwe.shape[0]):
            assert np.array_equal(out[i], oracle.bootstrap(p, lwe[i], bsk, ksk, tv))
        for i in range(glwe.shape[0]):
            assert np.array_equal(ep[i], oracle.external_product(p, bsk[0], glwe[i]))
            assert np.array_equal(dg[i], oracle.decompose_glwe_ciphertext(glwe[i], p.pbs))
            assert np.array_equal(ksw[i], oracle.key_switch_lwe(big[i], p.big_n, p.n, p.ks, ksk))
        assert np.array_equal(dp, oracle.decompose(p.pbs, vals))
        assert np.array_equal(dk, oracle.decompose(p.ks, vals))
        ebsk, eksk = oracle.bootstrapping_key_gen_from_samples(p, lwe_sk, glwe_sk, bs, kss)
        assert np.array_equal(gbsk, ebsk) and np.array_equal(gksk, eksk)
    # back in literal mode the context is the reference again
    for i in range(lwe.shape[0]):
        assert np.array_equal(literal[i], oracle.bootstrap(p, lwe[i], bsk, ksk, tv))


def test_full_scale_cfg2_decrypts_in_aligned_mode(oracle):
    """BASELINE cfg2 (N=1024, k=1, n=630, l=3, logB=7) end to end on the device with the aligned
    decomposer: GPU keygen -> GPU encryption -> PBS (identity LUT) -> GPU decryption; every message
    must come back, and one bootstrap is compared bit for bit with the oracle in aligned mode."""
    import torch
    p = oracle.CFG2
    m = pkg()
    rng = np.random.default_rng(777)
    lwe_sk = rng.integers(0, 2, size=p.n).astype(np.uint32)
    glwe_sk = rng.integers(0, 2, size=(p.k, p.N)).astype(np.uint32)
    bs = glwe_samples(rng, p, (p.n, p.R))
    kss = lwe_samples(rng, p.lwe_std_dev, p.big_n * p.ks.levels, p.n)
    batch = 512
    msgs = rng.integers(0, 1 << p.log_p, size=batch).astype(np.uint32)
    pts = (msgs << (32 - p.log_p - p.padding_bits)).astype(np.uint32)
    ls = lwe_samples(rng, p.lwe_std_dev, batch, p.n)
    as_dev = lambda a: torch.from_numpy(a.view(np.int32)).to("cuda:0")  # noqa: E731
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.use_torch_stream()
        ctx.set_decomposer_alignment(True)
        d_bsk, d_ksk = ctx.bootstrapping_key_gen(lwe_sk, glwe_sk, as_dev(bs), as_dev(kss), load=True)
        d_ct = ctx.lwe_encrypt(lwe_sk, as_dev(ls), as_dev(pts))
        d_out = ctx.bootstrap(d_ct, as_dev(m.construct_identity_test_vector(to_pkg_params(p))))
        dec = ctx.lwe_decrypt(lwe_sk, d_out).cpu().numpy().view(np.uint32)
        out = d_out.cpu().numpy().view(np.uint32)
        ct = d_ct.cpu().numpy().view(np.uint32)
        bsk, ksk = d_bsk.cpu().numpy().view(np.uint32), d_ksk.cpu().numpy().view(np.uint32)
    shift = 32 - p.log_p - p.padding_bits
    decoded = ((dec.astype(np.uint64) + (1 << (shift - 1))) >> shift) & ((1 << p.log_p) - 1)
    assert np.array_equal(decoded.astype(np.uint32), msgs)
    with oracle.decomposer_aligned(True):
        assert np.array_equal(out[3], oracle.bootstrap(p, ct[3], bsk, ksk, m.construct_identity_test_vector(to_pkg_params(p))))


@pytest.mark.parametrize("k,logn,n,pbs,log_p", [(2, 9, 4, (4, 6), 2), (1, 10, 5, (7, 3), 2), (2, 11, 2, (8, 4), 4)])
def test_ks_first_order_matches_oracle(oracle, k, logn, n, pbs, log_p):
    p = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    m = pkg()
    rng = np.random.default_rng(logn + 7 * k)
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=60 + logn)
    big = rand_u32(rng, (5, p.big_n + 1))
    big2 = rand_u32(rng, (5, p.big_n + 1))
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.set_bootstrap_order(True)
        assert ctx.io_dim == p.big_n
        out = ctx.bootstrap(big, tv)
        gate = ctx.gate(m.GATE_NAND, big, big2)
        neg = ctx.lwe_not(big)
        ctx.set_bootstrap_order(False)
        small = rand_u32(rng, (2, n + 1))
        ref = ctx.bootstrap(small, tv)
    assert out.shape == big.shape
    nand_tv = oracle.construct_test_vector_boolean(p, lambda l, r: 1 - (l & r))
    for i in range(big.shape[0]):
        assert np.array_equal(out[i], oracle.bootstrap_ks_first(p, big[i], bsk, ksk, tv))
        c_in = (np.uint32(2) * big2[i] + big[i]).astype(np.uint32)
        assert np.array_equal(gate[i], oracle.bootstrap_ks_first(p, c_in, bsk, ksk, nand_tv))
    want_neg = (0 - big.astype(np.int64)).astype(np.uint32)
    want_neg[:, p.big_n] += np.uint32(1 << (32 - p.log_p - p.padding_bits))
    assert np.array_equal(neg, want_neg)
    for i in range(2):
        assert np.array_equal(ref[i], oracle.bootstrap(p, small[i], bsk, ksk, tv))


def test_three_input_adder_at_reference_noise_needs_ks_first(oracle):
    """The 2-bootstraps-per-bit adder (3-input gates, log_p = 3) with the reference's own noise
    parameters: in the KS-then-PBS order every wire of 16 instances decrypts correctly (the linear
    combination 4*c2 + 2*c1 + c0 only amplifies blind-rotation noise).  Ciphertexts live under the
    flattened GLWE key and have k*N+1 words."""
    import torch
    p = oracle.Params(2, 9, 16, oracle.Decomposer(4, 6), log_p=3)  # lwe_std_dev = reference default
    rng = oracle.Rng(8086)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    big_sk = glwe_sk.reshape(-1)
    m = pkg()
    gates = importlib.import_module("tfhe_research_amd.gates")
    circuit, out_wires = gates.full_adder_lut3(3)
    inst = 16
    nprng = np.random.default_rng(11)
    a, b = nprng.integers(0, 8, size=inst), nprng.integers(0, 8, size=inst)
    bits = np.array([[(a[i] >> j) & 1 for j in range(3)] + [(b[i] >> j) & 1 for j in range(3)] for i in range(inst)])
    cts = np.stack([np.stack([oracle.encrypt_lwe(p, big_sk, int(bit), rng) for bit in row]) for row in bits])
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.set_bootstrap_order(True)
        ctx.use_torch_stream()
        wires = gates.evaluate(ctx, circuit, torch.from_numpy(cts.view(np.int32)).to("cuda:0"))
        torch.cuda.synchronize()
        wires = wires.cpu().numpy().view(np.uint32)
        ctx.set_stream(None)
    assert wires.shape[-1] == p.big_n + 1
    for i in range(inst):
        got = [oracle.decrypt_lwe_message(p, big_sk, wires[i, w]) for w in range(circuit.n_wires)]
        assert got == circuit.evaluate_clear(bits[i].tolist()), i
        assert sum(got[w] << j for j, w in enumerate(out_wires)) == a[i] + b[i]

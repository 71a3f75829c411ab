"""GPU parity tests of the wider gate API: NOT, gates of m inputs by the recipe of
notes/Boolean Gates.md:2-11 (c_in = sum 2^i c_i, one PBS), MUX, and device-resident gate graphs that
use them.  Expected values: the oracle's construct_test_from_lut + bootstrap on the same c_in
(bit-exact) and decryption under real keys."""
import importlib
import itertools

import numpy as np
import pytest

from gpu_common import pkg, to_pkg_params

pytestmark = pytest.mark.gpu


def oracle_lut_gate(oracle, p, truth, cts, bsk, ksk):
    m = len(cts)
    c_in = np.zeros_like(cts[0])
    for i, ct in enumerate(cts):
        c_in = (c_in + (np.uint32(1 << i) * ct)).astype(np.uint32)
    lut = [truth[x & ((1 << m) - 1)] for x in range(1 << p.log_p)]
    return oracle.bootstrap(p, c_in, bsk, ksk, oracle.construct_test_from_lut(p, lut))


@pytest.fixture(scope="module")
def keyed(oracle):
    """reference cfg(test) parameters with a 3-bit plaintext space, real keys"""
    p = oracle.Params(2, 9, 8, oracle.Decomposer(4, 6), log_p=3)
    rng = oracle.Rng(31337)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    ctx = pkg().Context(to_pkg_params(p))
    ctx.load_bootstrapping_key(bsk, ksk)
    yield p, rng, lwe_sk, bsk, ksk, ctx
    ctx.close()


def test_three_input_gates_bit_exact_and_decrypt(oracle, keyed):
    p, rng, lwe_sk, bsk, ksk, ctx = keyed
    combos = list(itertools.product((0, 1), repeat=3))  # (x2, x1, x0)
    c2 = np.stack([oracle.encrypt_lwe(p, lwe_sk, x[0], rng) for x in combos])
    c1 = np.stack([oracle.encrypt_lwe(p, lwe_sk, x[1], rng) for x in combos])
    c0 = np.stack([oracle.encrypt_lwe(p, lwe_sk, x[2], rng) for x in combos])
    tables = {
        "xor3": tuple((i ^ (i >> 1) ^ (i >> 2)) & 1 for i in range(8)),
        "maj3": tuple(1 if bin(i).count("1") >= 2 else 0 for i in range(8)),
        "mux": tuple(((i >> 1) & 1) if (i >> 2) & 1 else (i & 1) for i in range(8)),
        "count": tuple(bin(i).count("1") for i in range(8)),  # a LUT gate with a 2-bit output
    }
    for name, truth in tables.items():
        out = ctx.lut_gate(truth, [c0, c1, c2])
        for j, x in enumerate(combos):
            want = oracle_lut_gate(oracle, p, truth, [c0[j], c1[j], c2[j]], bsk, ksk)
            assert np.array_equal(out[j], want), (name, x)
            assert oracle.decrypt_lwe_message(p, lwe_sk, out[j]) == truth[(x[0] << 2) | (x[1] << 1) | x[2]], (name, x)
    # two-input gates in the 3-bit space are the m = 2 case of the same call and of tfhe_gate_batch
    m = pkg()
    assert np.array_equal(ctx.lut_gate(m.GATE_XOR, [c0, c1]), ctx.gate(m.GATE_XOR, c0, c1))
    # one input: a programmable bootstrap that negates
    out = ctx.lut_gate((1, 0), [c0])
    for j, x in enumerate(combos):
        assert oracle.decrypt_lwe_message(p, lwe_sk, out[j]) == 1 - x[2]
        assert np.array_equal(out[j], oracle_lut_gate(oracle, p, (1, 0), [c0[j]], bsk, ksk))


def test_not_without_bootstrap(oracle, keyed):
    p, rng, lwe_sk, bsk, ksk, ctx = keyed
    cts = np.stack([oracle.encrypt_lwe(p, lwe_sk, b, rng) for b in (0, 1, 1, 0, 1)])
    out = ctx.lwe_not(cts)
    want = (0 - cts.astype(np.int64)).astype(np.uint32)
    want[:, p.n] += np.uint32(1 << (32 - p.log_p - p.padding_bits))
    assert np.array_equal(out, want)
    assert [oracle.decrypt_lwe_message(p, lwe_sk, c) for c in out] == [1, 0, 0, 1, 0]
    assert np.array_equal(ctx.lwe_not(out), cts)  # involution, bit for bit


def test_gate_input_count_is_bounded_by_the_plaintext_space(oracle):
    m = pkg()
    p = oracle.REF_TEST  # log_p = 2
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(p, 2, cfg_index=9)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        with pytest.raises(m.TfheError) as e:
            ctx.lut_gate((0,) * 8, [lwe, lwe, lwe])
        assert e.value.status == m.TFHE_ERR_INVALID_ARGUMENT
        with pytest.raises(m.TfheError):
            ctx.lut_gate((0, 1, 2, 4), [lwe, lwe])  # 4 is not < 2^log_p (test_vector.rs:41 / glwe.rs:144)


@pytest.mark.parametrize("log_p,builder", [(2, "ripple_carry_adder"), (3, "full_adder_lut3")])
def test_gate_graphs_with_not_mux_and_lut_gates(oracle, log_p, builder):
    """Device-resident graphs: an adder (5 two-input gates per bit at log_p = 2, two 3-input gates
    per bit at log_p = 3) followed by NOT / MUX gates; 16 instances under real keys, every wire
    decrypted and compared with the clear evaluation."""
    import torch
    # The reference's LWE noise (2^-16.2) is sized for a 2-bit space: after a key switch the error
    # has std ~2^24, and 4*c2 + 2*c1 + c0 of three bootstrapped inputs would sit 1.5 sigma from the
    # 3-bit decision boundary (2^27).  A parameter set meant for 3-input gates carries less noise.
    std = 0.000013071021089943935 if log_p == 2 else 2.0 ** -22
    p = oracle.Params(2, 9, 16, oracle.Decomposer(4, 6), log_p=log_p, lwe_std_dev=std)
    rng = oracle.Rng(4242 + log_p)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    gates = importlib.import_module("tfhe_research_amd.gates")
    circuit, out_wires = getattr(gates, builder)(3)
    # select between the low sum bit and its complement with the carry-out: exercises NOT and MUX
    inv = circuit.not_(out_wires[0])
    sel = circuit.mux(out_wires[-1], inv, out_wires[0])
    if log_p >= 3:
        sel3 = circuit.lut(tuple(((i >> 1) & 1) if (i >> 2) & 1 else (i & 1) for i in range(8)),
                           out_wires[-1], inv, out_wires[0])
    inst = 16
    nprng = np.random.default_rng(log_p)
    a, b = nprng.integers(0, 8, size=inst), nprng.integers(0, 8, size=inst)
    bits = np.array([[(a[i] >> j) & 1 for j in range(3)] + [(b[i] >> j) & 1 for j in range(3)] for i in range(inst)])
    cts = np.stack([np.stack([oracle.encrypt_lwe(p, lwe_sk, int(bit), rng) for bit in row]) for row in bits])
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        ctx.use_torch_stream()
        wires = gates.evaluate(ctx, circuit, torch.from_numpy(cts.view(np.int32)).to("cuda:0"))
        torch.cuda.synchronize()
        wires = wires.cpu().numpy().view(np.uint32)
        ctx.set_stream(None)
    for i in range(inst):
        clear = circuit.evaluate_clear(bits[i].tolist())
        got = [oracle.decrypt_lwe_message(p, lwe_sk, wires[i, w]) for w in range(circuit.n_wires)]
        assert got == clear, i
        assert sum(got[w] << j for j, w in enumerate(out_wires)) == a[i] + b[i]
        assert got[sel] == ((1 - got[out_wires[0]]) if got[out_wires[-1]] else got[out_wires[0]])
        if log_p >= 3:
            assert got[sel3] == got[sel]


def test_gate_graph_captured_into_one_hip_graph(oracle):
    """gates.GraphedCircuit: a whole adder + NOT/MUX evaluation (several truth tables per level,
    so the context's per-table test-vector cache is exercised) replayed as ONE HIP graph on two
    different input sets; bit-exact with the eager evaluation and correct after decryption."""
    import torch
    p = oracle.Params(2, 9, 16, oracle.Decomposer(4, 6))
    rng = oracle.Rng(600613)
    lwe_sk, glwe_sk, bsk, ksk = oracle.keygen(p, rng)
    m = pkg()
    gates = importlib.import_module("tfhe_research_amd.gates")
    circuit, out_wires = gates.ripple_carry_adder(3)
    inv = circuit.not_(out_wires[0])
    sel = circuit.mux(out_wires[-1], inv, out_wires[0])
    inst = 8
    nprng = np.random.default_rng(5)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        gc = gates.GraphedCircuit(ctx, circuit, inst, torch.device("cuda:0"))
        for trial in range(2):
            a, b = nprng.integers(0, 8, size=inst), nprng.integers(0, 8, size=inst)
            bits = np.array([[(a[i] >> j) & 1 for j in range(3)] + [(b[i] >> j) & 1 for j in range(3)] for i in range(inst)])
            cts = np.stack([np.stack([oracle.encrypt_lwe(p, lwe_sk, int(bit), rng) for bit in row]) for row in bits])
            d_in = torch.from_numpy(cts.view(np.int32)).to("cuda:0")
            graphed = gc(d_in).cpu().numpy().view(np.uint32).copy()
            with torch.cuda.stream(gc.stream):
                eager = gates.evaluate(ctx, circuit, d_in)
                gc.stream.synchronize()
            assert np.array_equal(graphed, eager.cpu().numpy().view(np.uint32))
            for i in range(inst):
                got = [oracle.decrypt_lwe_message(p, lwe_sk, graphed[i, w]) for w in range(circuit.n_wires)]
                assert got == circuit.evaluate_clear(bits[i].tolist()), (trial, i)
                assert sum(got[w] << j for j, w in enumerate(out_wires)) == a[i] + b[i]
        ctx.set_stream(None)

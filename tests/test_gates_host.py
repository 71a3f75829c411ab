"""CPU: the gate-graph description (tfhe-research_amd/gates.py) -- clear evaluation and levelling of
circuits with 2-input gates, NOT, MUX and multi-input LUT gates."""
import importlib
import itertools

from gpu_common import pkg


def gates_mod():
    pkg()
    return importlib.import_module("tfhe_research_amd.gates")


def test_adders_add_in_the_clear():
    g = gates_mod()
    for build in (g.ripple_carry_adder, g.full_adder_lut3):
        circuit, out = build(3)
        for a, b in itertools.product(range(8), repeat=2):
            bits = [(a >> j) & 1 for j in range(3)] + [(b >> j) & 1 for j in range(3)]
            w = circuit.evaluate_clear(bits)
            assert sum(w[o] << j for j, o in enumerate(out)) == a + b
    c5, _ = g.ripple_carry_adder(4)
    c2, _ = g.full_adder_lut3(4)
    assert len(c5.gates) == 2 + 5 * 3 and len(c2.gates) == 2 + 2 * 3


def test_not_mux_lut_and_levels():
    g = gates_mod()
    c = g.Circuit(3)
    n = c.not_(0)
    mx = c.mux(0, 1, 2)
    sel3 = c.lut(tuple((i >> 2) & 1 and (i >> 1) & 1 or (1 - ((i >> 2) & 1)) and i & 1 for i in range(8)), 0, 1, 2)
    x = c.gate("xor", n, mx)
    for bits in itertools.product((0, 1), repeat=3):
        w = c.evaluate_clear(list(bits))
        assert w[n] == 1 - bits[0]
        assert w[mx] == (bits[1] if bits[0] else bits[2])
        assert w[sel3] == w[mx]            # the same multiplexer as one 3-input LUT
        assert w[x] == w[n] ^ w[mx]
    assert c.levels() == [[0, 1, 2], [3]]


def test_system_rng_is_the_default_source_of_key_material():
    """generate_keys / encrypt_bits draw from the OS CSPRNG unless the test hook rng= is passed (the
    reference requires R: CryptoRng + RngCore, lwe.rs:55): SystemRng offers exactly the three numpy
    Generator methods they use, with the right ranges, shapes and moments, and never repeats."""
    import numpy as np
    from gpu_common import pkg
    m = pkg()
    r = m.SystemRng()
    bits = r.integers(0, 2, size=(4, 4096))
    assert bits.shape == (4, 4096) and set(np.unique(bits)) == {0, 1} and abs(bits.mean() - 0.5) < 0.02
    words = r.integers(0, 1 << 32, size=100000, dtype=np.uint64)
    assert words.dtype == np.uint64 and int(words.max()) < 1 << 32 and int(words.max()) > 1 << 31
    assert abs(words.astype(np.float64).mean() / 2.0 ** 32 - 0.5) < 0.01
    z = r.normal(0.0, 3.0, size=(200000,))
    assert abs(z.mean()) < 0.05 and abs(z.std() - 3.0) < 0.05
    assert not np.array_equal(r.integers(0, 1 << 32, size=64, dtype=np.uint64), r.integers(0, 1 << 32, size=64, dtype=np.uint64))
    import inspect
    src = inspect.getsource(m.Context.generate_keys) + inspect.getsource(m.Context.encrypt_bits)
    assert "SystemRng()" in src and "default_rng" not in src


class _ClearContext:
    """Stands in for a Context on CPU tensors: a 'ciphertext' is one word holding the bit (width 1), a test vector is
    the gate's truth table in its first four words, bootstrap(x, tvs) = tvs[row][x[row]] -- enough to check that the
    planner feeds every gate the right operands, order (2 * ct1 + ct0, boolean.rs:18) and table, merged or not."""
    io_dim = 0

    class params:
        N = 8

    def __init__(self):
        self.calls = []

    def lwe_linear(self, c0, ct0, c1=0, ct1=None, out=None):
        return c0 * ct0 + (c1 * ct1 if ct1 is not None else 0)

    def bootstrap(self, x, tvs):
        import torch
        self.calls.append(("bootstrap", int(x.shape[0])))
        rows = torch.arange(x.shape[0])
        return tvs[rows, x[:, 0].long()].reshape(-1, 1).to(x.dtype)

    def gate(self, truth, ct0, ct1, out=None):
        import torch
        self.calls.append(("gate", int(ct0.shape[0])))
        t = torch.tensor(truth, dtype=ct0.dtype)
        return t[(2 * ct1 + ct0)[:, 0].long()].reshape(-1, 1)


def test_a_levels_two_input_gates_share_one_bootstrap_call(monkeypatch):
    """gates.plan merges all two-input gates of a level into ONE step (per-row test vectors); the merged evaluation equals
    the clear evaluation for every input of a 4-bit adder; above MERGE_LIMIT rows the per-truth-table calls come back"""
    import numpy as np
    import torch
    g = gates_mod()
    circuit, out = g.ripple_carry_adder(4)
    steps = g.plan(circuit, torch.device("cpu"))
    assert len(steps) == len(circuit.levels())              # one step per level: nothing but two-input gates here
    assert any(st.kind == "gate2" and len(st.parts) >= 2 for st in steps)
    ctx = _ClearContext()
    table = torch.zeros((len(g._KINDS), 8), dtype=torch.int32)
    for i, kind in enumerate(g._KINDS):
        table[i, :4] = torch.tensor(g.TRUTH[kind], dtype=torch.int32)
    ctx._gate_tv_table = table
    inputs = torch.tensor([[(a >> j) & 1 for j in range(4)] + [(b >> j) & 1 for j in range(4)]
                           for a in range(16) for b in range(16)], dtype=torch.int32).reshape(256, 8, 1)
    wires = g.evaluate(ctx, circuit, inputs, steps)[:, :, 0].numpy()
    for row, bits in zip(wires, inputs[:, :, 0].tolist()):
        assert row.tolist() == circuit.evaluate_clear(bits)
    merged_calls = [c for c in ctx.calls if c[0] == "bootstrap"]
    assert len(merged_calls) == sum(1 for st in steps if st.kind == "gate2")
    # a full chip: the same plan falls back to one call per truth table
    monkeypatch.setattr(g, "MERGE_LIMIT", 16)
    ctx2 = _ClearContext()
    ctx2._gate_tv_table = table
    wires2 = g.evaluate(ctx2, circuit, inputs, steps)[:, :, 0].numpy()
    assert np.array_equal(wires, wires2) and not any(c[0] == "bootstrap" for c in ctx2.calls)

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

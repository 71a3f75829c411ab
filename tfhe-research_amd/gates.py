"""Device-resident evaluation of boolean gate graphs.

The reference offers two gates, and()/or() (boolean.rs:9-53), each of which is one programmable
bootstrap of 2*ct1 + ct0 with a test vector built from a closure (test_vector.rs:5-20).  Any 2-input
gate is the same bootstrap with another truth table, and a gate's output is a fresh encryption of a
bit, so gates chain.  This module evaluates a whole graph of such gates with every ciphertext
staying in HBM: wires live in one device tensor, each level's gates that share a truth table go to
the GPU as ONE batched call of the C ABI's tfhe_gate_batch_device.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

TRUTH = {  # truth[(lhs << 1) | rhs], lhs = bit of ct1, rhs = bit of ct0 (boolean.rs:18)
    "and": (0, 0, 0, 1),
    "or": (0, 1, 1, 1),
    "nand": (1, 1, 1, 0),
    "nor": (1, 0, 0, 0),
    "xor": (0, 1, 1, 0),
    "xnor": (1, 0, 0, 1),
}


@dataclass
class Circuit:
    """Wires 0..n_inputs-1 are the inputs; every gate appends one wire."""
    n_inputs: int
    gates: List[Tuple[str, int, int]] = field(default_factory=list)  # (kind, wire of ct1, wire of ct0)

    def gate(self, kind: str, lhs: int, rhs: int) -> int:
        assert kind in TRUTH and max(lhs, rhs) < self.n_wires
        self.gates.append((kind, lhs, rhs))
        return self.n_wires - 1

    @property
    def n_wires(self) -> int:
        return self.n_inputs + len(self.gates)

    def levels(self) -> List[List[int]]:
        """Gate indices grouped by depth (a gate's level = 1 + max level of its operands)."""
        depth = [0] * self.n_inputs
        by_level: Dict[int, List[int]] = {}
        for g, (_, a, b) in enumerate(self.gates):
            d = 1 + max(depth[a], depth[b])
            depth.append(d)
            by_level.setdefault(d, []).append(g)
        return [by_level[d] for d in sorted(by_level)]

    def evaluate_clear(self, bits: Sequence[int]) -> List[int]:
        w = list(bits)
        for kind, a, b in self.gates:
            w.append(TRUTH[kind][(w[a] << 1) | w[b]])
        return w


def ripple_carry_adder(width: int) -> Tuple[Circuit, List[int]]:
    """a[0..w) + b[0..w) (little endian) -> wires of the w+1 sum bits.  5 gates per full adder."""
    c = Circuit(2 * width)
    a = list(range(width))
    b = list(range(width, 2 * width))
    out = []
    carry = None
    for i in range(width):
        axb = c.gate("xor", a[i], b[i])
        if carry is None:
            out.append(axb)
            carry = c.gate("and", a[i], b[i])
        else:
            out.append(c.gate("xor", axb, carry))
            t1 = c.gate("and", a[i], b[i])
            t2 = c.gate("and", axb, carry)
            carry = c.gate("or", t1, t2)
    out.append(carry)
    return c, out


def evaluate(ctx, circuit: Circuit, inputs):
    """inputs: device tensor [instances][n_inputs][n+1] (32-bit) of LWE encryptions of bits, one row
    of wires per independent instance of the circuit.  Returns [instances][n_wires][n+1] on the
    same device.  All instances advance level by level; per level and truth table one batched gate
    call covers instances x gates ciphertext pairs."""
    import torch
    inst, n_in, width = inputs.shape
    assert n_in == circuit.n_inputs and width == ctx.params.n + 1
    wires = torch.empty((inst, circuit.n_wires, width), dtype=inputs.dtype, device=inputs.device)
    wires[:, :n_in] = inputs
    for level in circuit.levels():
        by_kind: Dict[str, List[int]] = {}
        for g in level:
            by_kind.setdefault(circuit.gates[g][0], []).append(g)
        for kind, gs in by_kind.items():
            lhs = torch.tensor([circuit.gates[g][1] for g in gs], device=inputs.device)
            rhs = torch.tensor([circuit.gates[g][2] for g in gs], device=inputs.device)
            dst = torch.tensor([circuit.n_inputs + g for g in gs], device=inputs.device)
            ct1 = wires.index_select(1, lhs).reshape(-1, width).contiguous()
            ct0 = wires.index_select(1, rhs).reshape(-1, width).contiguous()
            out = ctx.gate(TRUTH[kind], ct0, ct1)
            wires[:, dst] = out.reshape(inst, len(gs), width)
    return wires

"""Device-resident evaluation of boolean gate graphs.

The reference offers two gates, and()/or() (boolean.rs:9-53), each of which is one programmable
bootstrap of 2*ct1 + ct0 with a test vector built from a closure (test_vector.rs:5-20).  Any 2-input
gate is the same bootstrap with another truth table, a gate of m inputs is the bootstrap of
sum_i 2^i * ct_i in a plaintext space of m bits (notes/Boolean Gates.md:2-11), and a gate's output
is a fresh encryption of a bit, so gates chain.  This module evaluates a whole graph of such gates
with every ciphertext staying in HBM: wires live in one device tensor, and ALL two-input gates of a
level -- whatever their truth tables -- go to the GPU as ONE bootstrap call with one test vector per
ciphertext (tfhe_lwe_linear_batch_device for 2*ct1 + ct0, then tfhe_bootstrap_batch_device with
tv[batch][N]: the reference's bootstrap() takes the test vector as an argument, bootstrapping.rs:64).
A narrow circuit's level is a few hundred ciphertexts -- far fewer than the chip holds -- so what a level
costs is the LATENCY of one blind rotation, and calls that run one after the other each pay it again:
merged, a level of a 16-bit adder over 64 instances costs one rotation instead of up to three
(profiles/r04_gate_graph_adder16_cfg3.txt).  Wide levels (more rows than MERGE_LIMIT) keep one call per
truth table: the chip is full either way and a shared test vector saves the per-row table.

Gate kinds
  and / or / nand / nor / xor / xnor   one PBS (tfhe_gate_batch_device)
  not                                  no PBS: (-a, enc(1) - b) (tfhe_lwe_not_batch_device)
  mux(sel, a, b) = sel ? a : b         two PBS and one addition at any log_p >= 2:
                                       AND(sel, a) + AND(NOT sel, b) -- the two terms are never both
                                       1, so their sum already encodes the OR
  lut(truth, x_{m-1}, ..., x_0)        one PBS of m inputs; needs a context with log_p >= m
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

TRUTH = {  # truth[(lhs << 1) | rhs], lhs = bit of ct1, rhs = bit of ct0 (boolean.rs:18)
    "and": (0, 0, 0, 1),
    "or": (0, 1, 1, 1),
    "nand": (1, 1, 1, 0),
    "nor": (1, 0, 0, 0),
    "xor": (0, 1, 1, 0),
    "xnor": (1, 0, 0, 1),
}
_ANDNOT = (0, 1, 0, 0)  # (NOT lhs) AND rhs
_KINDS = tuple(TRUTH)      # row order of the per-context test-vector table
MERGE_LIMIT = 8192         # rows (instances x gates of a level) up to which a level's two-input gates share one call


@dataclass
class Circuit:
    """Wires 0..n_inputs-1 are the inputs; every gate appends one wire.  A two-input gate is the
    tuple (kind, wire of ct1, wire of ct0); `not` is (kind, wire), `mux` (kind, sel, a, b) and `lut`
    (kind, operands most significant first..., ) with its truth table in `luts[gate index]`."""
    n_inputs: int
    gates: List[Tuple] = field(default_factory=list)
    luts: Dict[int, Tuple[int, ...]] = field(default_factory=dict)

    def _add(self, entry: Tuple) -> int:
        assert max(entry[1:]) < self.n_wires
        self.gates.append(entry)
        return self.n_wires - 1

    def gate(self, kind: str, lhs: int, rhs: int) -> int:
        assert kind in TRUTH
        return self._add((kind, lhs, rhs))

    def not_(self, a: int) -> int:
        return self._add(("not", a))

    def mux(self, sel: int, a: int, b: int) -> int:
        return self._add(("mux", sel, a, b))

    def lut(self, truth: Sequence[int], *operands: int) -> int:
        """operands[0] is the leftmost (most significant) input: truth[(x_{m-1} ... x_0)_2]."""
        assert len(truth) == 1 << len(operands)
        self.luts[len(self.gates)] = tuple(int(v) for v in truth)
        return self._add(("lut",) + tuple(operands))

    @property
    def n_wires(self) -> int:
        return self.n_inputs + len(self.gates)

    def levels(self) -> List[List[int]]:
        """Gate indices grouped by depth (a gate's level = 1 + max level of its operands)."""
        depth = [0] * self.n_inputs
        by_level: Dict[int, List[int]] = {}
        for g, entry in enumerate(self.gates):
            d = 1 + max(depth[w] for w in entry[1:])
            depth.append(d)
            by_level.setdefault(d, []).append(g)
        return [by_level[d] for d in sorted(by_level)]

    def evaluate_clear(self, bits: Sequence[int]) -> List[int]:
        w = list(bits)
        for g, entry in enumerate(self.gates):
            kind, ops = entry[0], entry[1:]
            if kind == "not":
                w.append(1 - w[ops[0]])
            elif kind == "mux":
                w.append(w[ops[1]] if w[ops[0]] else w[ops[2]])
            elif kind == "lut":
                idx = 0
                for o in ops:
                    idx = (idx << 1) | w[o]
                w.append(self.luts[g][idx])
            else:
                w.append(TRUTH[kind][(w[ops[0]] << 1) | w[ops[1]]])
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


def full_adder_lut3(width: int) -> Tuple[Circuit, List[int]]:
    """The same adder with three-input gates (needs log_p >= 3): sum = XOR3, carry = MAJ3 -- two
    bootstraps per bit instead of five."""
    xor3 = tuple((i ^ (i >> 1) ^ (i >> 2)) & 1 for i in range(8))
    maj3 = tuple(1 if bin(i).count("1") >= 2 else 0 for i in range(8))
    c = Circuit(2 * width)
    out = []
    carry: Optional[int] = None
    for i in range(width):
        a, b = i, width + i
        if carry is None:
            out.append(c.gate("xor", a, b))
            carry = c.gate("and", a, b)
        else:
            out.append(c.lut(xor3, a, b, carry))
            carry = c.lut(maj3, a, b, carry)
    out.append(carry)
    return c, out


@dataclass
class _Step:
    kind: str
    truth: Optional[Tuple[int, ...]]
    operands: list          # device index tensors, one per operand position (as listed in the gate)
    dst: object             # device index tensor of the output wires
    count: int
    kinds: object = None    # "gate2" steps: device tensor, row of the test-vector table (index into _KINDS) per gate
    parts: list = None      # "gate2" steps: the per-truth-table steps they replace (used above MERGE_LIMIT rows)
    tv_rows: dict = field(default_factory=dict)   # instances -> per-row table indices (kinds repeated per instance)


def plan(circuit: Circuit, device) -> List[_Step]:
    """Index tensors of every (level, gate kind) group, built once: evaluation itself then issues
    no host-to-device copies and can be captured into a HIP graph."""
    import torch
    steps: List[_Step] = []
    for level in circuit.levels():
        groups: Dict[Tuple, List[int]] = {}
        for g in level:
            kind = circuit.gates[g][0]
            key = (kind, len(circuit.gates[g]) - 1, circuit.luts.get(g))
            groups.setdefault(key, []).append(g)
        two_input: List[_Step] = []
        for (kind, arity, truth), gs in groups.items():
            ops = [torch.tensor([circuit.gates[g][pos] for g in gs], device=device) for pos in range(1, arity + 1)]
            dst = torch.tensor([circuit.n_inputs + g for g in gs], device=device)
            st = _Step(kind, truth, ops, dst, len(gs))
            (two_input if kind in TRUTH else steps).append(st)
        if len(two_input) == 1:
            steps.append(two_input[0])
        elif two_input:
            # one step for all two-input gates of the level: operands and outputs concatenated, one table row per gate
            merged = _Step("gate2", None, [torch.cat([st.operands[pos] for st in two_input]) for pos in range(2)],
                           torch.cat([st.dst for st in two_input]), sum(st.count for st in two_input),
                           kinds=torch.tensor([_KINDS.index(st.kind) for st in two_input for _ in range(st.count)], device=device),
                           parts=two_input)
            steps.append(merged)
    return steps


def _tv_table(ctx, device):
    """[len(_KINDS)][N] un-encoded test vectors of the six two-input truth tables (test_vector.rs:5-20), on the device,
    built once per context"""
    import numpy as np
    import torch
    from . import construct_test_vector_boolean
    table = getattr(ctx, "_gate_tv_table", None)
    if table is None or table.device != device:
        rows = np.stack([construct_test_vector_boolean(ctx.params, TRUTH[k]) for k in _KINDS]).astype(np.uint32)
        table = torch.from_numpy(rows.view(np.int32)).to(device)
        ctx._gate_tv_table = table
    return table


def _run(ctx, steps: List[_Step], wires):
    inst, _, width = wires.shape

    def operand(step, pos):
        return wires.index_select(1, step.operands[pos]).reshape(-1, width).contiguous()

    def gate2(st):
        return ctx.gate(TRUTH[st.kind], operand(st, 1), operand(st, 0))

    for st in steps:
        if st.kind == "gate2":
            if inst * st.count > MERGE_LIMIT:   # a full chip: one call per truth table, one shared test vector each
                for part in st.parts:
                    wires[:, part.dst] = gate2(part).reshape(inst, part.count, width)
                continue
            rows = st.tv_rows.get(inst)
            if rows is None:
                rows = st.tv_rows[inst] = st.kinds.repeat(inst)     # row (instance, gate): instance-major like operand()
            tvs = _tv_table(ctx, wires.device).index_select(0, rows)
            # boolean.rs:18: ct_in = 2 * ct1 + ct0 (gate tuple: (kind, wire of ct1, wire of ct0))
            out = ctx.bootstrap(ctx.lwe_linear(1, operand(st, 1), 2, operand(st, 0)), tvs)
        elif st.kind == "not":
            out = ctx.lwe_not(operand(st, 0))
        elif st.kind == "mux":
            sel, a, b = operand(st, 0), operand(st, 1), operand(st, 2)
            t1 = ctx.gate(TRUTH["and"], a, sel)
            t2 = ctx.gate(_ANDNOT, b, sel)
            out = ctx.lwe_linear(1, t1, 1, t2)
        elif st.kind == "lut":
            # operands are listed most significant first; the ABI takes cts[0] = least significant
            out = ctx.lut_gate(st.truth, [operand(st, pos) for pos in range(len(st.operands) - 1, -1, -1)])
        else:
            out = ctx.gate(TRUTH[st.kind], operand(st, 1), operand(st, 0))
        wires[:, st.dst] = out.reshape(inst, st.count, width)
    return wires


def evaluate(ctx, circuit: Circuit, inputs, steps: Optional[List[_Step]] = None):
    """inputs: device tensor [instances][n_inputs][words] (32-bit) of LWE encryptions of bits, one
    row of wires per independent instance of the circuit (words = ctx.io_dim + 1).  Returns
    [instances][n_wires][words] on the same device.  All instances advance level by level; per level
    and truth table one batched gate call covers instances x gates ciphertexts."""
    import torch
    inst, n_in, width = inputs.shape
    assert n_in == circuit.n_inputs and width == ctx.io_dim + 1
    wires = torch.empty((inst, circuit.n_wires, width), dtype=inputs.dtype, device=inputs.device)
    wires[:, :n_in] = inputs
    return _run(ctx, steps if steps is not None else plan(circuit, inputs.device), wires)


class GraphedCircuit:
    """A circuit for a fixed number of instances captured into ONE HIP graph: after the first call
    a whole evaluation -- every gather, linear combination, blind rotation, key switch and
    scatter of every level -- is a single graph launch.

        gc = GraphedCircuit(ctx, circuit, instances, device)
        wires = gc(inputs)        # [instances][n_wires][words]; the buffer is reused by the next call
    """

    def __init__(self, ctx, circuit: Circuit, instances: int, device, dtype=None):
        import torch
        self.ctx, self.circuit = ctx, circuit
        width = ctx.io_dim + 1
        dtype = dtype or torch.int32
        self.inputs = torch.zeros((instances, circuit.n_inputs, width), dtype=dtype, device=device)
        self.wires = torch.empty((instances, circuit.n_wires, width), dtype=dtype, device=device)
        self.steps = plan(circuit, device)
        self.stream = torch.cuda.Stream(device=device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.stream):
            ctx.use_torch_stream()
            # eager warm-up: sizes the context's workspace, uploads every truth table's test vector
            # and sets the kernels' one-time attributes -- none of which may happen while capturing
            self.wires[:, :circuit.n_inputs] = self.inputs
            _run(ctx, self.steps, self.wires)
            self.stream.synchronize()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.wires[:, :circuit.n_inputs] = self.inputs
                _run(ctx, self.steps, self.wires)

    def __call__(self, inputs):
        """Stream-ordered like a torch op: the graph's stream first waits for the caller's current
        stream (the kernels that produced `inputs`), and the caller's stream then waits for the
        replay, so `wires` can be consumed on it without a host synchronisation."""
        import torch
        caller = torch.cuda.current_stream(self.inputs.device)
        self.stream.wait_stream(caller)
        with torch.cuda.stream(self.stream):
            self.inputs.copy_(inputs, non_blocking=True)
            self.graph.replay()
        caller.wait_stream(self.stream)
        return self.wires

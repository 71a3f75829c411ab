"""Gate-graph throughput on the device: a 16-bit ripple-carry adder over many
independent instances at the reference's default parameters (cfg3: N=512, k=2, n=722), evaluated
eagerly (one ABI call per level and truth table) and as ONE captured HIP graph.
    python tools/gate_graph_bench.py > gpurun_out/gate_graph.txt
"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from gpu_common import to_pkg_params  # noqa: E402
from oracle import oracle  # noqa: E402

m = entry.load_package()
gates = importlib.import_module("tfhe_research_amd.gates")
dev = torch.device("cuda:0")
p = oracle.CFG3
bits = 16
circuit, outs = gates.ripple_carry_adder(bits)
pbs_gates = len(circuit.gates)
print(f"# {bits}-bit ripple-carry adder: {pbs_gates} gates (one PBS each), depth {len(circuit.levels())}; cfg3 parameters, synthetic keys and ciphertexts")
print("# instances  eager_ms  graph_ms  gates_per_s(graph)")
bsk = torch.randint(-2**31, 2**31 - 1, p.bsk_shape(), dtype=torch.int32, device=dev)
ksk = torch.randint(-2**31, 2**31 - 1, p.ksk_shape(), dtype=torch.int32, device=dev)
shape = {"auto": m.SHAPE_AUTO, "wide": m.SHAPE_WIDE, "team": m.SHAPE_TEAM}[os.environ.get("GATE_GRAPH_SHAPE", "auto")]
print(f"# kernel shape: {os.environ.get('GATE_GRAPH_SHAPE', 'auto')} (tfhe_context_set_kernel_shape)")
for inst in (64, 1024, 4096):
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.set_kernel_shape(shape)
        ctx.load_bootstrapping_key(bsk, ksk)
        x = torch.randint(-2**31, 2**31 - 1, (inst, circuit.n_inputs, p.n + 1), dtype=torch.int32, device=dev)
        gc = gates.GraphedCircuit(ctx, circuit, inst, dev)
        with torch.cuda.stream(gc.stream):
            steps = gates.plan(circuit, dev)
            gates.evaluate(ctx, circuit, x, steps)
            gc.stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                gates.evaluate(ctx, circuit, x, steps)
            gc.stream.synchronize()
            eager = (time.perf_counter() - t0) / 3 * 1e3
        gc(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            gc(x)
        torch.cuda.synchronize()   # (round 4: without this the "graph" column timed three graph LAUNCHES, not three replays)
        graph = (time.perf_counter() - t0) / 3 * 1e3
        print(f"{inst} {eager:.2f} {graph:.2f} {inst * pbs_gates / (graph * 1e-3):.0f}")
        ctx.set_stream(None)

"""GPU (-m gpu): the WIDE team -- the latency shape of the blind rotation (pbs_wave.h::blind_rotate_team_wide,
kernels.hip::blind_rotate_wide_kernel): 2 (k+1) waves per sample, split by digit level and key part.  The reference's
call shape is one ciphertext per bootstrap() (bootstrapping.rs:58-65) and one pair per gate (boolean.rs:9-37); the launcher
picks this kernel for batches that leave most of the chip idle.  It must return the words of the throughput kernel, of
the committed golden fixtures and of the oracle; both shapes are forced here (tfhe_context_set_kernel_shape) so that
neither hides behind the launcher's choice."""
import numpy as np
import pytest

import golden_common as gc
from gpu_common import pkg, rand_u32, to_pkg_params
from test_gpu_golden import SETS, pkg_params

pytestmark = pytest.mark.gpu

SHAPES = [
    (1, 10, 4, (7, 3), 2),    # cfg2's shape: levels 2 + 1 over the halves, the literal (misaligned) decomposer
    (1, 10, 3, (8, 4), 2),    # every CMUX depends on the key
    (2, 9, 4, (4, 6), 2),     # the reference's default shape (cfg3): six waves, 18 digit rows
    (1, 9, 5, (8, 2), 2),     # cfg1's shape
    (2, 10, 3, (8, 3), 2),    # N = 1024 with k = 2: six waves, nine rows
    (1, 10, 3, (9, 1), 2),    # ONE level: the second half has no forward transform
    (1, 9, 3, (5, 5), 4),     # five levels, log_p = 4
]


@pytest.mark.parametrize("shape_name", ["wide", "team"])
@pytest.mark.parametrize("name", SETS)
def test_golden_trace_under_both_kernel_shapes(name, shape_name):
    """bootstrap and blind rotation of the 8 golden rows with the shape forced: every output word as committed"""
    m = pkg()
    pd, a = gc.load_set(name)
    with m.Context(pkg_params(pd)) as ctx:
        assert ctx.backend == "fp64-fft"
        ctx.set_kernel_shape({"wide": m.SHAPE_WIDE, "team": m.SHAPE_TEAM}[shape_name])
        ctx.load_bootstrapping_key(a["bsk"], a["ksk"])
        plan = ctx.blind_rotate_plan(8)
        assert plan["kernel"].startswith(shape_name), plan
        assert np.array_equal(ctx.bootstrap(a["lwe_in"], a["tv"]), a["lwe_out"])
        assert np.array_equal(ctx.blind_rotate(a["lwe_in"], a["tv"]), a["acc_final"])
        assert np.array_equal(ctx.bootstrap(a["lwe_in"][:1], a["tv"]), a["lwe_out"][:1])   # ONE ciphertext: the reference's call


@pytest.mark.parametrize("k,logn,n,pbs,log_p", SHAPES)
def test_wide_team_vs_oracle_and_team(oracle, k, logn, n, pbs, log_p):
    """ragged batch of 5 with per-sample test vectors, a~ = 0 and b~ -> 2N rows: wide == team == oracle, GLWE accumulator and
    bootstrap output; aligned decomposer too (where log2 B does not divide 32 that is the data-dependent case)"""
    m = pkg()
    p = oracle.Params(k, logn, n, oracle.Decomposer(*pbs), log_p=log_p)
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(p, 5, cfg_index=150 + logn + k)
    rng = np.random.default_rng(logn * 7 + k)
    tvs = rng.integers(0, 1 << log_p, size=(5, p.N)).astype(np.uint32)
    lwe = lwe.copy()
    lwe[0, 0] = 0
    lwe[1, n] = 0xFFFFFFFF
    for aligned in (False, True):
        got = {}
        for shape in (m.SHAPE_WIDE, m.SHAPE_TEAM):
            with m.Context(to_pkg_params(p), backend=m.BACKEND_FP64_FFT) as ctx:
                ctx.set_kernel_shape(shape)
                ctx.set_decomposer_alignment(aligned)
                ctx.load_bootstrapping_key(bsk, ksk)
                got[shape] = (ctx.bootstrap(lwe, tvs), ctx.blind_rotate(lwe, tvs))
        assert np.array_equal(got[m.SHAPE_WIDE][0], got[m.SHAPE_TEAM][0])
        assert np.array_equal(got[m.SHAPE_WIDE][1], got[m.SHAPE_TEAM][1])
        with oracle.decomposer_aligned(aligned):
            for b in range(5):
                want, tr = oracle.bootstrap(p, lwe[b], bsk, ksk, tvs[b], trace=True)
                assert np.array_equal(got[m.SHAPE_WIDE][0][b], want), (aligned, b)
                assert np.array_equal(got[m.SHAPE_WIDE][1][b], tr["acc_final"]), (aligned, b)


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3", "cfg1"])
def test_wide_team_at_full_size(oracle, cfg):
    """BASELINE cfg2 / cfg3 / cfg1 at their full key length, 96 random ciphertexts, aligned decomposer where the base does not
    divide 32 (cfg2: otherwise nothing depends on the key): the wide team's words equal the throughput team's, rows 0, 47, 95
    the oracle's; and a single ciphertext -- the reference's call -- equals its row of the batch"""
    m = pkg()
    p = {"cfg1": oracle.CFG1, "cfg2": oracle.CFG2, "cfg3": oracle.CFG3}[cfg]
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=2)
    rng = np.random.default_rng(91)
    lwe = rand_u32(rng, (96, p.n + 1))
    aligned = p.pbs.log_base * (32 // p.pbs.log_base) != 32
    out = {}
    for shape in (m.SHAPE_WIDE, m.SHAPE_TEAM):
        with m.Context(to_pkg_params(p)) as ctx:
            ctx.set_kernel_shape(shape)
            ctx.set_decomposer_alignment(aligned)
            ctx.load_bootstrapping_key(bsk, ksk)
            out[shape] = ctx.bootstrap(lwe, tv)
            if shape == m.SHAPE_WIDE:
                one = ctx.bootstrap(lwe[47:48], tv)
    assert np.array_equal(out[m.SHAPE_WIDE], out[m.SHAPE_TEAM])
    assert np.array_equal(one[0], out[m.SHAPE_WIDE][47])
    with oracle.decomposer_aligned(aligned):
        for b in (0, 47, 95):
            assert np.array_equal(out[m.SHAPE_WIDE][b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


def test_launcher_picks_the_wide_team_for_small_batches_only(oracle):
    """TFHE_SHAPE_AUTO: batches that leave most CUs idle get the wide team, the headline batch the throughput team; the
    prime-field backends and N = 2048 only have the team; a forced shape is honoured; the plan says which"""
    m = pkg()
    with m.Context(to_pkg_params(oracle.CFG2)) as ctx:
        small, big = ctx.blind_rotate_plan(1), ctx.blind_rotate_plan(4096)
        assert small["kernel"].startswith("wide") and small["waves_per_team"] == 4 and small["launches"] == 1, small
        assert big["kernel"] == "team" and big["waves_per_team"] == 2 and big["streams"] == 2, big
        ctx.set_kernel_shape(m.SHAPE_TEAM)
        assert ctx.blind_rotate_plan(1)["kernel"] == "team"
        ctx.set_kernel_shape(m.SHAPE_WIDE)
        assert ctx.blind_rotate_plan(4096)["kernel"].startswith("wide")
        with pytest.raises(m.TfheError):
            ctx.set_kernel_shape(7)
    with m.Context(to_pkg_params(oracle.CFG2), backend=m.BACKEND_FP64) as ctx:
        ctx.set_kernel_shape(m.SHAPE_WIDE)            # not offered for this backend: the team runs
        assert ctx.blind_rotate_plan(1)["kernel"] == "team"
    with m.Context(to_pkg_params(oracle.CFG5)) as ctx:
        ctx.set_kernel_shape(m.SHAPE_WIDE)
        assert ctx.blind_rotate_plan(1)["kernel"] == "team" and ctx.blind_rotate_plan(1)["waves_per_team"] == 12


def test_wide_team_gates_and_graph_capture(oracle):
    """a 2-bit adder circuit at width 8 (every level a small batch: the wide team's case) eagerly and as one HIP graph,
    NAND gates against the oracle's boolean_gate"""
    import importlib
    import torch
    m = pkg()
    gates = importlib.import_module("tfhe_research_amd.gates")
    p = oracle.Params(2, 9, 6, oracle.Decomposer(4, 6))
    lwe, bsk, ksk, _ = oracle.synthetic_inputs(p, 8, cfg_index=171)
    ct1 = rand_u32(np.random.default_rng(3), lwe.shape)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        assert ctx.blind_rotate_plan(8)["kernel"].startswith("wide")
        got = ctx.gate(m.GATE_NAND, lwe, ct1)
        for b in (0, 7):
            assert np.array_equal(got[b], oracle.boolean_gate(p, lambda l, r: 1 - (l & r), lwe[b], ct1[b], bsk, ksk)), b
        dev = torch.device("cuda", 0)
        circuit, _ = gates.ripple_carry_adder(2)
        x = torch.from_numpy(rand_u32(np.random.default_rng(4), (8, circuit.n_inputs, p.n + 1)).view(np.int32)).to(dev)
        gc_ = gates.GraphedCircuit(ctx, circuit, 8, dev)
        graphed = gc_(x).cpu().numpy().copy()
        with torch.cuda.stream(gc_.stream):
            eager = gates.evaluate(ctx, circuit, x)
            gc_.stream.synchronize()
        assert np.array_equal(graphed, eager.cpu().numpy())
        # the same circuit under the throughput team: same words
        ctx.set_kernel_shape(m.SHAPE_TEAM)
        with torch.cuda.stream(gc_.stream):
            team = gates.evaluate(ctx, circuit, x)
            gc_.stream.synchronize()
        assert np.array_equal(graphed, team.cpu().numpy())
        ctx.set_stream(None)


def test_pair_kernel_takes_over_above_the_team_capacity(oracle):
    """N = 512, k = 1 (cfg1's shape): batches larger than the chip holds as two-wave teams run the PAIR kernel -- one wave per
    sample, both polynomials side by side in it (pbs_wave.h::blind_rotate_pair) -- smaller ones the team, the smallest the wide
    team; all three read ONE prepared key (laid out for the pair kernel) and must return the same words.  1,700 rows through
    the pair kernel (segmented, two streams) against the same rows in slices of 100 through the team and of 10 through the wide
    team; rows 0, 1 (b~ -> 2N), 850, 1699 against the oracle; aligned decomposer too."""
    m = pkg()
    p = oracle.Params(1, 9, 6, oracle.Decomposer(8, 2), log_p=2)
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 1700, cfg_index=181)
    lwe = lwe.copy()
    lwe[0, 0] = 0
    lwe[1, p.n] = 0xFFFFFFFF
    for aligned in (False, True):
        with m.Context(to_pkg_params(p)) as ctx:
            ctx.set_decomposer_alignment(aligned)
            ctx.load_bootstrapping_key(bsk, ksk)
            big, small, tiny = ctx.blind_rotate_plan(1700), ctx.blind_rotate_plan(400), ctx.blind_rotate_plan(10)
            assert big["kernel"].startswith("pair") and big["waves_per_team"] == 1, big
            assert small["kernel"] == "team" and small["waves_per_team"] == 2, small
            assert tiny["kernel"].startswith("wide"), tiny
            got = ctx.bootstrap(lwe, tv)
            acc = ctx.blind_rotate(lwe, tv)
            ctx.set_kernel_shape(m.SHAPE_TEAM)     # slices of 100: below the wide team's limit, so force the team
            assert ctx.blind_rotate_plan(100)["kernel"] == "team"
            by_team = np.concatenate([ctx.bootstrap(lwe[i:i + 100], tv) for i in range(0, 1700, 100)])
            ctx.set_kernel_shape(m.SHAPE_AUTO)
            by_wide = np.concatenate([ctx.bootstrap(lwe[i:i + 10], tv) for i in range(0, 200, 10)])
        assert np.array_equal(got, by_team)
        assert np.array_equal(got[:200], by_wide)
        with oracle.decomposer_aligned(aligned):
            for b in (0, 1, 850, 1699):
                want, tr = oracle.bootstrap(p, lwe[b], bsk, ksk, tv, trace=True)
                assert np.array_equal(got[b], want), (aligned, b)
                assert np.array_equal(acc[b], tr["acc_final"]), (aligned, b)

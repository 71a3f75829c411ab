"""GPU (-m gpu): the multi-GPU pool of the C ABI (tfhe_pool_*, include/tfhe_hip.h) on the one-GPU box.

A pool shards a batch of independent bootstrap() calls (bootstrapping.rs:58-65) over one context per listed device
with ONE BootstrappingKey (bootstrapping.rs:18-21) prepared once and replicated device to device.  A device may be
listed several times, so pools [0], [0, 0] and [0, 0, 0] exercise everything but the xGMI hop on a single GPU: member
creation, the prepared-key replication (a device-to-device copy instead of a peer copy), the slice rule, one host thread
per member, per-member streams, the device-shard entry point.  Outputs must equal the single-context run, the committed
golden fixtures and the oracle."""
import numpy as np
import pytest

import golden_common as gc
from gpu_common import pkg, rand_u32, to_pkg_params
from test_gpu_golden import pkg_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [[0], [0, 0], [0, 0, 0]])
@pytest.mark.parametrize("name", ["ref_test", "n1024_full_word"])
def test_pool_reproduces_the_golden_outputs(name, devices):
    """8 golden rows through a pool: slices of 8 / 4+4 / 3+3+2 rows, every output word as committed"""
    m = pkg()
    pd, a = gc.load_set(name)
    with m.Pool(pkg_params(pd), devices) as pool:
        assert len(pool) == len(devices)
        pool.load_bootstrapping_key(a["bsk"], a["ksk"])
        assert np.array_equal(pool.bootstrap(a["lwe_in"], a["tv"]), a["lwe_out"])
        # fewer rows than members: the tail members get nothing
        assert np.array_equal(pool.bootstrap(a["lwe_in"][:2], a["tv"]), a["lwe_out"][:2])


def test_pool_slice_rule_matches_the_torch_distributed_path():
    m = pkg()
    import importlib
    sharding = importlib.import_module("tfhe_research_amd.sharding")
    pd, _ = gc.load_set("ref_test")
    with m.Pool(pkg_params(pd), [0, 0, 0]) as pool:
        for batch in (0, 1, 2, 3, 7, 8, 100, 1 << 20):
            for i in range(3):
                first, count = pool.shard(batch, i)
                lo, hi = sharding.shard_range(batch, 3, i)
                assert (first, first + count) == (lo, hi), (batch, i)


@pytest.mark.parametrize("backend", ["auto", "fp64"])
def test_pool_equals_single_context_on_a_ragged_full_size_batch(oracle, backend):
    """cfg2 parameters, 1,001 random ciphertexts with per-sample test vectors over three members (334 + 334 + 333):
    every word equals the single-context run; rows 0, 334, 1000 against the oracle; NAND gates through the pool too"""
    m = pkg()
    p = oracle.CFG2
    bid = {"auto": m.BACKEND_AUTO, "fp64": m.BACKEND_FP64}[backend]
    _, bsk, ksk, _ = oracle.synthetic_inputs(p, 1, cfg_index=2)
    rng = np.random.default_rng(77)
    batch = 1001
    lwe = rand_u32(rng, (batch, p.n + 1))
    tvs = rng.integers(0, 1 << p.log_p, size=(batch, p.N)).astype(np.uint32)
    ct1 = rand_u32(rng, (batch, p.n + 1))
    with m.Context(to_pkg_params(p), backend=bid) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        want = ctx.bootstrap(lwe, tvs)
        want_gate = ctx.gate(m.GATE_NAND, lwe, ct1)
    with m.Pool(to_pkg_params(p), [0, 0, 0], backend=bid) as pool:
        pool.load_bootstrapping_key(bsk, ksk)
        got = pool.bootstrap(lwe, tvs)
        got_gate = pool.gate(m.GATE_NAND, lwe, ct1)
        assert pool.backend == ("fp64-fft" if backend == "auto" else "fp64-p42")
    assert np.array_equal(got, want)
    assert np.array_equal(got_gate, want_gate)
    for b in (0, 334, 1000):
        assert np.array_equal(got[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tvs[b])), b


def test_pool_members_with_two_stream_plans_share_a_device(oracle):
    """Two members on one device, each with a shard larger than the chip over a key of two slices: every member's blind
    rotation forks onto its own second stream (kernels.hip::blind_rotate_plan) while the other member's two streams run
    beside it -- four streams on the card, one host thread per member.  Every word equals the single-context run; rows of
    both shards (and both halves of each) against the oracle."""
    m = pkg()
    p = oracle.Params(1, 10, 16, oracle.Decomposer(7, 3))
    batch = 4608  # 2,304 per member > the 1,024 samples the chip rotates at once
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, batch, cfg_index=21)
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.load_bootstrapping_key(bsk, ksk)
        plan = ctx.blind_rotate_plan(batch // 2)
        assert plan["streams"] == 2 and plan["segments"] == 2, plan
        want = ctx.bootstrap(lwe, tv)
    with m.Pool(to_pkg_params(p), [0, 0]) as pool:
        pool.load_bootstrapping_key(bsk, ksk)
        got = pool.bootstrap(lwe, tv)
        again = pool.bootstrap(lwe, tv)
    assert np.array_equal(got, want) and np.array_equal(again, want)
    for b in (0, 1151, 1153, 2303, 2304, 3457, 4607):
        assert np.array_equal(got[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


def test_pool_device_shards_and_device_key(oracle):
    """keys handed over as device tensors (member 0's device), shards resident on the members' devices, enqueue-only
    entry point + tfhe_pool_synchronize; aligned decomposer set through the pool so that the rotation depends on the key"""
    import torch
    m = pkg()
    p = oracle.Params(1, 10, 24, oracle.Decomposer(7, 3))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 96, cfg_index=41)
    dev = torch.device("cuda", 0)
    to_d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with m.Pool(to_pkg_params(p), [0, 0]) as pool:
        pool.set_decomposer_alignment(True)
        pool.load_bootstrapping_key(to_d(bsk), to_d(ksk))
        pool.reserve(96)
        shards = [to_d(lwe[:48]), to_d(lwe[48:])]
        tvs = [to_d(tv), to_d(tv)]
        outs = [torch.empty_like(s) for s in shards]
        torch.cuda.synchronize()
        pool.bootstrap_shards(shards, tvs, outs)
        pool.bootstrap_shards([shards[0], None], tvs, [outs[0], None])   # a member without work is skipped
        pool.synchronize()
        got = np.concatenate([o.cpu().numpy().view(np.uint32) for o in outs])
    with oracle.decomposer_aligned(True):
        for b in (0, 47, 48, 95):
            assert np.array_equal(got[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.set_decomposer_alignment(True)
        ctx.load_bootstrapping_key(bsk, ksk)
        assert np.array_equal(got, ctx.bootstrap(lwe, tv))


def test_pool_error_behaviour(oracle):
    m = pkg()
    p = oracle.Params(1, 10, 4, oracle.Decomposer(7, 3))
    with pytest.raises(m.TfheError) as e:
        m.Pool(to_pkg_params(p), [0, 99])            # no such device
    assert e.value.status == m.TFHE_ERR_NO_DEVICE
    with m.Pool(to_pkg_params(p), [0, 0]) as pool:
        lwe = np.zeros((4, p.n + 1), dtype=np.uint32)
        tv = np.zeros(p.N, dtype=np.uint32)
        with pytest.raises(m.TfheError) as e:
            pool.bootstrap(lwe, tv)                  # no key yet
        assert e.value.status == m.TFHE_ERR_NO_KEY and "member 0" in str(e.value)

"""GPU (-m gpu): the multi-GPU pool of the C ABI (tfhe_pool_*, include/tfhe_hip.h) on the one-GPU box.

A pool shards a batch of independent bootstrap() calls (bootstrapping.rs:58-65) over one context per listed device
with ONE BootstrappingKey (bootstrapping.rs:18-21) prepared once and replicated device to device.  A device may be
listed several times, so pools [0], [0, 0] and [0, 0, 0] exercise everything but the xGMI hop on a single GPU: member
creation, the prepared-key replication (a device-to-device copy instead of a peer copy), the slice rule, one host thread
per member, per-member streams, the device-shard entry point.  Outputs must equal the single-context run, the committed
golden fixtures and the oracle."""
import os

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


def test_pool_keygen_installs_the_new_key_on_every_member(oracle):
    """a key GENERATED through the pool replaces a key LOADED before it on every member -- not on member 0
    alone (the other members would then bootstrap their slices under the old key, silently).  Pool [0, 0]: load key A,
    generate key B (tfhe_pool_bootstrapping_key_gen, load = 1), bootstrap: every slice equals a single context under B,
    and differs from the run under A; replicate_key() after a member-0-only install repairs the same situation."""
    m = pkg()
    p = oracle.Params(1, 10, 8, oracle.Decomposer(8, 4))   # log2 B divides 32: the rotation depends on the key
    pp = to_pkg_params(p)
    lwe, bsk_a, ksk_a, tv = oracle.synthetic_inputs(p, 10, cfg_index=51)
    rng = np.random.default_rng(5)
    with m.Pool(pp, [0, 0]) as pool:
        pool.load_bootstrapping_key(bsk_a, ksk_a)
        under_a = pool.bootstrap(lwe, tv)
        lwe_sk, glwe_sk, bsk_b, ksk_b = pool.generate_keys(rng=rng, load=True)
        under_b = pool.bootstrap(lwe, tv)
        # member-0-only install of A again, then the public re-replication
        pool.member(0).load_bootstrapping_key(bsk_a, ksk_a)
        pool.replicate_key()
        again_a = pool.bootstrap(lwe, tv)
    with m.Context(pp) as ctx:
        ctx.load_bootstrapping_key(bsk_b, ksk_b)
        want_b = ctx.bootstrap(lwe, tv)
    assert np.array_equal(under_b, want_b)            # BOTH slices (rows 0-4 and 5-9) under the generated key
    assert not np.array_equal(under_b[5:], under_a[5:])
    assert np.array_equal(again_a, under_a)
    for b in (0, 4, 5, 9):
        assert np.array_equal(under_b[b], oracle.bootstrap(p, lwe[b], bsk_b, ksk_b, tv)), b


def test_pool_shards_are_all_or_nothing_on_bad_arguments(oracle):
    """tfhe_pool_bootstrap_shards_device validates every member before it enqueues anything"""
    import torch
    m = pkg()
    p = oracle.Params(1, 10, 4, oracle.Decomposer(7, 3))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 8, cfg_index=61)
    dev = torch.device("cuda", 0)
    to_d = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.int32)).to(dev)
    with m.Pool(to_pkg_params(p), [0, 0]) as pool:
        pool.load_bootstrapping_key(bsk, ksk)
        pool.reserve(8)
        outs = [torch.full((4, p.n + 1), 7, dtype=torch.int32, device=dev) for _ in range(2)]
        bad_tv = to_d(np.zeros((3, p.N), dtype=np.uint32))   # 3 test vectors for 4 rows: invalid for member 1
        with pytest.raises(m.TfheError) as e:
            pool.bootstrap_shards([to_d(lwe[:4]), to_d(lwe[4:])], [to_d(tv), bad_tv], outs)
        assert e.value.status == m.TFHE_ERR_INVALID_ARGUMENT and "member 1" in str(e.value)
        pool.synchronize()
        assert int((outs[0] != 7).sum()) == 0     # member 0 launched nothing either


# ---- more than one GPU: skipped on the one-GPU development box, run wherever a node is visible --------------------------
def _device_count() -> int:
    import torch
    return torch.cuda.device_count()


def _multi_device_lists():
    n = _device_count()
    lists = [[0, 1]] if n >= 2 else []
    if n > 2:
        lists.append(list(range(n)))
    return lists


@pytest.mark.parametrize("name", ["ref_test", "n1024_full_word"])
def test_pool_over_distinct_devices_reproduces_the_golden_outputs(name, oracle):
    """The peer-copy branch of the key replication (capi.cpp adopt_prepared_key: hipDeviceCanAccessPeer /
    hipDeviceEnablePeerAccess / hipMemcpyPeerAsync) and members on DIFFERENT devices: pools [0, 1] and all visible
    devices ≡ the golden files ≡ a single context on device 0 ≡ a single context on the last device."""
    lists = _multi_device_lists()
    if not lists:
        pytest.skip(f"needs >= 2 GPUs (hipGetDeviceCount = {_device_count()})")
    m = pkg()
    pd, a = gc.load_set(name)
    for devices in lists:
        with m.Pool(pkg_params(pd), devices) as pool:
            pool.load_bootstrapping_key(a["bsk"], a["ksk"])
            assert np.array_equal(pool.bootstrap(a["lwe_in"], a["tv"]), a["lwe_out"]), devices
    with m.Context(pkg_params(pd), device=lists[-1][-1]) as ctx:   # a context that is not on device 0
        ctx.load_bootstrapping_key(a["bsk"], a["ksk"])
        assert np.array_equal(ctx.bootstrap(a["lwe_in"], a["tv"]), a["lwe_out"])


def test_pool_over_distinct_devices_full_size(oracle):
    """cfg2 parameters, a ragged 1,001-row batch with the aligned decomposer (key-dependent) over every visible device,
    device-resident shards included: every word equals the single-context run, edge rows of every slice the oracle's"""
    lists = _multi_device_lists()
    if not lists:
        pytest.skip(f"needs >= 2 GPUs (hipGetDeviceCount = {_device_count()})")
    import torch
    m = pkg()
    p = oracle.CFG2
    _, bsk, ksk, tv = oracle.synthetic_inputs(p, 1, cfg_index=2)
    rng = np.random.default_rng(78)
    batch = 1001
    lwe = rand_u32(rng, (batch, p.n + 1))
    with m.Context(to_pkg_params(p)) as ctx:
        ctx.set_decomposer_alignment(True)
        ctx.load_bootstrapping_key(bsk, ksk)
        want = ctx.bootstrap(lwe, tv)
    devices = lists[-1]
    with m.Pool(to_pkg_params(p), devices) as pool:
        pool.set_decomposer_alignment(True)
        pool.load_bootstrapping_key(bsk, ksk)
        got = pool.bootstrap(lwe, tv)
        assert np.array_equal(got, want)
        # the enqueue-only entry point with shards resident on the members' own devices
        pool.reserve(batch)
        shards, tvs, outs = [], [], []
        for i, d in enumerate(devices):
            first, count = pool.shard(batch, i)
            dev = torch.device("cuda", d)
            shards.append(torch.from_numpy(lwe[first:first + count].view(np.int32)).to(dev))
            tvs.append(torch.from_numpy(tv.view(np.int32)).to(dev))
            outs.append(torch.empty_like(shards[-1]))
        pool.bootstrap_shards(shards, tvs, outs)
        pool.synchronize()
        assert np.array_equal(np.concatenate([o.cpu().numpy().view(np.uint32) for o in outs]), want)
        edges = sorted({pool.shard(batch, i)[0] for i in range(len(devices))} | {batch - 1})
    with oracle.decomposer_aligned(True):
        for b in edges[:6]:
            assert np.array_equal(got[b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b


def test_sharded_bootstrap_over_rccl_two_ranks(tmp_path):
    """One process per GPU: `torch.distributed.run` with 2 ranks and backend nccl (= RCCL) runs
    sharding.replicate_keys + sharding.bootstrap_sharded (scatter over RCCL, local bootstrap, gather) and rank 0
    compares the gathered rows with a single-context run and the golden fixture."""
    if _device_count() < 2:
        pytest.skip(f"needs >= 2 GPUs (hipGetDeviceCount = {_device_count()})")
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rccl_two_rank_probe.py")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", script],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "rccl two-rank OK" in out.stdout


def test_pool_kernel_shape_reaches_every_member(oracle):
    """tfhe_pool_set_kernel_shape: every member's blind rotations take the forced shape; same words"""
    m = pkg()
    p = oracle.Params(1, 10, 6, oracle.Decomposer(8, 4))
    lwe, bsk, ksk, tv = oracle.synthetic_inputs(p, 12, cfg_index=71)
    outs = {}
    for shape in (m.SHAPE_AUTO, m.SHAPE_WIDE, m.SHAPE_TEAM):
        with m.Pool(to_pkg_params(p), [0, 0]) as pool:
            pool.set_kernel_shape(shape)
            pool.load_bootstrapping_key(bsk, ksk)
            for i in range(2):
                plan = pool.member(i).blind_rotate_plan(6)
                assert plan["kernel"].startswith("team" if shape == m.SHAPE_TEAM else "wide"), (shape, i, plan)
            outs[shape] = pool.bootstrap(lwe, tv)
    assert np.array_equal(outs[m.SHAPE_WIDE], outs[m.SHAPE_TEAM]) and np.array_equal(outs[m.SHAPE_AUTO], outs[m.SHAPE_TEAM])
    for b in (0, 5, 6, 11):
        assert np.array_equal(outs[m.SHAPE_WIDE][b], oracle.bootstrap(p, lwe[b], bsk, ksk, tv)), b

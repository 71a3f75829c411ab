"""GPU box: randomised differential run of tfhe_bootstrap_batch against the CPU oracle.
usage: fuzz_gpu.py [seconds, default 300] [seed, default 1]
Every round draws a parameter set (N in {512, 1024, 2048}, k in {1, 2}, any decomposer with levels x log2 B <= 32, small n so that
the oracle finishes), a backend (mostly AUTO), a kernel shape (auto / wide / team), the decomposer alignment, a batch size (1 ...
a few thousand: the wide team, the team, the pair kernel, segment launches on two streams), the order (blind rotation first / key switch
first), a single context or a pool over the device listed two or three times, and one test vector or one per row;
uniform u32 words for ciphertexts and keys (the arithmetic is total), messages below 2^log_p in the test vectors.  A handful of rows (first, last, random) are re-computed by the oracle
(schoolbook product) on the host's threads and compared word for word.  Parameter sets a backend cannot lift exactly are counted
as refused (TFHE_ERR_EXACTNESS / _UNSUPPORTED).  Exit code 1 on any mismatch; every round is one line of the log.
Test infrastructure: the oracle is the checker here, never the thing shipped."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
m = g.load_package()
from oracle import oracle as orc
orc.build()
orc.set_poly_mul_mode(1)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
u32 = lambda *shape: rng.integers(0, 1 << 32, size=shape, dtype=np.uint64).astype(np.uint32)
threads = max(1, min(16, len(os.sched_getaffinity(0))))
pool = ThreadPoolExecutor(threads)
BACKENDS = [("auto", m.BACKEND_AUTO)] * 6 + [("fp64-fft", m.BACKEND_FP64_FFT), ("fp64-p49", m.BACKEND_FP64_P49), ("fp64", m.BACKEND_FP64),
                                             ("goldilocks", m.BACKEND_GOLDILOCKS), ("goldilocks-split", m.BACKEND_GOLDILOCKS_SPLIT)]
SHAPES = [("auto", m.SHAPE_AUTO), ("wide", m.SHAPE_WIDE), ("team", m.SHAPE_TEAM)]
BATCHES = [1, 1, 2, 3, 5, 17, 64, 255, 257, 300, 1025, 1537, 2100, 4099]

t0 = time.time()
rounds = refused = bad = 0
while time.time() - t0 < budget:
    logn = int(rng.choice([9, 9, 10, 10, 11])); k = int(rng.choice([1, 1, 2]))
    levels = int(rng.integers(1, 9)); log_b = int(rng.integers(1, 32 // levels + 1))
    ks_levels = int(rng.integers(1, 9)); ks_log_b = int(rng.integers(1, 32 // ks_levels + 1))
    log_p = int(rng.integers(1, 5))
    # the oracle's cost per row ~ n (k+1)^2 l N^2: keep a row under about a second
    cost = (k + 1) ** 2 * levels * (1 << (2 * logn))
    n = int(max(1, min(int(rng.integers(1, 14)), 2.5e9 // cost)))
    if rng.integers(5) == 0:  # now and then a longer key: segment launches on two streams need n >= 8 and a batch above the resident count
        n = int(max(1, min(int(rng.integers(16, 121)), 2.5e9 // cost)))
    ks_first = bool(rng.integers(4) == 0)   # notes/TFHE.md:367-400: key switch before the blind rotation, [kN + 1] in and out
    members = int(rng.choice([1, 1, 1, 2, 3]))  # 2, 3: a pool over the same device listed twice / three times (csrc/pool.cpp)
    bname, backend = BACKENDS[int(rng.integers(len(BACKENDS)))]
    sname, shape = SHAPES[int(rng.integers(len(SHAPES)))]
    aligned = bool(rng.integers(2))
    batch = int(BATCHES[int(rng.integers(len(BATCHES)))])
    if logn == 11:
        batch = min(batch, 1025)
    per_row_tv = bool(rng.integers(3) == 0)
    P = orc.Params(k, logn, n, orc.Decomposer(log_b, levels), orc.Decomposer(ks_log_b, ks_levels), log_p=log_p)
    pp = m.TfheParams(k, logn, n, m.DecomposerParams(log_b, levels), m.DecomposerParams(ks_log_b, ks_levels), log_p=log_p)
    tag = (f"N={1 << logn} k={k} n={n} pbs=({log_b},{levels}) ks=({ks_log_b},{ks_levels}) log_p={log_p} backend={bname} shape={sname} "
           f"aligned={int(aligned)} batch={batch} tv={'per-row' if per_row_tv else 'one'} order={'ks-first' if ks_first else 'pbs-first'} members={members}")
    rounds += 1
    try:
        ctx = m.Context(pp, backend=backend) if members == 1 else m.Pool(pp, [0] * members, backend=backend)
    except m.TfheError as e:
        refused += 1
        print(f"refused  {tag}: {str(e)[:80]}", flush=True)
        continue
    try:
        lwe, bsk, ksk = u32(batch, (P.big_n if ks_first else n) + 1), u32(*P.bsk_shape()), u32(*P.ksk_shape())
        tv = rng.integers(0, 1 << log_p, size=(batch, P.N) if per_row_tv else (P.N,)).astype(np.uint32)  # messages (glwe.rs:144)
        ctx.set_decomposer_alignment(aligned)
        ctx.set_kernel_shape(shape)
        ctx.set_bootstrap_order(ks_first)
        ctx.load_bootstrapping_key(bsk, ksk)
        got = ctx.bootstrap(lwe, tv)
        first = ctx if members == 1 else ctx.member(0)
        kernel = first.blind_rotate_plan(batch if members == 1 else ctx.shard(batch, 0)[1])["kernel"]
        rows = sorted({0, batch - 1, int(rng.integers(batch)), int(rng.integers(batch))})
        with orc.decomposer_aligned(aligned):
            fn = orc.bootstrap_ks_first if ks_first else orc.bootstrap
            want = list(pool.map(lambda r: fn(P, lwe[r], bsk, ksk, tv[r] if per_row_tv else tv), rows))
        wrong = [r for r, w in zip(rows, want) if not np.array_equal(w, got[r])]
        if wrong:
            bad += 1
            print(f"MISMATCH {tag} kernel={kernel} rows {wrong}", flush=True)
        else:
            print(f"ok       {tag} kernel={kernel.split(' (')[0]} ({first.backend}) rows={len(rows)}", flush=True)
    finally:
        ctx.close()
print(f"# {rounds} rounds in {time.time() - t0:.0f} s (seed {seed}): {rounds - refused - bad} bit-exact, {refused} refused by the exactness bounds, {bad} MISMATCHES", flush=True)
sys.exit(1 if bad else 0)

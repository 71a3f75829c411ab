#!/usr/bin/env python3
"""bench.py -- programmable bootstraps per second on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (bootstrap(): blind rotation + sample extract + key switch,
reference bootstrapping.rs:58-120) over one batch of synthetic LWE ciphertexts that is already
resident in HBM.  Default workload = BASELINE.json configs[1]: batch 4096, N=1024, k=1, n=630,
l=3, log2B=7 (KS log2B=4, l=5; log_p=2, padding 1 as fixed in SURVEY 8d), identity LUT.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Multi-GPU: independent LWE bootstraps shard across ranks with no data-path collective (keys are
replicated per GPU); weak scaling: every rank processes `--batch` ciphertexts per step.  The only
collectives are the timing barrier / max-reduction (RCCL).

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     : the blind-rotation kernel (the dominant one) against the HBM roofline, priced in
                 ALGORITHMIC bytes = external products x 4*N*(k+1)*((k+1)*l+2) bytes,
                 duration measured with HIP events on the kernel's own stream during the timed
                 region;
  cpu_baseline : the literal CPU restatement of the reference (oracle, kind "port": the Rust
                 reference cannot be built in this image) timed single-threaded on a bounded
                 sample of the same workload, on this box's host cores (N=1, rank 0 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)
HBM_MEASURED_GBS = 6290.0  # same guide: 6.29 TB/s measured with a float4 copy kernel


def measure_hbm_copy_gbs(torch, dev, mib: int = 1024, reps: int = 10, ctx=None) -> float:
    """HBM roofline measured in this run: a 16-byte-per-lane stream copy of `mib` MiB by the
    library's own probe kernel (read + write bytes over HIP-event time); torch's device-to-device
    copy if the context is gone; the guide's figure if both fail."""
    if ctx is not None:
        try:
            return ctx.measure_hbm_copy(mib, reps)
        except Exception:  # noqa: BLE001 - a failed probe must not lose the benchmark line
            pass
    try:
        src = torch.empty(mib << 20, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        src.fill_(1)
        for _ in range(2):
            dst.copy_(src)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        for _ in range(reps):
            dst.copy_(src)
        stop.record()
        stop.synchronize()
        ms = start.elapsed_time(stop) / reps
        del src, dst
        return 2.0 * (mib << 20) / (ms * 1e-3) / 1e9
    except Exception:  # noqa: BLE001 - a failed probe must not lose the benchmark line
        return HBM_MEASURED_GBS


def kernel_source_hash() -> str:
    """sha256 over the CODE of the sources the DEVICE code is built from (csrc/kernels.hip and the csrc headers it
    includes, the build recipe), first 16 hex digits: identifies the kernels a PMC record was measured on.  Host-only
    sources (capi.cpp, pool.cpp, context.h, the C ABI header) do not count -- a host edit leaves a kernel measurement
    valid -- and neither do comments and white space (a reworded comment must not invalidate a measurement; any token
    that reaches the device compiler does).  Only regular files of the explicit list are read: an editor backup or a
    directory under csrc/ changes nothing.  (Not the .so itself: hipcc output embeds build paths, and the box that
    collects counters builds nothing -- it runs the library this tree built.)"""
    import hashlib
    import re
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "tfhe-research_amd", "csrc")
    host_only = {"context.h"}
    names = ["kernels.hip"] + sorted(f for f in os.listdir(csrc) if f.endswith(".h") and f not in host_only)
    files = [os.path.join(csrc, f) for f in names if os.path.isfile(os.path.join(csrc, f))]
    files.append(os.path.join(ROOT, "tfhe-research_amd", "build.py"))
    for f in files:
        with open(f, "r", errors="replace") as fh:
            text = fh.read()
        if f.endswith(".py"):
            text = re.sub(r"#[^\n]*", "", text)
        else:
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)   # block comments
            # line comments; a "//" inside a string literal stays (the literal is matched first and kept)
            text = re.sub(r'"(?:\\.|[^"\\\n])*"|//[^\n]*', lambda m: m.group(0) if m.group(0).startswith('"') else "", text)
        text = re.sub(r"\s+", " ", text).strip()
        h.update(os.path.basename(f).encode() + b"\0" + text.encode() + b"\0")
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str, workload: str, batch: int):
    """HBM/fabric bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes
    (FETCH_SIZE and WRITE_SIZE are collected in separate runs; FETCH_SIZE doubled per the gfx950
    correction).  Counters cannot be collected inside this process, so the number is only reported
    when the committed measurement is for this very kernel and workload AND was taken on the kernels this
    tree builds (`kernel_source_hash`): a record of an older kernel is refused, not silently reused."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None, None, None
    if rec.get("kernel") != kernel or rec.get("workload") != f"{workload} batch {batch}":
        return None, None, None
    if rec.get("kernel_source_hash") != kernel_source_hash():
        return None, (f"profiles/pmc_traffic.json is stale: measured on kernel sources {rec.get('kernel_source_hash')}, "
                      f"this tree is {kernel_source_hash()} (redo tools/final_profile.sh)"), None
    return rec["traffic_bytes_per_launch"], rec.get("source"), rec.get("valu") or None

WORKLOADS = {
    # name: (k, logN, n, (pbs logB, l), (ks logB, l), log_p, default batch)
    "cfg2": (1, 10, 630, (7, 3), (4, 5), 2, 4096),
    "cfg3": (2, 9, 722, (4, 6), (4, 5), 2, 4096),
    "cfg1": (1, 9, 500, (8, 2), (4, 5), 2, 4096),
    "cfg5": (2, 11, 630, (8, 4), (4, 5), 4, 4096),
    # BASELINE configs[3]: batch 2^20 sharded over 8 GPUs = 2^17 ciphertexts per GPU, cfg2 parameters
    "cfg4": (1, 10, 630, (7, 3), (4, 5), 2, 1 << 17),
}
CPU_BASELINE_PBS = 3        # >= 3 PBS on one thread (the reference is single-threaded)
CPU_ALIGNED_CHECK_ROWS = 16  # rows of the aligned-decomposer leg the oracle re-computes (the leg whose digits are non-zero)


def usable_cores() -> dict:
    """Host cores this process may really use: the scheduler affinity, capped by the cgroup CPU quota when the box
    sets one (a GPU box hands a one-GPU job a share of its host, whatever os.cpu_count() says)."""
    logical = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = logical
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:          # cgroup v2: "<quota|max> <period>"
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    usable = affinity if quota is None else max(1, min(affinity, int(quota + 0.5)))
    return {"logical": logical, "affinity": affinity, "cgroup_quota": quota, "usable": usable}


def spaced_rows(batch: int, count: int):
    """`count` distinct row indices of a batch, first and last included, evenly spaced."""
    count = max(1, min(count, batch))
    if count == 1:
        return [0]
    return sorted({round(i * (batch - 1) / (count - 1)) for i in range(count)})


def cpu_legs(workload: str, budget_s: float, host: dict):
    """The CPU side of a run, on rows OF THE TIMED BATCH (host: numpy copies of `lwe` rows, bsk, ksk, tv and of the rows
    the GPU wrote for them), so that the baseline also checks the bits that were timed:

      1. `single`: CPU_BASELINE_PBS whole bootstraps, literal Toeplitz path (utils.rs:113-160, as the reference does),
         one thread (the reference is single-threaded) -> cpu_baseline.value; every row compared with the GPU's.
      2. `cores_all`: one bootstrap per usable host core ("an all-cores run, nproc stated"), same path, one
         round of independent ciphertexts of the timed batch (ctypes releases the GIL) -> cpu_baseline.cores_all; every
         row compared.
      3. the aligned-decomposer leg's rows (the cfg2 timing in which digits are non-zero and the result depends on the
         key), up to CPU_ALIGNED_CHECK_ROWS, re-computed with the oracle's aligned decomposer (schoolbook product: this
         leg is a check, not a timing) and compared.

    -> (cpu_baseline dict, verified dict).  Kept short on purpose (about 25 s of wall clock at cfg2)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as orc
    orc.build()
    k, logn, n, pbs, ks, log_p, _ = WORKLOADS[workload]
    p = orc.Params(k, logn, n, orc.Decomposer(*pbs), orc.Decomposer(*ks), log_p=log_p)
    bsk, ksk, tv = host["bsk"], host["ksk"], host["tv"]
    idx, lwe, gpu = host["rows"], host["lwe_rows"], host["gpu_rows"]
    pos = {r: i for i, r in enumerate(idx)}
    cores = usable_cores()
    mismatches = []

    def check(row, want, leg):
        if not np.array_equal(want, gpu[pos[row]]):
            mismatches.append({"row": int(row), "leg": leg, "words_differing": int(np.count_nonzero(want != gpu[pos[row]]))})

    orc.set_poly_mul_mode(0)
    single_rows = host["single_rows"]
    done, t0 = 0, time.perf_counter()
    for r in single_rows:
        check(r, orc.bootstrap(p, lwe[pos[r]], bsk, ksk, tv), "single")
        done += 1
        if time.perf_counter() - t0 >= budget_s:
            break
    dt = time.perf_counter() - t0
    baseline = {
        "value": done / dt, "unit": "PBS/s", "cores": 1, "kind": "port",
        "sample": f"rows {single_rows[:done]} of the timed batch: {done} full bootstraps of {workload} (literal Toeplitz path, "
                  f"gcc -O2, 1 thread) in {dt:.1f} s",
        "host_cores_available": cores["logical"], "host_cores": cores,
    }
    checked = done
    threads = min(cores["usable"], len(idx))
    if threads > 1 and budget_s > 0:
        rows_all = [idx[i] for i in spaced_rows(len(idx), threads)]
        t1 = time.perf_counter()
        with ThreadPoolExecutor(threads) as pool:
            wants = list(pool.map(lambda r: orc.bootstrap(p, lwe[pos[r]], bsk, ksk, tv), rows_all))
        dt_all = time.perf_counter() - t1
        for r, w in zip(rows_all, wants):
            check(r, w, "cores_all")
        checked += len(rows_all)
        baseline["cores_all"] = {
            "value": len(rows_all) / dt_all, "unit": "PBS/s", "cores": threads,
            "sample": f"{len(rows_all)} bootstraps of rows of the timed batch, one per thread on {threads} threads "
                      f"(= the host cores this process may use: affinity {cores['affinity']}, cgroup quota {cores['cgroup_quota']}, "
                      f"{cores['logical']} logical) in {dt_all:.1f} s"}
    verified = {"rows": checked, "bit_exact": not mismatches,
                "what": f"oracle (literal Toeplitz path) on rows of the timed batch, every word of {checked} output rows "
                        "against what the timed steps left in HBM"}
    aligned = host.get("aligned_gpu_rows")
    if aligned is not None:
        orc.set_poly_mul_mode(1)
        rows_al = [idx[i] for i in spaced_rows(len(idx), min(CPU_ALIGNED_CHECK_ROWS, max(threads, 3)))]
        bad = []
        with orc.decomposer_aligned(True):
            with ThreadPoolExecutor(max(1, min(threads, len(rows_al)))) as pool:
                wants = list(pool.map(lambda r: orc.bootstrap(p, lwe[pos[r]], bsk, ksk, tv), rows_al))
        for r, w in zip(rows_al, wants):
            if not np.array_equal(w, aligned[pos[r]]):
                bad.append(int(r))
        verified["aligned_decomposer"] = {"rows": len(rows_al), "bit_exact": not bad, "rows_differing": bad}
        if bad:
            mismatches.append({"leg": "aligned_decomposer", "rows": bad})
    if mismatches:
        verified["mismatches"] = mismatches[:8]
        verified["bit_exact"] = False
    orc.set_poly_mul_mode(1)
    return baseline, verified


def bench_external_product(args, pkg, params, batch, dev, rand_words, world, rank, local_rank, backend):
    """Standalone external product (ggsw.rs:132-161): out[b] = GGSW (x) GLWE[b] with the GGSW already in
    the NTT domain.  Shared GGSW = the blind-rotation shape; --ggsw-per-sample streams one prepared
    GGSW per sample from HBM.  Algorithmic bytes per product: 4*N*(k+1)*((k+1)*l + 2)."""
    import torch
    import torch.distributed as dist
    ctx = pkg.Context(params, device=local_rank, backend=backend)
    ctx.use_torch_stream()
    ctx.set_timing(True)
    count = batch if args.ggsw_per_sample else 1
    ggsw = rand_words(count, params.R, params.k + 1, params.N)
    prepared = ctx.prepare_ggsw_device(ggsw)
    glwe = rand_words(batch, params.k + 1, params.N)
    out = torch.empty_like(glwe)
    use_dist = dist.is_initialized()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # One launch is ~0.1 ms: a handful of launches runs on a chip that has not reached its sustained clocks
    # (20 launches after 3 of warm-up measured 0.123 ms per launch, 200 after 300 measured 0.103 ms, 1000
    # after 1000 the same 0.103 ms).  So: at least 300 untimed launches, at least 200 timed ones.
    steps = max(args.steps, 200)
    warmup = max(args.warmup, 300)
    for _ in range(warmup):
        ctx.external_product_prepared(prepared, glwe, out=out)
    barrier()
    kms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        ctx.external_product_prepared(prepared, glwe, out=out)
        kms.append(ctx.last_kernel_ms()[0])
    barrier()
    dt = time.perf_counter() - t0
    # The reference's call shape hands over the GGSW as raw u32 (ggsw.rs:132-161): 4x fewer key bytes than
    # the prepared form, but every key polynomial needs F::kParts forward transforms first.  Timed here as
    # what the host entry point does on the device: prepare kernel + product kernel, per launch pair.
    raw_ms = None
    if args.ggsw_per_sample:
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        scratch = torch.empty_like(prepared)
        for _ in range(2):
            ctx.prepare_ggsw_device(ggsw, out=scratch)
            ctx.external_product_prepared(scratch, glwe, out=out)
        start.record()
        for _ in range(steps):
            ctx.prepare_ggsw_device(ggsw, out=scratch)
            ctx.external_product_prepared(scratch, glwe, out=out)
        stop.record()
        stop.synchronize()
        raw_ms = start.elapsed_time(stop) / steps
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kernel_ms = float(np.mean(kms))
    algo = batch * params.external_product_bytes()
    physical = (count * prepared.shape[1] * 8) + 2 * batch * (params.k + 1) * params.N * 4
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    # with ONE GGSW shared by the batch a product only moves its GLWE in and out -- 8 N (k+1) bytes --
    # and the GGSW once per launch; that accounting is reported beside the 4 N (k+1) ((k+1) l + 2) one, labelled
    glwe_only = 8 * params.N * (params.k + 1)
    algo_shared = batch * glwe_only + params.R * (params.k + 1) * params.N * 4
    achieved_shared = algo_shared / (kernel_ms * 1e-3) / 1e9
    hbm_copy = measure_hbm_copy_gbs(torch, dev, ctx=ctx)
    result = {
        "metric": "external_products_per_sec", "value": batch * world * steps / dt, "unit": "products/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if ctx.backend.startswith("fp64") else "u64", "data": "synthetic",
        "config": {"workload": f"external_product {args.workload}: batch {batch}/GPU, N={params.N}, k={params.k}, "
                               f"l={params.pbs_decomposer.levels}, log2B={params.pbs_decomposer.log_base}, "
                               + ("one prepared GGSW per sample" if args.ggsw_per_sample else "one GGSW shared by the batch")},
        "roofline": {"kernel": f"external_product_kernel<{ctx.backend},{params.glwe_poly_degree},{params.k}>",
                     "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "hbm_copy_measured_GBps": hbm_copy,
                     "frac_of_measured_hbm": achieved / hbm_copy,
                     "traffic": None, "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo,
                     "physical_operand_bytes_per_launch": physical,
                     "physical_GBps": physical / (kernel_ms * 1e-3) / 1e9},
    }
    if not args.ggsw_per_sample:
        result["roofline"].update({
            "accounting": "achieved/frac price every product at 4N(k+1)((k+1)l+2) bytes (GGSW + GLWE in + GLWE out, SURVEY 8d's "
                          "unit) although this launch reads ONE GGSW for the whole batch; achieved_shared_ggsw prices what "
                          "the shape really moves: 8N(k+1) bytes per product + the GGSW once",
            "achieved_shared_ggsw": achieved_shared, "frac_shared_ggsw": achieved_shared / HBM_PEAK_GBS,
            "frac_of_measured_hbm_shared_ggsw": achieved_shared / hbm_copy,
            "algorithmic_bytes_per_launch_shared_ggsw": algo_shared,
            "bound": "valu-issue (the operands are L2-resident; neither accounting comes near HBM)"})
    if raw_ms is not None:
        raw_bytes = count * params.R * (params.k + 1) * params.N * 4 + 2 * batch * (params.k + 1) * params.N * 4
        result["raw_u32_ggsw"] = {
            "ms_per_launch_pair": raw_ms, "what": "bsk_prepare_kernel (forward transforms of the raw u32 GGSWs) + external_product_kernel, "
            "the raw GGSW being the only key bytes read from HBM by the pair's first kernel",
            "raw_operand_bytes": raw_bytes, "algorithmic_GBps": algo / (raw_ms * 1e-3) / 1e9,
            "faster_than_prepared_stream": bool(raw_ms < kernel_ms)}
    ctx.close()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()


def bench_batch_sweep(args, pkg, dev, local_rank, backend):
    """The reference's own call shape (bootstrap() takes ONE ciphertext, and()/or() one pair: bootstrapping.rs:58-65,
    boolean.rs:9-37) and everything between it and the headline batch: for each workload and batch size
      latency_ms   one synchronised call (enqueue -> results complete), median of `reps` calls: what a caller that needs
                   the result before it goes on (a dependent gate) waits;
      pbs_per_s    `reps` calls enqueued back to back, one synchronisation: the rate of independent small batches;
      plan         how the blind rotation of that batch goes out (kernel shape, launches, streams).
    One JSON line: {"metric": "batch_sweep", "rows": [...]}."""
    import torch
    rows = []
    workloads = [w for w in args.sweep_workloads.split(",") if w]
    for wl in workloads:
        k, logn, n, pbs, ks, log_p, _ = WORKLOADS[wl]
        params = pkg.TfheParams(k, logn, n, pkg.DecomposerParams(*pbs), pkg.DecomposerParams(*ks), log_p=log_p)
        gen = torch.Generator(device=dev)
        gen.manual_seed(0x74666865)
        rw = lambda *shape: torch.randint(-(1 << 31), (1 << 31) - 1, shape, dtype=torch.int32, device=dev, generator=gen)
        sizes = [int(b) for b in args.sweep_batches.split(",")]
        lwe_all = rw(max(sizes), n + 1)
        tv = torch.from_numpy(pkg.construct_identity_test_vector(params).astype(np.int32)).to(dev)
        ctx = pkg.Context(params, device=local_rank, backend=backend)
        ctx.use_torch_stream()
        ctx.load_bootstrapping_key(rw(*params.bsk_shape()), rw(*params.ksk_shape()))
        if pbs[0] * (32 // pbs[0]) != 32:
            ctx.set_decomposer_alignment(True)   # cfg2: the data-dependent timing (the literal decomposer multiplies zeros)
        ctx.reserve(max(sizes))
        ctx.set_timing(True)
        shapes = {"auto": pkg.SHAPE_AUTO, "wide": pkg.SHAPE_WIDE, "team": pkg.SHAPE_TEAM}
        for shape_name, b in [(sn, b) for sn in args.sweep_shapes.split(",") for b in sizes]:
            ctx.set_kernel_shape(shapes[shape_name])
            lwe = lwe_all[:b]
            out = torch.empty_like(lwe)
            reps = max(10, min(200, 4096 // b))
            for _ in range(max(3, reps // 4)):
                ctx.bootstrap(lwe, tv, out=out)
            torch.cuda.synchronize()
            lat = []
            for _ in range(reps):
                t0 = time.perf_counter()
                ctx.bootstrap(lwe, tv, out=out)
                torch.cuda.synchronize()
                lat.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.bootstrap(lwe, tv, out=out)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            br, ksm = ctx.kernel_ms_ago(0)
            rows.append({"workload": wl, "shape": shape_name, "batch": b, "latency_ms": float(np.median(lat)) * 1e3,
                         "latency_ms_min": float(np.min(lat)) * 1e3, "pbs_per_s": b * reps / dt,
                         "blind_rotate_ms": br, "key_switch_ms": ksm, "reps": reps, "plan": ctx.blind_rotate_plan(b),
                         "backend": ctx.backend})
        ctx.close()
    print(json.dumps({"metric": "batch_sweep", "unit": "PBS/s and ms per call", "n_gpus": 1, "data": "synthetic",
                      "note": "cfg2 rows use the aligned decomposer (non-zero digits); inputs resident in HBM; latency = one "
                              "synchronised tfhe_bootstrap_batch_device call",
                      "rows": rows}), flush=True)


def bench_pool(args, pkg, params, batch, devices, backend):
    """One process, several GPUs: the multi-GPU pool of the C ABI (tfhe_pool_*).  Weak scaling like the
    torch.distributed path: every member bootstraps `batch` ciphertexts per step from a shard resident in ITS device's
    HBM; the key is drawn on member 0's device, prepared once and replicated device to device.  A device may be listed
    twice (rehearsal on a one-GPU box: the members then share the card and the rate means nothing)."""
    import torch
    k, n, logn = params.k, params.n, params.glwe_poly_degree
    members = len(devices)
    dev0 = torch.device("cuda", devices[0])
    gen = torch.Generator(device=dev0)
    gen.manual_seed(0x74666865)
    rand = lambda dev, *shape: torch.randint(-(1 << 31), (1 << 31) - 1, shape, dtype=torch.int32, device=dev,
                                             generator=gen if dev == dev0 else None)
    pool = pkg.Pool(params, devices, backend=backend)
    t_key = time.perf_counter()
    bsk, ksk = rand(dev0, *params.bsk_shape()), rand(dev0, *params.ksk_shape())
    pool.load_bootstrapping_key(bsk, ksk)
    key_s = time.perf_counter() - t_key
    del bsk, ksk
    tv_host = pkg.construct_identity_test_vector(params).astype(np.int32)
    shards, tvs, outs = [], [], []
    for d in devices:
        dev = torch.device("cuda", d)
        shards.append(rand(dev, batch, n + 1))
        tvs.append(torch.from_numpy(tv_host).to(dev))
        outs.append(torch.empty_like(shards[-1]))
    pool.reserve(batch * members)
    ctx0 = pool.member(0)
    ctx0.set_timing(True)
    for d in set(devices):
        torch.cuda.synchronize(d)
    for _ in range(args.warmup):
        pool.bootstrap_shards(shards, tvs, outs)
    pool.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pool.bootstrap_shards(shards, tvs, outs)   # enqueue only: every member has its own stream
    pool.synchronize()
    dt = time.perf_counter() - t0
    br = float(np.mean([ctx0.kernel_ms_ago(i)[0] for i in range(min(args.steps, 64))]))
    algo = batch * n * params.external_product_bytes()
    achieved = algo / (br * 1e-3) / 1e9
    gpus = len(set(devices))
    result = {
        "metric": "programmable_bootstraps_per_sec", "value": batch * members * args.steps / dt, "unit": "PBS/s",
        "n_gpus": gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64" if pool.backend.startswith("fp64") else "u64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: batch {batch}/member, N={1 << logn}, k={k}, n={n}, "
                               f"l={params.pbs_decomposer.levels}, log2B={params.pbs_decomposer.log_base}, identity LUT",
                   "global_batch": batch * members,
                   "parallelism": f"tfhe_pool over devices {devices} in ONE process (C ABI): contiguous shards resident per "
                                  f"device, no data-path collective; key prepared once on device {devices[0]} and replicated "
                                  f"device to device ({key_s:.2f} s incl. drawing it)"
                                  + ("; SEVERAL MEMBERS SHARE A GPU: rehearsal, the rate is not a scaling figure" if gpus < members else "")},
        "roofline": {"kernel": f"blind_rotate_kernel<{pool.backend},{logn},{k}>", "bound": "valu-issue", "priced_against": "hbm",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel_ms": br, "algorithmic_bytes_per_launch": algo,
                     "note": "member 0's kernel; every member runs the same launch on its own shard"},
    }
    pool.close()
    print(json.dumps(result), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="ciphertexts per GPU per step (default: workload's)")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0,
                    help="time budget of the single-thread CPU leg (it stops after CPU_BASELINE_PBS bootstraps anyway)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="auto", choices=["auto", "goldilocks", "fp64", "goldilocks-split", "fp64-p49", "fp64-fft"])
    ap.add_argument("--kernel", default="bootstrap", choices=["bootstrap", "external_product"],
                    help="external_product: time the standalone GGSW x GLWE kernel (ggsw.rs:132-161) instead of the PBS")
    ap.add_argument("--ggsw-per-sample", action="store_true",
                    help="external_product only: one prepared GGSW per sample (streams from HBM) instead of one shared by the batch")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="include the RCCL scatter of the input batch from rank 0 and the gather of the results "
                         "in every step (tfhe_research_amd.sharding); default: shards are resident per rank")
    ap.add_argument("--scatter-gather-figure", action="store_true",
                    help="N > 1: after the timed region, also time the step with rank 0 owning the whole batch and add it "
                         "to the line as `scatter_gather` (never `value`).  Off by default: the headline line must not "
                         "depend on an optional point-to-point exchange")
    ap.add_argument("--pool-devices", default="",
                    help="comma-separated HIP device ordinals: time the multi-GPU pool of the C ABI (tfhe_pool_*) in ONE "
                         "process instead of one process per GPU, e.g. 0,1,2,3,4,5,6,7 (0,0 rehearses it on one GPU)")
    ap.add_argument("--no-secondary-legs", action="store_true",
                    help="skip the figures taken after the timed region (aligned decomposer, exact prime-field backend): for "
                         "rocprofv3 --stats runs whose per-kernel averages should contain the timed launches only")
    ap.add_argument("--batch-sweep", action="store_true",
                    help="latency of one call and rate of back-to-back calls at batches 1 .. 4096 (the reference's call shape is "
                         "ONE ciphertext per bootstrap()); prints one JSON line with a row per (workload, batch)")
    ap.add_argument("--sweep-batches", default="1,8,64,256,1024,4096")
    ap.add_argument("--sweep-workloads", default="cfg2,cfg3")
    ap.add_argument("--sweep-shapes", default="auto",
                    help="kernel shapes to sweep (tfhe_context_set_kernel_shape): auto, wide, team; e.g. team,wide for the A/B")
    ap.add_argument("--gate", default="", choices=["", "nand", "and", "or", "xor"],
                    help="step = one homomorphic gate over the batch (boolean.rs: bootstrap(2*ct1 + ct0)) instead of a plain PBS")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    if args.pool_devices:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the product has no CPU path")
        pkg = entry.load_package()
        k, logn, n, pbs, ks, log_p, default_batch = WORKLOADS[args.workload]
        params = pkg.TfheParams(k, logn, n, pkg.DecomposerParams(*pbs), pkg.DecomposerParams(*ks), log_p=log_p)
        backend = {"auto": pkg.BACKEND_AUTO, "goldilocks": pkg.BACKEND_GOLDILOCKS, "fp64": pkg.BACKEND_FP64,
                   "goldilocks-split": pkg.BACKEND_GOLDILOCKS_SPLIT, "fp64-p49": pkg.BACKEND_FP64_P49,
                   "fp64-fft": pkg.BACKEND_FP64_FFT}[args.backend]
        return bench_pool(args, pkg, params, args.batch or default_batch, [int(d) for d in args.pool_devices.split(",")], backend)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if os.environ.get("TFHE_BENCH_BACKEND", "nccl") != "nccl":
        local_rank %= torch.cuda.device_count()  # rehearsal: ranks share the GPUs that exist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # one process per GPU; under torch.distributed.run the group is created even for a single rank so
    # that the RCCL path (barrier + max-reduction) is the one that runs at every N
    use_dist = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # TFHE_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path with several ranks on ONE GPU (RCCL refuses
        # two ranks per device); the measured configuration is always nccl = RCCL
        dist_backend = os.environ.get("TFHE_BENCH_BACKEND", "nccl")
        if dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    pkg = entry.load_package()
    k, logn, n, pbs, ks, log_p, default_batch = WORKLOADS[args.workload]
    batch = args.batch or default_batch
    params = pkg.TfheParams(k, logn, n, pkg.DecomposerParams(*pbs), pkg.DecomposerParams(*ks), log_p=log_p)

    # synthetic, uniformly random u32 words (the arithmetic is total); HBM resident before timing
    gen = torch.Generator(device=dev)
    gen.manual_seed(0x74666865 + rank)

    def rand_words(*shape):
        return torch.randint(-(1 << 31), (1 << 31) - 1, shape, dtype=torch.int32, device=dev, generator=gen)

    lwe = rand_words(batch, n + 1)
    tv = torch.from_numpy(pkg.construct_identity_test_vector(params).astype(np.int32)).to(dev)
    out = torch.empty_like(lwe)
    key_replication = "one GPU"
    if use_dist and world > 1 and args.kernel == "bootstrap":
        # the read-only keys exist once (on rank 0, like a BootstrappingKey the host
        # uploaded there, bootstrapping.rs:18-21) and are REPLICATED to every GPU with one RCCL
        # broadcast per tensor at key-load time; nothing else of the path ever crosses ranks
        import importlib
        sharding = importlib.import_module("tfhe_research_amd.sharding")
        t_rep = time.perf_counter()
        held = [rand_words(*params.bsk_shape()), rand_words(*params.ksk_shape())] if rank == 0 else None
        bsk, ksk = sharding.replicate_keys(held, [params.bsk_shape(), params.ksk_shape()], root=0, like=lwe)
        torch.cuda.synchronize()
        del held
        key_replication = (f"BSK+KSK drawn on rank 0 and broadcast over {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend()} to {world} ranks "
                           f"({(bsk.numel() + ksk.numel()) * 4 / 1e6:.0f} MB, {time.perf_counter() - t_rep:.2f} s incl. generation)")
    else:
        bsk = rand_words(*params.bsk_shape())
        ksk = rand_words(*params.ksk_shape())

    backend = {"auto": pkg.BACKEND_AUTO, "goldilocks": pkg.BACKEND_GOLDILOCKS, "fp64": pkg.BACKEND_FP64,
               "goldilocks-split": pkg.BACKEND_GOLDILOCKS_SPLIT, "fp64-p49": pkg.BACKEND_FP64_P49,
               "fp64-fft": pkg.BACKEND_FP64_FFT}[args.backend]
    if args.batch_sweep:
        return bench_batch_sweep(args, pkg, dev, local_rank, backend)
    if args.kernel == "external_product":
        return bench_external_product(args, pkg, params, batch, dev, rand_words, world, rank, local_rank, backend)
    ctx = pkg.Context(params, device=local_rank, backend=backend)
    backend_name = ctx.backend
    ctx.use_torch_stream()
    ctx.load_bootstrapping_key(bsk, ksk)
    # the CPU legs re-compute rows of the TIMED batch with the oracle: they need the same keys on the host
    plain_pbs = not args.gate and not args.scatter_gather
    check_bits = plain_pbs and not args.no_cpu_baseline
    host = None
    if check_bits:
        u32 = lambda t: t.cpu().numpy().view(np.uint32)
        host = {"bsk": u32(bsk), "ksk": u32(ksk), "tv": u32(tv)}
    del bsk
    ctx.reserve(batch)
    ctx.set_timing(True)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if args.gate:
        truth = {"nand": pkg.GATE_NAND, "and": pkg.GATE_AND, "or": pkg.GATE_OR, "xor": pkg.GATE_XOR}[args.gate]
        lwe2 = rand_words(batch, n + 1)

        def step():
            ctx.gate(truth, lwe, lwe2, out=out)
    elif args.scatter_gather:
        if not use_dist:
            raise SystemExit("--scatter-gather needs torch.distributed.run (RANK/WORLD_SIZE in the environment)")
        import importlib
        sharding = importlib.import_module("tfhe_research_amd.sharding")
        full = rand_words(batch * world, n + 1) if rank == 0 else None

        def step():
            # rank 0 owns the whole batch: scatter rows over RCCL, bootstrap the local shard, gather
            sharding.bootstrap_sharded(lambda shard, tvv: ctx.bootstrap(shard, tvv), full, tv, root=0,
                                       batch=batch * world, width=n + 1, like=lwe)
    else:
        def step():
            ctx.bootstrap(lwe, tv, out=out)

    for _ in range(args.warmup):
        step()
    barrier()
    br_ms, ks_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()  # enqueued back to back: the context records this step's events in its ring, nothing waits here
    barrier()
    dt = time.perf_counter() - t0
    # per-launch durations of the timed steps (HIP events on the kernel's own stream, the last 64 at most)
    for ago in range(min(args.steps, 64)):
        b, s = ctx.kernel_ms_ago(ago)
        br_ms.append(b)
        ks_ms.append(s)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if check_bits:
        # rows the CPU legs will re-compute: first, middle, last for the single-thread leg (N = 1), one row per usable host
        # core for the all-cores leg; at N > 1 every rank checks one row of ITS shard
        cores = usable_cores()["usable"]
        single = sorted({0, batch // 2 - 1 if batch > 1 else 0, batch - 1}) if world == 1 else [min(batch - 1, rank * 7919 % batch)]
        rows = sorted(set(single) | (set(spaced_rows(batch, cores)) if world == 1 else set()))
        sel = torch.tensor(rows, device=dev, dtype=torch.long)
        host.update({"rows": rows, "single_rows": single,
                     "lwe_rows": lwe.index_select(0, sel).cpu().numpy().view(np.uint32),
                     "gpu_rows": out.index_select(0, sel).cpu().numpy().view(np.uint32)})

    total = batch * world * args.steps
    value = total / dt
    br_avg = float(np.mean(br_ms))
    ks_avg = float(np.mean(ks_ms))
    ext_products = batch * n  # per launch
    algo_bytes = ext_products * params.external_product_bytes()
    achieved = algo_bytes / (br_avg * 1e-3) / 1e9
    kernel_name = f"blind_rotate_kernel<{backend_name},{logn},{k}>"
    traffic, traffic_source, valu_view = pmc_traffic(kernel_name, args.workload, batch)
    hbm_copy = measure_hbm_copy_gbs(torch, dev, ctx=ctx)
    result = {
        "metric": "homomorphic_gates_per_sec" if args.gate else "programmable_bootstraps_per_sec",
        "value": value,
        "unit": "gates/s" if args.gate else "PBS/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64" if backend_name.startswith("fp64") else "u64",
        "dtype_note": ("complex FFT in fp64, exact by a proven rounding-error bound (csrc/field_fft.h); same bits as the exact-NTT backends"
                       if backend_name == "fp64-fft"
                       else "exact NTT over the 49-bit prime 671317819555841 in fp64" if backend_name == "fp64-p49"
                       else "exact NTT over the 42-bit prime 2^42-24575 in fp64" if backend_name.startswith("fp64")
                       else "exact NTT over the Goldilocks prime in u64") + "; ciphertext words are wrapping u32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.workload}: batch {batch}/GPU, N={1 << logn}, k={k}, n={n}, l={pbs[1]}, log2B={pbs[0]}, "
                        f"KS l={ks[1]} log2B={ks[0]}, log_p={log_p}, " + (f"{args.gate.upper()} gate stream" if args.gate else "identity LUT"),
            "global_batch": batch * world,
            "parallelism": f"dp{world} (independent LWE shards resident per GPU, no data-path collective; keys: {key_replication}"
                           + ("; batch scattered from / gathered to rank 0 over RCCL every step)" if args.scatter_gather else ")"),
        },
        "roofline": {
            "kernel": kernel_name,
            # what binds the kernel (PMC: counter traffic is a tenth of the algorithmic bytes); achieved/peak/frac
            # stay priced against the HBM roofline, which is the metric BASELINE.json asks for
            "bound": "valu-issue",
            "priced_against": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "valu": valu_view,  # VALU-issue view from the same PMC passes (SURVEY 8d asks for both)
            "hbm_copy_measured_GBps": hbm_copy,
            "frac_of_measured_hbm": achieved / hbm_copy,
            # kernel_ms: HIP events on the context's stream around ALL blind-rotation launches of one step (launch_plan:
            # a batch larger than the chip goes out as `segments` launches per rotation -- one slice of the key each -- on
            # `streams` alternating streams, joined before the closing event); "per launch" below = per step's rotations
            "kernel_ms": br_avg,
            "launch_plan": ctx.blind_rotate_plan(batch),
            "algorithmic_bytes_per_launch": algo_bytes,
            "external_products_per_s": ext_products / (br_avg * 1e-3),
            "key_switch_kernel_ms": ks_avg,
            "note": ("fp64-fft: SIMD issue -- VALU instructions about three quarters of the time, the LDS traffic of the transposes most of the rest (DESIGN.md 2); "
                     if backend_name == "fp64-fft" else "VALU-issue bound by design (SURVEY 8d); ")
                    + "the key is shared by the batch and stays in L2/Infinity Cache; HBM fraction reported as the metric asks",
        },
    }
    if use_dist and world > 1 and args.scatter_gather_figure and not args.scatter_gather and not args.gate:
        # second figure, outside the timed region above: the same step with rank 0 owning the whole
        # batch -- RCCL scatter of [B/N][n+1] rows, local bootstrap, gather (sharding.bootstrap_sharded)
        import importlib
        sharding = importlib.import_module("tfhe_research_amd.sharding")
        full = rand_words(batch * world, n + 1) if rank == 0 else None

        def sg_step():
            sharding.bootstrap_sharded(lambda shard, tvv: ctx.bootstrap(shard, tvv), full, tv, root=0,
                                       batch=batch * world, width=n + 1, like=lwe)
        sg_step()
        barrier()
        sg_steps = max(2, min(args.steps, 5))
        t1 = time.perf_counter()
        for _ in range(sg_steps):
            sg_step()
        barrier()
        t_sg = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        dist.all_reduce(t_sg, op=dist.ReduceOp.MAX)
        result["scatter_gather"] = {
            "value": batch * world * sg_steps / float(t_sg.item()), "unit": "PBS/s", "steps": sg_steps,
            "ms_per_step": float(t_sg.item()) / sg_steps * 1e3,
            "what": f"rank 0 holds all {batch * world} ciphertexts: isend/recv scatter of {batch * (n + 1) * 4 / 1e6:.0f} MB per peer "
                    "over RCCL/xGMI, local bootstrap, gather back; NOT the headline value"}
    if plain_pbs and not args.no_secondary_legs and pbs[0] * (32 // pbs[0]) != 32:
        # log2 B does not divide 32 (cfg2: 7): with the reference's literal decomposer the top 32 mod log2 B bits of a
        # word are never decomposed, a trivially encrypted accumulator has no bit below them, every digit is zero and
        # the blind rotation never depends on the key -- in the reference too (decomposer.rs:42-80).  The
        # timed region above therefore multiplies zeros.  The kernels have no data-dependent branch; this leg puts
        # that on record: the same step with the aligned decomposer, where every CMUX depends on key and data, with the
        # headline's warm-up and step counts, its own roofline block, and its rows checked against the oracle below.
        try:
            ctx.set_decomposer_alignment(True)
            for _ in range(max(1, args.warmup)):
                ctx.bootstrap(lwe, tv, out=out)
            barrier()
            t_al = time.perf_counter()
            for _ in range(args.steps):
                ctx.bootstrap(lwe, tv, out=out)
            barrier()
            dt_al = time.perf_counter() - t_al
            if use_dist:
                t = torch.tensor([dt_al], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt_al = float(t.item())
            al_ms = float(np.mean([ctx.kernel_ms_ago(i)[0] for i in range(min(args.steps, 64))]))
            distinct = int(torch.unique(out[:256]).numel())
            al_achieved = algo_bytes / (al_ms * 1e-3) / 1e9
            result["aligned_decomposer"] = {
                "kernel_ms": al_ms,
                "value": batch * world * args.steps / dt_al, "unit": "PBS/s", "steps": args.steps, "warmup": max(1, args.warmup),
                "ms_per_step": dt_al / args.steps * 1e3,
                "roofline": {"kernel": kernel_name, "bound": "valu-issue", "priced_against": "hbm", "achieved": al_achieved,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": al_achieved / HBM_PEAK_GBS,
                             "frac_of_measured_hbm": al_achieved / hbm_copy, "kernel_ms": al_ms,
                             "algorithmic_bytes_per_launch": algo_bytes, "traffic": None},
                "literal_kernel_ms": br_avg,
                "distinct_output_words_in_256_rows": distinct,
                "what": "same step, tfhe_context_set_decomposer_alignment(1): digits are non-zero and the rotation depends on "
                        "the key (the literal cfg2 decomposer yields all-zero digits, in the reference as well); the kernels "
                        "have no data-dependent branch: what difference there is is the clock the chip sustains "
                        "with non-zero operands in the fp64 datapath, not skipped work.  THIS is the engine's cfg2 rate on "
                        "data that exercises the datapath"}
            if check_bits:
                sel = torch.tensor(host["rows"], device=dev, dtype=torch.long)
                host["aligned_gpu_rows"] = out.index_select(0, sel).cpu().numpy().view(np.uint32)
            ctx.set_decomposer_alignment(False)
        except Exception as e:  # noqa: BLE001 - a secondary figure must not lose the benchmark line
            result["aligned_decomposer"] = {"error": str(e)}
    ctx.close()
    if world == 1 and backend_name == "fp64-fft" and not args.gate and args.backend == "auto" and not args.no_secondary_legs:
        # beside the headline (never instead of it): the same step in the exact prime-field NTT the complex-FFT backend
        # replaced as the default (same ciphertexts and key-switching key, a fresh random bootstrapping key: the time does
        # not depend on the data), so that both arithmetic routes are in one record
        try:
            try:  # the fastest exact prime field that lifts this parameter set: 49-bit (one spectrum), else 42-bit
                ctx2 = pkg.Context(params, device=local_rank, backend=pkg.BACKEND_FP64_P49)
            except pkg.TfheError:
                ctx2 = pkg.Context(params, device=local_rank, backend=pkg.BACKEND_FP64)
            ctx2.use_torch_stream()
            ctx2.load_bootstrapping_key(rand_words(*params.bsk_shape()), ksk)
            ctx2.reserve(batch)
            ctx2.set_timing(True)
            out2 = torch.empty_like(lwe)
            ctx2.bootstrap(lwe, tv, out=out2)
            barrier()
            t2 = time.perf_counter()
            for _ in range(3):
                ctx2.bootstrap(lwe, tv, out=out2)
            barrier()
            dt2 = time.perf_counter() - t2
            result["exact_ntt_backend"] = {
                "backend": ctx2.backend, "value": batch * 3 / dt2, "unit": "PBS/s", "steps": 3,
                "kernel_ms": float(np.mean([ctx2.kernel_ms_ago(i)[0] for i in range(3)])),
                "what": "the same step in the fastest exact prime-field NTT that lifts this parameter set (49-bit, else 42-bit: "
                        "exact integer arithmetic); not the headline value"}
            ctx2.close()
        except Exception as e:  # noqa: BLE001 - a secondary figure must not lose the benchmark line
            result["exact_ntt_backend"] = {"error": str(e)}
    # who took part: every rank reports the world size it saw and its device (the first multi-GPU run must show
    # that N distinct devices really ran)
    seen = f"rank {rank}/{dist.get_world_size() if use_dist else 1}: cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}"
    if use_dist:
        gathered = [None] * world
        dist.all_gather_object(gathered, seen)
        seen = gathered
    else:
        seen = [seen]
    result["config"]["ranks_seen"] = seen
    failed = False
    if check_bits:
        if world == 1:
            result["cpu_baseline"], result["verified"] = cpu_legs(args.workload, args.cpu_baseline_seconds, host)
            failed = not result["verified"]["bit_exact"]
        else:
            # N > 1: no CPU baseline (rank 0 at N = 1 only), but every rank checks one row of ITS shard (schoolbook product)
            from oracle import oracle as orc
            orc.build()
            orc.set_poly_mul_mode(1)
            p = orc.Params(k, logn, n, orc.Decomposer(*pbs), orc.Decomposer(*ks), log_p=log_p)
            ok = bool(np.array_equal(orc.bootstrap(p, host["lwe_rows"][0], host["bsk"], host["ksk"], host["tv"]), host["gpu_rows"][0]))
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            result["verified"] = {"rows": world, "bit_exact": bool(flag.item()),
                                  "what": "one row of every rank's shard of the timed batch re-computed by the oracle on that rank's host"}
            failed = not result["verified"]["bit_exact"]
    elif plain_pbs:
        result["verified"] = None  # --no-cpu-baseline: nothing was re-computed
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: the GPU's output differs from the oracle's on rows of the timed batch (see `verified`)")


if __name__ == "__main__":
    main()

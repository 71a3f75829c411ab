#!/usr/bin/env python3
"""After tools/final_profile.sh ran on the GPU box (gpurun merges gpurun_out/final/ back): copy the bench lines, rocprofv3
stats and PMC summaries that are to be judged into profiles/<tag>_*, refresh profiles/pmc_traffic.json (stamped with the
kernel source hash bench.py checks) and print a one-line summary per workload.
    python tools/collect_profiles.py r03_f
The instruction mixes per wave and CMUX iteration (fma, mul, add, rndne, cvt, integer) are read off tools/isa_report.py's
loop report of the shipped library (profiles/<round>_isa_blind_rotate_loops.txt)."""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
F, P = os.path.join(ROOT, "gpurun_out", "final"), os.path.join(ROOT, "profiles")
BENCH = {"cfg2": "cfg2", "cfg2_fp64_p42": "cfg2_fp64_p42", "cfg4": "cfg4", "cfg3_nand": "cfg3_nand", "cfg3": "cfg3", "cfg5": "cfg5",
         "cfg1": "cfg1", "pool_1member": "pool_1member", "cfg2_torchrun1": "cfg2_torchrun1",
         "ep_shared": "external_product_shared_ggsw", "ep_shared_64k": "external_product_shared_ggsw_batch65536",
         "ep_streamed": "external_product_streamed_ggsw"}
for src, dst in BENCH.items():
    shutil.copy(os.path.join(F, f"bench_{src}.json.log"), os.path.join(P, f"{tag}_bench_{dst}.json.log"))
for src, dst in (("kernel_stats_cfg2.csv", "kernel_stats_cfg2_default_bench.csv"),
                 ("kernel_stats_cfg2_timed_only.csv", "kernel_stats_cfg2_timed_only.csv"),
                 ("kernel_stats_ep.csv", "kernel_stats_external_product.csv")):
    shutil.copy(os.path.join(F, src), os.path.join(P, f"{tag}_{dst}"))



def span_summary(trace_csv, stats_line, steps, out_path, what):
    """A step's blind rotations are many launches on two streams: rocprofv3's per-kernel average is per launch and its sum counts
    overlapped time twice.  What compares with bench.py's event-timed kernel_ms is the SPAN of a step's launches."""
    rows = [r for r in csv.DictReader(open(trace_csv)) if "blind_rotate_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = len(rows) // steps
    text = [f"# {what}: rocprofv3 --kernel-trace of bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary-legs",
            f"# {len(rows)} blind_rotate_kernel dispatches = {steps} steps x {per} launches (key slices x streams: kernels.hip::blind_rotate_plan)",
            "# step  launches  span_ms (first start -> last end)  sum_of_durations_ms  mean_launch_ms"]
    spans = []
    for i in range(steps):
        part = rows[i * per:(i + 1) * per]
        t0 = min(int(r["Start_Timestamp"]) for r in part)
        t1 = max(int(r["End_Timestamp"]) for r in part)
        total = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in part)
        spans.append((t1 - t0) / 1e6)
        text.append(f"{i}  {len(part)}  {(t1 - t0) / 1e6:.3f}  {total / 1e6:.3f}  {total / 1e6 / len(part):.4f}")
    text.append(f"# mean span of the 5 timed steps: {sum(spans[-5:]) / 5:.3f} ms   (bench.py of the same build, HIP events: {stats_line})")
    open(out_path, "w").write("\n".join(text) + "\n")
    print(text[-1])


for wl in ("cfg2", "cfg3"):
    line = [l for l in open(os.path.join(F, f"bench_{wl}.json.log")) if l.startswith("{")][-1]
    span_summary(os.path.join(F, f"kernel_trace_{wl}_timed_only.csv"), f"kernel_ms {json.loads(line)['roofline']['kernel_ms']:.3f}", 7,
                 os.path.join(P, f"{tag}_kernel_span_{wl}.txt"), wl)
shutil.copy(os.path.join(F, "kernel_stats_cfg3_timed_only.csv"), os.path.join(P, f"{tag}_kernel_stats_cfg3_timed_only.csv"))

PMC = [  # directory, label, batch, n, algorithmic bytes per product, kernel substring, mix, output, extra args
    ("pmc", "cfg2 batch 4096", 4096, 630, 65536, "blind_rotate_kernel<tfhe::FftField, 10, 1>", "1176,144,320,0,48,561",
     "pmc_blind_rotate_cfg2_fp64_fft.txt", ["--json", os.path.join(P, "pmc_traffic.json"), "--bench-kernel", "blind_rotate_kernel<fp64-fft,10,1>"]),
    ("pmc_one_launch", "cfg2 batch 4096 in ONE launch (TFHE_BR_SEGMENTS=1 TFHE_BR_STREAMS=1: the kernel by itself)", 4096, 630, 65536,
     "blind_rotate_kernel<tfhe::FftField, 10, 1>", "1176,144,320,0,48,561", "pmc_blind_rotate_cfg2_fp64_fft_one_launch.txt", []),
    ("pmc_cfg5", "cfg5 batch 4096", 4096, 630, 344064, "blind_rotate_kernel<tfhe::FftField, 11, 2>", "1888,160,352,0,64,979",
     "pmc_blind_rotate_cfg5_fp64_fft.txt", []),
    ("pmc_cfg3", "cfg3 batch 4096", 4096, 722, 122880, "blind_rotate_kernel<tfhe::FftField, 9, 2>", "2432,128,288,0,96,966",
     "pmc_blind_rotate_cfg3_fp64_fft.txt", []),
    ("pmc_ep", "external_product cfg2 batch 4096, one GGSW shared", 4096, 1, 65536, "external_product_kernel<tfhe::FftField, 10, 1>",
     "1176,144,320,0,48,561", "pmc_external_product_cfg2_shared.txt", []),
]
for d, label, batch, n, algo, kernel, mix, out, extra in PMC:
    path = os.path.join(P, f"{tag}_{out}")
    cmd = [sys.executable, os.path.join(ROOT, "tools", "pmc_summarize.py"), os.path.join(F, d), label, str(batch), str(n), str(algo),
           "--kernel", kernel, "--mix", mix, "--source", f"profiles/{tag}_{out}"] + extra + (["--steps", "1"] if "blind_rotate" in kernel else [])
    text = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    open(path, "w").write(text)
    print(out, *[l for l in text.splitlines() if l.startswith("# derived: kernel") or l.startswith("# derived: fabric")], sep="\n  ")

for src in BENCH:
    line = [l for l in open(os.path.join(F, f"bench_{src}.json.log")) if l.startswith("{")][-1]
    r = json.loads(line)
    rf = r["roofline"]
    al = r.get("aligned_decomposer") or {}
    print(f"{src:16s} {r['value']:12.0f} {r['unit']:10s} kernel {rf['kernel_ms']:9.3f} ms  frac {rf['frac']:.3f}  traffic {rf.get('traffic')}"
          + (f"  aligned {al.get('value', 0):.0f} ({al.get('kernel_ms', 0):.2f} ms)" if al else ""))

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

PMC = [  # directory, label, batch, n, algorithmic bytes per product, kernel substring, mix, output, extra args
    ("pmc", "cfg2 batch 4096", 4096, 630, 65536, "blind_rotate_kernel<tfhe::FftField, 10, 1>", "1176,144,320,0,48,561",
     "pmc_blind_rotate_cfg2_fp64_fft.txt", ["--json", os.path.join(P, "pmc_traffic.json"), "--bench-kernel", "blind_rotate_kernel<fp64-fft,10,1>"]),
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
           "--kernel", kernel, "--mix", mix, "--source", f"profiles/{tag}_{out}"] + extra
    text = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    if d == "pmc":  # literal launch against the aligned leg's launches: same cycles, the time differs by clock
        rows = list(csv.DictReader(open(os.path.join(F, d, "sq2.csv"))))
        text += ("# per dispatch (bench.py --steps 1 --warmup 0: the first launch is the timed literal-decomposer step, the next one the\n"
                 "# aligned leg's warm-up, the last three the aligned leg's timed launches): GPU cycles = GRBM_GUI_ACTIVE / 8 XCDs, in millions\n")
        for r in rows:
            if kernel.split("<")[1] in r["Kernel_Name"] and "blind_rotate" in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                text += f"# dispatch {r['Dispatch_Id']}: {float(r['Counter_Value']) / 8 / 1e6:.1f} M cycles\n"
    open(path, "w").write(text)
    print(out, *[l for l in text.splitlines() if l.startswith("# derived: kernel") or l.startswith("# derived: fabric")], sep="\n  ")

for src in BENCH:
    line = [l for l in open(os.path.join(F, f"bench_{src}.json.log")) if l.startswith("{")][-1]
    r = json.loads(line)
    rf = r["roofline"]
    al = r.get("aligned_decomposer") or {}
    print(f"{src:16s} {r['value']:12.0f} {r['unit']:10s} kernel {rf['kernel_ms']:9.3f} ms  frac {rf['frac']:.3f}  traffic {rf.get('traffic')}"
          + (f"  aligned {al.get('value', 0):.0f} ({al.get('kernel_ms', 0):.2f} ms)" if al else ""))

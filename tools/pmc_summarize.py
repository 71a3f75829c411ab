"""Sums the counters of the dominant kernel over the rocprofv3 --pmc passes that
tools/profile_pmc.sh leaves under gpurun_out/<tag>/ and prints the summary committed under
profiles/ (plus, with --json, the traffic figure bench.py reports in roofline.traffic).
    python tools/pmc_summarize.py gpurun_out/final/pmc "cfg2 batch 4096" 4096 630 65536 [--json profiles/pmc_traffic.json]
"""
import csv
import json
import sys
from collections import defaultdict

d, label, batch, n, algo_bytes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
tot = defaultdict(float)
launches = defaultdict(set)
kernel = None
for name in ("fetch", "write", "sq", "sq2"):
    try:
        rows = list(csv.DictReader(open(f"{d}/{name}.csv")))
    except FileNotFoundError:
        continue
    for r in rows:
        if "blind_rotate" not in r["Kernel_Name"]:
            continue
        kernel = r["Kernel_Name"].split("(")[0]
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Counter_Name"]].add(r["Dispatch_Id"])
print(f"# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2), bench.py --steps 1, {label}")
print("# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM);")
print("# both count the L2's fabric-side requests, Infinity-Cache hits included (same guide).  Values are per launch.")
print(f"# kernel: {kernel}")
per = {}
for k in sorted(tot):
    per[k] = tot[k] / max(1, len(launches[k]))
    print(f"{k}\t{per[k]:.0f}")
products = batch * n
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    traffic = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
    print(f"# derived: fabric traffic per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {traffic / 1e9:.1f} GB "
          f"(algorithmic {products * algo_bytes / 1e9:.1f} GB)")
if "SQ_INSTS_VALU" in per:
    print(f"# derived: VALU instructions per external product = {per['SQ_INSTS_VALU'] / products:.0f}")
if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
    cyc = per["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
    print(f"# derived: kernel = {cyc / 1e6:.1f} M cycles; cycles per VALU instruction per SIMD = "
          f"{cyc * 1024 / per['SQ_INSTS_VALU']:.2f}")
if "--json" in sys.argv and "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    out = sys.argv[sys.argv.index("--json") + 1]
    valu = {}
    if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
        cpi = per["GRBM_GUI_ACTIVE"] / 8 * 1024 / per["SQ_INSTS_VALU"]
        valu = {"valu_insts_per_external_product": per["SQ_INSTS_VALU"] / products,
                "cycles_per_valu_inst_per_simd": cpi,
                # issue cost of this kernel's mix at 2 waves per SIMD: 84 % fp64 ops at 5.3 cycles,
                # 16 % integer ops at 4.4 (profiles/r01_dp_chain_latency_gfx950.txt,
                # profiles/r01_valu_issue_rates_gfx950.txt; the split is DESIGN.md section 4)
                "issue_floor_cycles_per_inst": 0.84 * 5.3 + 0.16 * 4.4,
                "valu_issue_frac": min(1.0, (0.84 * 5.3 + 0.16 * 4.4) / cpi),
                "floor_source": "microbenchmarks under profiles/ (fp64 5.3, integer 4.4 cycles per wave-instruction at 2 waves per SIMD)"}
    # bench.py matches on "workload" == "<name> batch <batch>": keep it to exactly that
    json.dump({"kernel": "blind_rotate_kernel<fp64-p42,10,1>", "workload": label.split(",")[0], "valu": valu,
               "fetch_size_kib": per["FETCH_SIZE"], "write_size_kib": per["WRITE_SIZE"],
               "traffic_bytes_per_launch": traffic,
               "note": "L2 fabric-side requests (Infinity-Cache hits included); varies with the drift of the teams inside an XCD",
               "source": "profiles/r01_h_pmc_blind_rotate_cfg2_fp64.txt"}, open(out, "w"), indent=1)

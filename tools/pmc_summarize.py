"""Sums the counters of the dominant kernel over the rocprofv3 --pmc passes that
tools/profile_pmc.sh leaves under gpurun_out/<tag>/ and prints the summary committed under
profiles/ (plus, with --json, the traffic figure bench.py reports in roofline.traffic).
    python tools/pmc_summarize.py gpurun_out/final/pmc "cfg2 batch 4096" 4096 630 65536 [--json profiles/pmc_traffic.json]
        [--kernel external_product] [--source profiles/<the file this output is committed as>]
        [--steps 1: the pass ran that many steps and a step's blind rotations are SEVERAL launches (key slices, two streams:
        kernels.hip::blind_rotate_plan) -- sum the dispatches and divide by the steps instead of averaging per dispatch]
        [--bench-kernel 'blind_rotate_kernel<fp64-fft,10,1>'] [--mix fma,mul,add,rndne,cvt,int]  (instruction mix per wave
        and iteration from tools/isa_report.py, for the issue floor; default: the 42-bit field's cfg2 kernel)
(for the standalone external-product kernel pass n = 1: one product per sample and launch)
"""
import csv
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

d, label, batch, n, algo_bytes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
want = sys.argv[sys.argv.index("--kernel") + 1] if "--kernel" in sys.argv else "blind_rotate"
source = sys.argv[sys.argv.index("--source") + 1] if "--source" in sys.argv else ""
# issue floor of the cfg2 kernel's instruction mix at 2 waves per SIMD, from the per-instruction rates of
# profiles/r02_valu_issue_rates_gfx950.txt (warm clocks: v_fma_f64 4.78, v_mul_f64 4.48, v_add_f64 4.48,
# v_rndne_f64 4.34, 32-bit integer ops ~4.4) weighted with the shipped kernel's mix
# (profiles/r02_d_isa_blind_rotate_loops.txt: per wave and iteration 1200 fma, 1140 mul, 1472 add, 584 rndne,
# 48 cvt, 643 integer)
MIX = [int(x) for x in (sys.argv[sys.argv.index("--mix") + 1] if "--mix" in sys.argv else "1200,1140,1472,584,48,643").split(",")]
FLOOR = (MIX[0] * 4.78 + MIX[1] * 4.48 + MIX[2] * 4.48 + MIX[3] * 4.34 + MIX[4] * 4.8 + MIX[5] * 4.4) / sum(MIX)
BENCH_KERNEL = sys.argv[sys.argv.index("--bench-kernel") + 1] if "--bench-kernel" in sys.argv else "blind_rotate_kernel<fp64-p42,10,1>"
STEPS = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 0
tot = defaultdict(float)
launches = defaultdict(set)
kernel = None
for name in ("fetch", "write", "sq", "sq2"):
    try:
        rows = list(csv.DictReader(open(f"{d}/{name}.csv")))
    except FileNotFoundError:
        continue
    for r in rows:
        if want not in r["Kernel_Name"] or "bmmp" in r["Kernel_Name"]:
            continue
        kernel = r["Kernel_Name"].split("(")[0]
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        launches[r["Counter_Name"]].add(r["Dispatch_Id"])
print(f"# rocprofv3 --pmc passes (separate runs: FETCH_SIZE | WRITE_SIZE | SQ set 1 | SQ set 2), bench.py --steps 1, {label}")
print("# FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 1/2 of wide coalesced reads (MI355X_MICROARCH.md, HBM);")
print("# both count the L2's fabric-side requests, Infinity-Cache hits included (same guide).  Values are per "
      + ("launch." if not STEPS else "STEP = the sum over all blind-rotation dispatches of one step"))
print(f"# kernel: {kernel}")
per = {}
for k in sorted(tot):
    per[k] = tot[k] / (STEPS if STEPS else max(1, len(launches[k])))
    print(f"{k}\t{per[k]:.0f}")
if STEPS:
    counts = sorted({len(v) for v in launches.values()})
    per_step = counts[0] // STEPS
    print(f"# dispatches per step: {per_step}" + (" (a rotation is cut into key slices, the halves of the batch alternate on two streams; under "
          "--pmc the dispatches run one after the other, so the cycle counts below include every launch's tail, which the two "
          "streams hide in a normal run)" if per_step > 1 else " (the whole rotation of the batch in one launch)"))
products = batch * n
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    traffic = (2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024
    print(f"# derived: fabric traffic per {'step' if STEPS else 'launch'} = (2*FETCH_SIZE + WRITE_SIZE)*1024 = {traffic / 1e9:.1f} GB "
          f"(algorithmic {products * algo_bytes / 1e9:.1f} GB)")
if "SQ_INSTS_VALU" in per:
    print(f"# derived: VALU instructions per external product = {per['SQ_INSTS_VALU'] / products:.0f}")
if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
    cyc = per["GRBM_GUI_ACTIVE"] / 8  # summed over the 8 XCDs
    print(f"# derived: kernel = {cyc / 1e6:.1f} M cycles; cycles per VALU instruction per SIMD = "
          f"{cyc * 1024 / per['SQ_INSTS_VALU']:.2f} (issue floor of this mix {FLOOR:.2f}: VALU issue at {min(1.0, FLOOR * per['SQ_INSTS_VALU'] / (cyc * 1024)):.2f})")
if "SQ_WAIT_ANY" in per and "SQ_WAVE_CYCLES" in per:
    print(f"# derived: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {per['SQ_WAIT_ANY'] / per['SQ_WAVE_CYCLES']:.2f} (share of a wave's cycles parked in s_waitcnt / barriers)")
if "SQ_INSTS_VMEM_RD" in per:
    print(f"# derived: vector-memory read instructions per external product = {per['SQ_INSTS_VMEM_RD'] / products:.1f}")
if "--json" in sys.argv and "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    out = sys.argv[sys.argv.index("--json") + 1]
    valu = {}
    if "GRBM_GUI_ACTIVE" in per and "SQ_INSTS_VALU" in per:
        cpi = per["GRBM_GUI_ACTIVE"] / 8 * 1024 / per["SQ_INSTS_VALU"]
        valu = {"valu_insts_per_external_product": per["SQ_INSTS_VALU"] / products,
                "cycles_per_valu_inst_per_simd": cpi,
                "issue_floor_cycles_per_inst": FLOOR,
                "valu_issue_frac": min(1.0, FLOOR / cpi),
                "floor_source": "per-instruction issue rates at 2 waves per SIMD (profiles/r02_valu_issue_rates_gfx950.txt) "
                                "weighted with the shipped kernel's instruction mix (profiles/r02_*_isa_blind_rotate_loops.txt): "
                                + ",".join(str(x) for x in MIX) + " fma,mul,add,rndne,cvt,integer per wave and iteration"}
    # bench.py matches on "workload" == "<name> batch <batch>": keep it to exactly that
    from bench import kernel_source_hash  # the kernels this record was measured on: bench.py refuses it for any other
    json.dump({"kernel": BENCH_KERNEL, "workload": label.split(",")[0], "kernel_source_hash": kernel_source_hash(), "valu": valu,
               "fetch_size_kib": per["FETCH_SIZE"], "write_size_kib": per["WRITE_SIZE"],
               "traffic_bytes_per_launch": traffic,
               "note": "L2 fabric-side requests (Infinity-Cache hits included); varies with the drift of the teams inside an XCD",
               "source": source}, open(out, "w"), indent=1)

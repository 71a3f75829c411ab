"""Effective shader clock of the blind-rotation launches, per leg of one bench.py run under
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-include-regex blind_rotate -d DIR -o run --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline
usage: effective_clock.py DIR/.../run_counter_collection.csv [launches per step] [kernel name filter, default FftField]
GRBM_GUI_ACTIVE is summed over the XCDs (8 on gfx950); a launch's cycles / 8 over its End - Start time is the clock the
launch really ran at.  Launches are taken in dispatch order: step 0 = the literal decomposer's timed step, the following
steps = the aligned decomposer's (every digit and rotation depends on key and data)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "blind_rotate" in r["Kernel_Name"]
        and (sys.argv[3] if len(sys.argv) > 3 else "FftField") in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
per = int(sys.argv[2]) if len(sys.argv) > 2 else 116
print(f"# {len(rows)} blind_rotate launches, {per} per step; XCDs assumed: 8")
for s in range(0, len(rows), per):
    leg = rows[s:s + per]
    cyc = sum(float(r["Counter_Value"]) for r in leg) / 8
    ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in leg)
    print(f"step {s // per} ({'literal' if s == 0 else 'aligned'}): {len(leg)} launches, {cyc / 1e6:7.2f} M cycles in {ns / 1e6:7.2f} ms "
          f"(launches serialised by the profiler) -> {cyc / ns * 1e3:6.0f} MHz effective")

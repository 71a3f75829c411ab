"""GPU box: shader clock and board power while the cfg2 step runs with the literal decomposer (all digits zero)
and with the aligned one (every digit and rotation depends on key and data).  The PMC record of round 3
(profiles/r03_h_cycles_literal_vs_aligned_cfg2.txt) shows the same CYCLES for both; this shows where the time goes.
usage: clock_watch.py [cfg2|cfg3|cfg1] [seconds per leg]
Samples come from the amdgpu sysfs nodes of the card (hwmon power1_average / power1_input, freq1_input or pp_dpm_sclk);
when the box hides them, from `rocm-smi --showclocks --showpower --json` (slower: a few samples per leg)."""
import glob, json, os, subprocess, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
m = g.load_package()

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
leg_s = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
k, logn, n, pbs, batch = {"cfg1": (1, 9, 500, (8, 2), 4096), "cfg2": (1, 10, 630, (7, 3), 4096),
                          "cfg3": (2, 9, 722, (4, 6), 4096)}[which]


def read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def sysfs_nodes():
    """(sclk reader, power reader) over the first card that answers; None where the node is hidden"""
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        hw = sorted(glob.glob(card + "/hwmon/hwmon*"))
        if not hw:
            continue
        h = hw[0]
        power = next((p for p in (h + "/power1_average", h + "/power1_input") if read(p)), None)
        freq = h + "/freq1_input" if read(h + "/freq1_input") else None
        dpm = card + "/pp_dpm_sclk" if read(card + "/pp_dpm_sclk") else None
        if power or freq or dpm:
            return card, power, freq, dpm
    return None, None, None, None


CARD, POWER, FREQ, DPM = sysfs_nodes()


def sample_sysfs():
    mhz = w = None
    if FREQ:
        mhz = int(read(FREQ)) / 1e6
    elif DPM:
        for line in read(DPM).splitlines():
            if line.rstrip().endswith("*"):
                mhz = float(line.split(":")[1].strip().rstrip("*").strip().lower().replace("mhz", ""))
    if POWER:
        w = int(read(POWER)) / 1e6
    return mhz, w


def sample_smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
        card = next(iter(json.loads(out).values()))
    except Exception:
        return None, None
    mhz = w = None
    for key, val in card.items():
        lk = key.lower()
        if "sclk" in lk and "(" in str(val):
            mhz = float(str(val).split("(")[1].split("M")[0])
        if "power" in lk and "socket" in lk or "average graphics package power" in lk:
            try:
                w = float(val)
            except ValueError:
                pass
    return mhz, w


USE_SYSFS = bool(POWER or FREQ or DPM)


class Watch(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.rows = []

    def run(self):
        while not self.stop:
            self.rows.append(sample_sysfs() if USE_SYSFS else sample_smi())
            time.sleep(0.02 if USE_SYSFS else 0.2)


P = m.TfheParams(k, logn, n, m.DecomposerParams(*pbs))
dev = torch.device("cuda", 0)
gen = torch.Generator(device=dev); gen.manual_seed(1)
rw = lambda *s: torch.randint(-(1 << 31), (1 << 31) - 1, s, dtype=torch.int32, device=dev, generator=gen)
lw, bk, kk = rw(batch, n + 1), rw(*P.bsk_shape()), rw(*P.ksk_shape())
tvd = torch.from_numpy(m.construct_identity_test_vector(P).astype(np.int32)).to(dev)
ctx = m.Context(P); ctx.use_torch_stream(); ctx.load_bootstrapping_key(bk, kk); ctx.reserve(batch); ctx.set_timing(True)
out = torch.empty_like(lw)
print(f"# {which} batch {batch}, {leg_s:.0f} s per leg; samples from {'sysfs ' + str(CARD) if USE_SYSFS else 'rocm-smi'}", flush=True)
print(f"# idle: sclk {sample_sysfs()[0] if USE_SYSFS else sample_smi()[0]} MHz", flush=True)
for rep in range(2):
    for name, aligned in (("literal", False), ("aligned", True)):
        ctx.set_decomposer_alignment(aligned)
        ctx.bootstrap(lw, tvd, out=out); torch.cuda.synchronize()
        w = Watch(); w.start()
        t0 = time.time(); kernel = []
        while time.time() - t0 < leg_s:
            ctx.bootstrap(lw, tvd, out=out); kernel.append(sum(ctx.last_kernel_ms()))
        torch.cuda.synchronize()
        w.stop = True; w.join()
        rows = w.rows[len(w.rows) // 4:]  # the first quarter of a leg is the clock settling
        mhz = [r[0] for r in rows if r[0]]; pw = [r[1] for r in rows if r[1]]
        fmt = lambda v: f"{np.mean(v):7.0f} (min {np.min(v):.0f}, max {np.max(v):.0f}, {len(v)} samples)" if v else "    n/a"
        print(f"{name:8s} round {rep}: kernels {np.mean(kernel[len(kernel) // 4:]):6.2f} ms per step | sclk MHz {fmt(mhz)} | power W {fmt(pw)}", flush=True)

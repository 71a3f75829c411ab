#!/usr/bin/env python3
"""ISA-level evidence of the SHIPPED library: per-kernel resource usage and a loop-aware instruction
histogram, read back from tfhe-research_amd/libtfhe_hip.so (not from a side compile).

    python tools/isa_report.py                                  # resource table of every kernel
    python tools/isa_report.py --kernel 'blind_rotate_kernel<tfhe::FpField, 10, 1>' [--asm out.s]

How: the gfx950 code object is unbundled from the .hip_fatbin section (llvm-objcopy +
clang-offload-bundler), its AMDGPU metadata note gives VGPR / SGPR / AGPR counts, spills, scratch
and LDS per kernel, llvm-objdump disassembles one kernel, and backward branches give the loop nest.
For each loop the static instruction mix of its body is printed, so that dynamic counts (PMC
SQ_INSTS_VALU per launch) can be reconciled:  dynamic = sum over loops of body x trip count.
Runs without a GPU.  Output goes to stdout; commit what you want judged under profiles/.
"""
from __future__ import annotations

import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "tfhe-research_amd", "libtfhe_hip.so")


def run(*cmd, **kw):
    return subprocess.run(cmd, check=True, capture_output=True, text=True, **kw).stdout


def extract_code_object(lib: str, tmp: str) -> str:
    fat = os.path.join(tmp, "fat.bin")
    co = os.path.join(tmp, "dev.co")
    run(f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat)
    run(f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}")
    return co


def kernel_table(co: str):
    """-> list of dicts from the AMDGPU metadata note (one per kernel)"""
    notes = run(f"{LLVM}/llvm-readelf", "--notes", co)
    kernels, cur = [], None
    for line in notes.splitlines():
        m = re.match(r"\s+(- )?\.(\w+):\s+(.*)$", line)
        if not m:
            continue
        dash, key, val = m.groups()
        if dash and key == "agpr_count":  # first key of a kernel entry (keys are sorted)
            cur = {}
            kernels.append(cur)
        if cur is None:
            continue
        if key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                   "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size",
                   "kernarg_segment_size", "wavefront_size"):
            cur[key] = int(val)
        elif key in ("name", "symbol"):
            cur[key] = val.strip()
    names = [k["name"] for k in kernels]
    dem = run("c++filt", *names).splitlines()
    for k, d in zip(kernels, dem):
        d = d.replace("void ", "").replace("tfhe::(anonymous namespace)::", "")
        k["demangled"] = re.sub(r"\(.*$", "", d)
    return kernels


def occupancy(vgprs: int, agprs: int) -> int:
    """waves per SIMD by registers on gfx950: 512 unified VGPR+AGPR per lane, allocation granule 8"""
    total = ((vgprs + 7) // 8) * 8 + ((agprs + 7) // 8) * 8
    return min(8, 512 // max(total, 8))


CLASSES = [
    ("fp64 fma", r"v_fma_f64|v_fmac_f64"),
    ("fp64 mul", r"v_mul_f64"),
    ("fp64 add", r"v_add_f64"),
    ("fp64 rndne", r"v_rndne_f64"),
    ("fp64 cvt/other", r"v_cvt_f64|v_cvt_.*_f64|v_.*_f64"),
    ("int mad64", r"v_mad_u64_u32|v_mad_i64_i32"),
    ("valu mov", r"v_mov_b32|v_mov_b64|v_accvgpr"),
    ("valu cndmask", r"v_cndmask"),
    ("valu other", r"v_"),
    ("lds read", r"ds_read|ds_load"),
    ("lds write", r"ds_write|ds_store"),
    ("lds other", r"ds_"),
    ("global load", r"global_load|buffer_load|flat_load"),
    ("global store", r"global_store|buffer_store|flat_store"),
    ("scratch", r"scratch_"),
    ("s_waitcnt", r"s_waitcnt"),
    ("s_barrier", r"s_barrier"),
    ("branch", r"s_cbranch|s_branch"),
    ("salu/other", r"s_"),
]


def classify(mn: str) -> str:
    for name, pat in CLASSES:
        if re.match(pat, mn):
            return name
    return "other"


def disassemble(co: str, mangled: str):
    """-> list of (address, mnemonic, operands) of one kernel"""
    out = run(f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={mangled}", co)
    insts = []
    for line in out.splitlines():
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m:
            insts.append((int(m.group(3), 16), m.group(1), m.group(2)))
    return insts, out


def loops_of(insts):
    """backward branches -> [(head index, tail index)] (a loop = the address range [target, branch])"""
    addr_index = {a: i for i, (a, _, _) in enumerate(insts)}
    loops = []
    for i, (a, mn, ops) in enumerate(insts):
        if mn.startswith("s_cbranch") or mn == "s_branch":
            m = re.search(r"(-?\d+)\s*$", ops)
            if not m:
                continue
            off = int(m.group(1))
            if off >= 32768:
                off -= 65536
            target = a + 4 + 4 * off
            if target in addr_index and addr_index[target] <= i:
                loops.append((addr_index[target], i))
    return sorted(set(loops), key=lambda t: (t[0], -t[1]))


def histogram(insts):
    h = collections.Counter(classify(mn) for _, mn, _ in insts)
    return h


def fmt_hist(h, indent="    "):
    valu = sum(v for k, v in h.items() if k.startswith(("fp64", "int mad", "valu")))
    fp64 = sum(v for k, v in h.items() if k.startswith("fp64"))
    lines = [f"{indent}instructions {sum(h.values())}: VALU {valu} (fp64 {fp64}, other {valu - fp64})"]
    for name, _ in CLASSES + [("other", "")]:
        if h.get(name):
            lines.append(f"{indent}  {name:<16}{h[name]:>7}")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=LIB)
    ap.add_argument("--kernel", action="append", default=[], help="substring of the demangled kernel name; repeatable")
    ap.add_argument("--asm", default="", help="also write the disassembly of the (last) selected kernel to this file")
    args = ap.parse_args()
    with tempfile.TemporaryDirectory() as tmp:
        co = extract_code_object(args.lib, tmp)
        kernels = kernel_table(co)
        print(f"# {os.path.relpath(args.lib, ROOT)}: {len(kernels)} gfx950 kernels (code object unbundled from .hip_fatbin)")
        print(f"{'kernel':<74}{'VGPR':>5}{'AGPR':>5}{'SGPR':>5}{'spillV':>7}{'scratch B':>10}{'waves/SIMD':>11}")
        for k in sorted(kernels, key=lambda k: k["demangled"]):
            if args.kernel and not any(s in k["demangled"] for s in args.kernel):
                continue
            print(f"{k['demangled']:<74}{k['vgpr_count']:>5}{k['agpr_count']:>5}{k['sgpr_count']:>5}"
                  f"{k.get('vgpr_spill_count', 0):>7}{k['private_segment_fixed_size']:>10}"
                  f"{occupancy(k['vgpr_count'], k['agpr_count']):>11}")
        for sel in args.kernel:
            for k in kernels:
                if sel not in k["demangled"]:
                    continue
                insts, text = disassemble(co, k["name"])
                print(f"\n## {k['demangled']}: {len(insts)} instructions, static mix of the whole kernel")
                print(fmt_hist(histogram(insts)))
                loops = loops_of(insts)
                for depth_sorted in loops:
                    head, tail = depth_sorted
                    inner = [l for l in loops if l != depth_sorted and l[0] >= head and l[1] <= tail]
                    own = [ins for i, ins in enumerate(insts[head:tail + 1], start=head)
                           if not any(a <= i <= b for a, b in inner)]
                    print(f"\n  loop @{insts[head][0]:#x}..{insts[tail][0]:#x} ({tail - head + 1} instructions, "
                          f"{len(inner)} nested loop(s)); body outside nested loops:")
                    print(fmt_hist(histogram(own), indent="      "))
                if args.asm:
                    with open(args.asm, "w") as f:
                        f.write(text)


if __name__ == "__main__":
    sys.exit(main())

"""Builds the gfx950 shared library of the package in-tree with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtfhe_hip.so")
# test instrumentation, never loaded by the package itself: the complex-FFT kernels with the rounding-margin probe
# compiled in (csrc/field_fft.h, tfhe_debug_fft_margin); tests load it through TFHE_HIP_LIB in a child process
PROBE_LIB = os.path.join(HERE, "libtfhe_hip_probe.so")
SOURCES = ["kernels.hip", "capi.cpp", "pool.cpp"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps.append(os.path.join(ROOT, "include", "tfhe_hip.h"))
    deps.append(os.path.abspath(__file__))
    return deps


def _deps():
    return _headers() + [os.path.join(CSRC, s) for s in SOURCES]


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile(lib: str, extra, verbose: bool, tag: str, force: bool = False) -> str:
    """One object per source under _build/<tag>/ (kernels.hip is 2.5 minutes of device code, the host sources are
    seconds: an edit of capi.cpp or pool.cpp does not rebuild the kernels), then one link."""
    # the shipped recipe never builds a library that may compute wrong bits (csrc/dev_switches.h)
    for flag in extra:
        if "TFHE_DEV_BUILD" in flag or "TFHE_PROBE_" in flag:
            raise ValueError(f"build.py does not build dev/probe libraries: {flag} (use tools/dev_build.sh)")
    common = [_hipcc(), "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC",
              "-DTFHE_WAVES_PER_SIMD_FP=2", "-DTFHE_WAVES_PER_SIMD_GL=2",
              "-I", os.path.join(ROOT, "include"), "-I", CSRC] + list(extra)
    objdir = os.path.join(HERE, "_build", tag)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        path = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src + ".o")
        objs.append(obj)
        # the device code sees neither the C ABI header nor the host-side context: edits there leave it alone
        host_only = ("tfhe_hip.h", "context.h")
        newest_header = max(os.path.getmtime(h) for h in _headers()
                            if not (src.endswith(".hip") and os.path.basename(h) in host_only))
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(newest_header, os.path.getmtime(path)):
            continue
        cmd = common + ["-c", path, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> tfhe-research_amd/libtfhe_hip.so"""
    if not force and not is_stale():
        return LIB
    return _compile(LIB, [], verbose, "product", force)


def build_probe(force: bool = False, verbose: bool = False) -> str:
    """The same sources with -DTFHE_FFT_TRACK_ERROR, complex-FFT kernels only -> libtfhe_hip_probe.so"""
    if not force and not is_stale(PROBE_LIB):
        return PROBE_LIB
    return _compile(PROBE_LIB, ["-DTFHE_FFT_TRACK_ERROR", "-DTFHE_DEV_FIELD_FFT_ONLY"], verbose, "probe", force)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_probe(force=True, verbose=True))

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
SOURCES = ["kernels.hip", "capi.cpp"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _deps():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(ROOT, "include", "tfhe_hip.h"))
    return deps


def is_stale(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(d) > t for d in _deps())


def _compile(lib: str, extra, verbose: bool) -> str:
    cmd = [_hipcc(), "-O3", "-std=c++17", "-ffp-contract=off", "--offload-arch=gfx950", "-fPIC", "-shared",
           "-DTFHE_WAVES_PER_SIMD_FP=2", "-DTFHE_WAVES_PER_SIMD_GL=2",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC] + list(extra)
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", lib + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(lib + ".tmp", lib)
    return lib


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 -> tfhe-research_amd/libtfhe_hip.so"""
    if not force and not is_stale():
        return LIB
    return _compile(LIB, [], verbose)


def build_probe(force: bool = False, verbose: bool = False) -> str:
    """The same sources with -DTFHE_FFT_TRACK_ERROR, complex-FFT kernels only -> libtfhe_hip_probe.so"""
    if not force and not is_stale(PROBE_LIB):
        return PROBE_LIB
    return _compile(PROBE_LIB, ["-DTFHE_FFT_TRACK_ERROR", "-DTFHE_DEV_FIELD_FFT_ONLY"], verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_probe(force=True, verbose=True))

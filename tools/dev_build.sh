#!/bin/bash
# Dev build of ONE shape for fast kernel iteration: tools/dev_build.sh <out.so> <logn> <k> [extra -D flags...]
# (the shipped library is built by tfhe-research_amd/build.py; use TFHE_HIP_LIB=<out.so> to load a dev build)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$1"; LOGN="$2"; K="$3"; shift 3
# a WRONG-BITS timing probe (csrc/dev_switches.h) only compiles in a build that declares itself one
DEV=""; for a in "$@"; do case "$a" in -DTFHE_PROBE_*) DEV="-DTFHE_DEV_BUILD";; esac; done
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -fPIC -shared \
  -DTFHE_WAVES_PER_SIMD_FP=2 -DTFHE_WAVES_PER_SIMD_GL=2 -DTFHE_DEV_CFG2_ONLY -DTFHE_DEV_LOGN=$LOGN -DTFHE_DEV_K=$K $DEV "$@" \
  -I "$ROOT/include" -I "$ROOT/tfhe-research_amd/csrc" \
  "$ROOT/tfhe-research_amd/csrc/kernels.hip" "$ROOT/tfhe-research_amd/csrc/capi.cpp" "$ROOT/tfhe-research_amd/csrc/pool.cpp" -o "$OUT"

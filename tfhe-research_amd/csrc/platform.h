// platform.h -- one source for the GPU build (hipcc, gfx950) and for the host SIMT emulator that
// the CPU tests use to run the very same per-lane code (tests/emu).  Not a portability layer for
// other GPUs: the device side is written for 64-wide CDNA4 wavefronts only.
#pragma once
#include <stdint.h>

#include "dev_switches.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define TFHE_HD __host__ __device__ __forceinline__
#define TFHE_D __device__ __forceinline__
#else
#define TFHE_HD inline
#define TFHE_D inline
#endif

namespace tfhe {

typedef uint32_t u32;
typedef uint64_t u64;
typedef int32_t i32;
typedef int64_t i64;

constexpr int kWave = 64;  // CDNA wavefront width

// s_setprio: the issue priority of this wave among the waves of its SIMD (0 = lowest, the state a wave starts in).
// The host emulator runs one lane at a time: nothing to arbitrate.
template <int PRIORITY>
TFHE_HD void wave_priority() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_s_setprio(PRIORITY);
#endif
}

}  // namespace tfhe

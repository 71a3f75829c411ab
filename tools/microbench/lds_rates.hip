// LDS pipe microbenchmark for gfx950 (MI355X): what the transposes of the blind-rotation kernel's transforms can
// cost at best.  (VERDICT r3 next#5a: "measure the LDS pipe first".)
//
// The fp64-fft kernel at N = 1024 moves 512 complex points (16 B) per polynomial through a wave-private 8 KiB LDS buffer
// between its three register passes: 8 ds_write_b128 + 8 ds_read_b128 per lane and transpose, ten transposes per CMUX
// iteration, with the XOR swizzle of wave_ntt.h::ntt_swizzle<9, 1>.  This program issues exactly those address patterns
// (and a linear, trivially conflict-free one as the baseline) from 1, 2 and 3 waves per SIMD on every CU and reports,
// per instruction form:
//     cyc/winstr/CU  shader cycles the CU's LDS path spends per wave-instruction (in-kernel s_memtime, all waves busy)
//     B/clk/CU       bytes moved per shader cycle and CU
// Forms: the shipped b128 stores and loads; the same 16 bytes as two b64 halves (re / im planes: VERDICT's candidate);
// ds_write2_b64 / ds_read2_b64 pairs; and the whole transpose (8 stores, 8 loads, one wait) in both element widths.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_rates.hip -o tools/microbench/lds_rates && tools/microbench/lds_rates
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x)                                                                                     \
  do {                                                                                               \
    hipError_t e_ = (x);                                                                             \
    if (e_ != hipSuccess) {                                                                          \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
      exit(1);                                                                                       \
    }                                                                                                \
  } while (0)

constexpr int ITERS = 512;
constexpr int kWarmupLaunches = 100;
constexpr int kTimedLaunches = 20;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

// wave_ntt.h: 512 elements, 8 per lane, windows [6,9) -> [3,6) -> [0,3); element index of register r of lane `tid`
__device__ __forceinline__ int ntt_index(int tid, int r, int lo) { return ((tid >> lo) << (lo + 3)) | (r << lo) | (tid & ((1 << lo) - 1)); }
__device__ __forceinline__ int ntt_swizzle9(int j) { return j ^ ((j >> 3) & 7) ^ (((j >> 6) & 3) << 3); }

enum Pattern { kLinear = 0, kTop = 1, kMid = 2, kLow = 3 };  // kTop/kMid/kLow: the register windows [6,9) / [3,6) / [0,3)

// byte address of register r of this lane inside the wave's 8 KiB region; elem_bytes 16 (b128) or 8 (one plane of two)
__device__ __forceinline__ unsigned slot_addr(int lane, int r, int pattern, int elem_bytes, unsigned base) {
  const int j = pattern == kLinear ? r * 64 + lane : ntt_swizzle9(ntt_index(lane, r, pattern == kTop ? 6 : pattern == kMid ? 3 : 0));
  return base + (unsigned)j * (unsigned)elem_bytes;
}

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// MODE: 0 ds_write_b128, 1 ds_read_b128, 2 2x ds_write_b64 (planes), 3 2x ds_read_b64 (planes), 4 ds_write2_b64, 5 ds_read2_b64,
//       6 transpose b128 (8 stores from window A, 8 loads in window B, one wait), 7 transpose in b64 planes (16 + 16)
template <int MODE>
__global__ void __launch_bounds__(256) lds_kernel(unsigned long long* cycles, unsigned* sink, int pat_a, int pat_b) {
  const int lane = (int)(threadIdx.x & 63u);
  const unsigned base = (threadIdx.x >> 6) * 8192u;  // one 8 KiB region per wave, as in the kernel
  unsigned wa[8], ra[8], wp[8], rp[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    wa[r] = slot_addr(lane, r, pat_a, 16, base);
    ra[r] = slot_addr(lane, r, pat_b, 16, base);
    wp[r] = slot_addr(lane, r, pat_a, 8, base);  // plane 0 at [0, 4 KiB), plane 1 at +4096 (offset immediate)
    rp[r] = slot_addr(lane, r, pat_b, 8, base);
  }
  u32x4 d[8];
  u32x2 h[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    d[r] = u32x4{(unsigned)lane, (unsigned)r, threadIdx.x, blockIdx.x};
    h[r] = u32x2{(unsigned)lane, (unsigned)r};
  }
  // fill the region once so that loads return defined data
#pragma unroll
  for (int r = 0; r < 8; ++r) *reinterpret_cast<u32x4*>(g_lds + slot_addr(lane, r, kLinear, 16, base)) = d[r];
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITERS; ++it) {
    if (MODE == 0 || MODE == 6) {
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_write_b128 %0, %1" ::"v"(wa[r]), "v"(d[r]) : "memory");
    }
    if (MODE == 1 || MODE == 6) {
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(d[r]) : "v"(ra[r]) : "memory");
    }
    if (MODE == 2 || MODE == 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("ds_write_b64 %0, %1" ::"v"(wp[r]), "v"(h[r]) : "memory");
        asm volatile("ds_write_b64 %0, %1 offset:4096" ::"v"(wp[r]), "v"(h[(r + 1) & 7]) : "memory");
      }
    }
    if (MODE == 3 || MODE == 7) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        asm volatile("ds_read_b64 %0, %1" : "=v"(h[r]) : "v"(rp[r]) : "memory");
        asm volatile("ds_read_b64 %0, %1 offset:4096" : "=v"(h[(r + 1) & 7]) : "v"(rp[r]) : "memory");
      }
    }
    if (MODE == 4) {  // two 8-byte words 4096 bytes apart from one address register (offsets count 8-byte units)
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_write2_b64 %0, %1, %2 offset1:128" ::"v"(wp[r]), "v"(h[r]), "v"(h[(r + 1) & 7]) : "memory");
    }
    if (MODE == 5) {
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_read2_b64 %0, %1 offset1:128" : "=v"(d[r]) : "v"(rp[r]) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  unsigned acc = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) acc ^= d[r].x ^ d[r].y ^ d[r].z ^ d[r].w ^ h[r].x ^ h[r].y;
  sink[blockIdx.x * 256 + threadIdx.x] = acc;
  if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ void k_clock(unsigned long long* out) {
  unsigned long long c0 = __builtin_readcyclecounter(), w0 = wall_clock64();
  unsigned a = threadIdx.x;
  for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, %0\n" : "+v"(a));
  unsigned long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = w1 - w0;
    out[2] = a;
  }
}

typedef void (*kern_t)(unsigned long long*, unsigned*, int, int);
struct Row {
  const char* name;
  kern_t k;
  int winstr_per_iter;  // LDS wave-instructions per loop trip
  int bytes_per_lane;   // bytes one lane moves per loop trip
  int pat_a, pat_b;
};

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* dclk;
  CHECK(hipMalloc(&dclk, 64));
  k_clock<<<1, 64>>>(dclk);
  CHECK(hipDeviceSynchronize());
  unsigned long long hclk[3];
  CHECK(hipMemcpy(hclk, dclk, 24, hipMemcpyDeviceToHost));
  const double counter_hz = (double)hclk[0] / ((double)hclk[1] / 100e6);  // wall clock = 100 MHz
  printf("device %s CUs=%d; cycle counter runs at %.1f MHz (idle chip, single wave)\n", prop.name, cus, counter_hz / 1e6);
  printf("# one 256-thread workgroup = one wave per SIMD, 8 KiB of LDS per wave; w/SIMD workgroups per CU; %d trips per launch,\n"
         "# one s_waitcnt lgkmcnt(0) per trip; cycles from hipEvent time x 2.4 GHz (nominal) and, beside it, the same from the\n"
         "# in-kernel counter scaled by its measured rate\n", ITERS);
  const int max_blocks = cus * 3;
  unsigned long long* cyc;
  unsigned* sink;
  CHECK(hipMalloc(&cyc, sizeof(unsigned long long) * max_blocks * 4));
  CHECK(hipMalloc(&sink, sizeof(unsigned) * max_blocks * 256));
  const std::vector<Row> rows = {
      {"ds_write_b128 linear", lds_kernel<0>, 8, 128, kLinear, kLinear},
      {"ds_write_b128 window[6,9) swz", lds_kernel<0>, 8, 128, kTop, kTop},
      {"ds_write_b128 window[3,6) swz", lds_kernel<0>, 8, 128, kMid, kMid},
      {"ds_write_b128 window[0,3) swz", lds_kernel<0>, 8, 128, kLow, kLow},
      {"ds_read_b128 linear", lds_kernel<1>, 8, 128, kLinear, kLinear},
      {"ds_read_b128 window[6,9) swz", lds_kernel<1>, 8, 128, kTop, kTop},
      {"ds_read_b128 window[3,6) swz", lds_kernel<1>, 8, 128, kMid, kMid},
      {"ds_read_b128 window[0,3) swz", lds_kernel<1>, 8, 128, kLow, kLow},
      {"2x ds_write_b64 planes linear", lds_kernel<2>, 16, 128, kLinear, kLinear},
      {"2x ds_write_b64 planes [6,9)", lds_kernel<2>, 16, 128, kTop, kTop},
      {"2x ds_write_b64 planes [3,6)", lds_kernel<2>, 16, 128, kMid, kMid},
      {"2x ds_read_b64 planes linear", lds_kernel<3>, 16, 128, kLinear, kLinear},
      {"2x ds_read_b64 planes [3,6)", lds_kernel<3>, 16, 128, kMid, kMid},
      {"2x ds_read_b64 planes [0,3)", lds_kernel<3>, 16, 128, kLow, kLow},
      {"ds_write2_b64 planes linear", lds_kernel<4>, 8, 128, kLinear, kLinear},
      {"ds_write2_b64 planes [6,9)", lds_kernel<4>, 8, 128, kTop, kTop},
      {"ds_read2_b64 planes linear", lds_kernel<5>, 8, 128, kLinear, kLinear},
      {"ds_read2_b64 planes [3,6)", lds_kernel<5>, 8, 128, kMid, kMid},
      {"transpose b128 [6,9)->[3,6)", lds_kernel<6>, 16, 256, kTop, kMid},
      {"transpose b128 [3,6)->[0,3)", lds_kernel<6>, 16, 256, kMid, kLow},
      {"transpose b128 [0,3)->[3,6)", lds_kernel<6>, 16, 256, kLow, kMid},
      {"transpose b64 planes [6,9)->[3,6)", lds_kernel<7>, 32, 256, kTop, kMid},
      {"transpose b64 planes [3,6)->[0,3)", lds_kernel<7>, 32, 256, kMid, kLow},
  };
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  printf("%-36s %6s %10s %16s %12s %18s %12s\n", "form", "w/SIMD", "ms", "cyc/winstr/CU", "B/clk/CU", "cyc/winstr/CU(ctr)", "B/clk(ctr)");
  std::vector<unsigned long long> host(max_blocks * 4);
  for (const Row& row : rows) {
    for (int wps : {1, 2, 3}) {
      const int blocks = cus * wps;
      const size_t lds = 4 * 8192;
      for (int i = 0; i < kWarmupLaunches; ++i) hipLaunchKernelGGL(row.k, dim3(blocks), dim3(256), lds, 0, cyc, sink, row.pat_a, row.pat_b);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < kTimedLaunches; ++i) hipLaunchKernelGGL(row.k, dim3(blocks), dim3(256), lds, 0, cyc, sink, row.pat_a, row.pat_b);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= kTimedLaunches;
      CHECK(hipMemcpy(host.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
      double mean = 0;
      for (int i = 0; i < blocks * 4; ++i) mean += (double)host[i];
      mean /= blocks * 4;
      // per CU: wps * 4 waves, each ITERS * winstr wave-instructions
      const double winstr_cu = (double)ITERS * row.winstr_per_iter * wps * 4;
      const double bytes_cu = (double)ITERS * row.bytes_per_lane * 64 * wps * 4;
      const double cyc_evt = ms * 1e-3 * 2.4e9;
      const double cyc_ctr = mean / counter_hz * 2.4e9;  // the counter's ticks as 2.4 GHz cycles (it may not run at the shader clock)
      printf("%-36s %6d %10.4f %16.2f %12.1f %18.2f %12.1f\n", row.name, wps, ms, cyc_evt / winstr_cu, bytes_cu / cyc_evt,
             cyc_ctr / winstr_cu, bytes_cu / cyc_ctr);
    }
  }
  return 0;
}

// fp64 dependent-chain latency vs independent-chain throughput on gfx950, by waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int ITERS = 4096;

// CHAINS independent mulmod-like chains per thread: h=a*w; l=fma(a,w,-h); q=rint(h*pinv); r=fma(-q,p,h)+l
template <int CHAINS>
__global__ void k_mulmod(double* out, double seed) {
  double a[CHAINS];
  const double w = 1234567.0 + threadIdx.x, P = 4398046486529.0, PINV = 1.0 / 4398046486529.0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = seed + c * 17.0 + threadIdx.x;
  for (int it = 0; it < ITERS / CHAINS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) {
      const double h = a[c] * w;
      const double l = __builtin_fma(a[c], w, -h);
      const double q = __builtin_rint(h * PINV);
      a[c] = __builtin_fma(-q, P, h) + l;
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_fma(double* out, double seed) {
  double a[CHAINS];
  const double w = 1.0000001, b = 1e-9;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) a[c] = seed + c;
  for (int it = 0; it < ITERS * 4 / CHAINS; ++it) {
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) a[c] = __builtin_fma(a[c], w, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; ++c) s += a[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
int run(const char* name, K kern, double ops_per_thread, double* out, int cus) {
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int wps : {1, 2, 3, 4, 8}) {
    int blocks = cus * wps;
    kern<<<blocks, 256>>>(out, 1.0); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); kern<<<blocks, 256>>>(out, 2.0); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double ns_per_op_per_simd = ms * 1e6 / (ops_per_thread * wps);
    printf("%-22s waves/SIMD=%d  %8.3f ms  %6.2f cycles/DPop/SIMD (at 2.4GHz)  per-wave op interval %6.2f cycles\n", name, wps, ms,
           ns_per_op_per_simd * 2.4, ms * 1e6 / ops_per_thread * 2.4);
  }
  return 0;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double* out; CHECK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
  // mulmod = 6 DP ops per chain step; ITERS steps per thread in total
  run("mulmod chains=1", k_mulmod<1>, ITERS * 6.0, out, cus);
  run("mulmod chains=2", k_mulmod<2>, ITERS * 6.0, out, cus);
  run("mulmod chains=4", k_mulmod<4>, ITERS * 6.0, out, cus);
  run("mulmod chains=8", k_mulmod<8>, ITERS * 6.0, out, cus);
  run("fma chains=1", k_fma<1>, ITERS * 4.0, out, cus);
  run("fma chains=2", k_fma<2>, ITERS * 4.0, out, cus);
  run("fma chains=4", k_fma<4>, ITERS * 4.0, out, cus);
  run("fma chains=8", k_fma<8>, ITERS * 4.0, out, cus);
  return 0;
}

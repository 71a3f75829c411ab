// Instruction-issue-rate microbenchmark for gfx950 (MI355X).
// Measures sustained wave-instructions / cycle / SIMD for the VALU ops the exact-NTT
// design space depends on (32-bit integer multiply, 64-bit mad, carry chains, fp64, fp32).
// Output: one line per (instruction, waves-per-SIMD) with cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;
constexpr int kWarmupLaunches = 200;  // >= 15 ms of work before anything is timed
constexpr int kTimedLaunches = 20;
constexpr int UNROLL = 16;   // instructions per loop body (8 independent chains x 2)

#define R8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// 32-bit ops: 8 independent accumulators a0..a7, operands b, c
#define DEF_KERNEL32(NAME, ASMSTR)                                                     \
__global__ void NAME(unsigned* out, unsigned seed) {                                   \
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3; \
  unsigned a4 = a0 * 11u + 4, a5 = a0 * 13u + 5, a6 = a0 * 17u + 6, a7 = a0 * 19u + 7;   \
  unsigned b = seed * 2654435761u + 12345u + threadIdx.x, c = seed ^ 0x9e3779b9u;      \
  for (int it = 0; it < ITERS; ++it) {                                                 \
    asm volatile(                                                                      \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
      : "v"(b), "v"(c) : "vcc");                                                       \
  }                                                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
}

#define I_ADD(A)      "v_add_u32 " A ", " A ", %8\n"
#define I_ADDCO(A)    "v_add_co_u32 " A ", vcc, " A ", %8\n"
#define I_ADDC(A)     "v_addc_co_u32 " A ", vcc, " A ", %8, vcc\n"
#define I_MULLO(A)    "v_mul_lo_u32 " A ", " A ", %8\n"
#define I_MULHI(A)    "v_mul_hi_u32 " A ", " A ", %8\n"
#define I_MUL24(A)    "v_mul_u32_u24 " A ", " A ", %8\n"
#define I_MAD24(A)    "v_mad_u32_u24 " A ", " A ", %8, %9\n"
#define I_MULHI24(A)  "v_mul_hi_u32_u24 " A ", " A ", %8\n"
#define I_FMA32(A)    "v_fma_f32 " A ", " A ", %8, %9\n"
#define I_CNDMASK(A)  "v_cndmask_b32 " A ", " A ", %8, vcc\n"
#define I_ALIGNBIT(A) "v_alignbit_b32 " A ", " A ", %8, 7\n"
#define I_BFE(A)      "v_bfe_u32 " A ", " A ", 3, 9\n"
#define I_LSHLADD(A)  "v_lshl_add_u32 " A ", " A ", 3, %8\n"
#define I_ADD3(A)     "v_add3_u32 " A ", " A ", %8, %9\n"
#define I_XAD(A)      "v_xad_u32 " A ", " A ", %8, %9\n"
#define I_DPP(A)      "v_mov_b32_dpp " A ", " A " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_SUBB(A)     "v_subb_co_u32 " A ", vcc, " A ", %8, vcc\n"

DEF_KERNEL32(k_add, I_ADD)
DEF_KERNEL32(k_addco, I_ADDCO)
DEF_KERNEL32(k_addc, I_ADDC)
DEF_KERNEL32(k_subb, I_SUBB)
DEF_KERNEL32(k_mullo, I_MULLO)
DEF_KERNEL32(k_mulhi, I_MULHI)
DEF_KERNEL32(k_mul24, I_MUL24)
DEF_KERNEL32(k_mad24, I_MAD24)
DEF_KERNEL32(k_mulhi24, I_MULHI24)
DEF_KERNEL32(k_fma32, I_FMA32)
DEF_KERNEL32(k_cndmask, I_CNDMASK)
// v_cndmask_b32 again with its lane mask defined inside the block: k_cndmask above reads a VCC that
// nothing in the kernel ever wrote, and round 1 measured 22.8 cycles for it at every occupancy -- five
// times the other VALU rows.  k_cndmask_vccset writes VCC once per 16 selects (SALU), k_cndmask_sgpr
// takes the mask from an ordinary SGPR pair (VOP3 form, what the compiler emits for a select on a
// wave-uniform condition), k_cndmask_vcmp produces VCC with a VALU compare before every select (what
// a per-lane conditional correction such as a modular reduction costs: 2 instructions).
#define DEF_KERNEL_SEL(NAME, PRE, ASMSTR, OPS)                                         \
__global__ void NAME(unsigned* out, unsigned seed) {                                   \
  unsigned a0 = seed + threadIdx.x, a1 = a0 * 3u + 1, a2 = a0 * 5u + 2, a3 = a0 * 7u + 3; \
  unsigned a4 = a0 * 11u + 4, a5 = a0 * 13u + 5, a6 = a0 * 17u + 6, a7 = a0 * 19u + 7;   \
  unsigned b = seed * 2654435761u + 12345u + threadIdx.x, c = seed ^ 0x9e3779b9u;      \
  unsigned long long mask = 0x5555333300ff0f0full * (seed | 1u);                       \
  for (int it = 0; it < ITERS; ++it) {                                                 \
    asm volatile(                                                                      \
      PRE                                                                              \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
      : "v"(b), "v"(c), "s"(mask) : "vcc");                                            \
  }                                                                                    \
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;  \
}
#define I_CNDMASK_SGPR(A) "v_cndmask_b32_e64 " A ", " A ", %8, %10\n"
#define I_CMP_CNDMASK(A)  "v_cmp_lt_u32 vcc, " A ", %9\nv_cndmask_b32 " A ", " A ", %8, vcc\n"
DEF_KERNEL_SEL(k_cndmask_vccset, "s_mov_b64 vcc, %10\n", I_CNDMASK, 16)
DEF_KERNEL_SEL(k_cndmask_sgpr, "", I_CNDMASK_SGPR, 16)
DEF_KERNEL_SEL(k_cndmask_vcmp, "", I_CMP_CNDMASK, 32)
DEF_KERNEL32(k_alignbit, I_ALIGNBIT)
DEF_KERNEL32(k_bfe, I_BFE)
DEF_KERNEL32(k_lshladd, I_LSHLADD)
DEF_KERNEL32(k_add3, I_ADD3)
DEF_KERNEL32(k_dpp, I_DPP)

// 64-bit ops: accumulators are register pairs
#define DEF_KERNEL64(NAME, ASMSTR, T, INIT)                                            \
__global__ void NAME(unsigned* out, unsigned seed) {                                   \
  T a0 = INIT(1), a1 = INIT(2), a2 = INIT(3), a3 = INIT(4);                             \
  T a4 = INIT(5), a5 = INIT(6), a6 = INIT(7), a7 = INIT(8);                             \
  T b = INIT(9), c = INIT(10);                                                         \
  unsigned m = seed * 2654435761u + threadIdx.x;                                       \
  for (int it = 0; it < ITERS; ++it) {                                                 \
    asm volatile(                                                                      \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      ASMSTR("%0") ASMSTR("%1") ASMSTR("%2") ASMSTR("%3")                              \
      ASMSTR("%4") ASMSTR("%5") ASMSTR("%6") ASMSTR("%7")                              \
      : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
      : "v"(b), "v"(c), "v"(m) : "vcc");                                               \
  }                                                                                    \
  T s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                         \
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(long long)s;                 \
}
#define INIT_U64(k) ((unsigned long long)(seed + threadIdx.x) * (0x9e3779b97f4a7c15ull + k) + k)
#define INIT_F64(k) (1.0 + 1e-9 * (double)((seed + threadIdx.x) % 97 + k))

#define I_MAD64(A)   "v_mad_u64_u32 " A ", vcc, %10, %10, " A "\n"
#define I_MAD64S(A)  "v_mad_u64_u32 " A ", vcc, %10, %10, " A "\n"
#define I_LSHL64(A)  "v_lshlrev_b64 " A ", 3, " A "\n"
#define I_FMA64(A)   "v_fma_f64 " A ", " A ", %8, %9\n"
#define I_MUL64(A)   "v_mul_f64 " A ", " A ", %8\n"
#define I_ADD64(A)   "v_add_f64 " A ", " A ", %8\n"
#define I_RND64(A)   "v_rndne_f64 " A ", " A "\n"
#define I_PKFMA32(A) "v_pk_fma_f32 " A ", " A ", %8, %9\n"
#define I_PKADD32(A) "v_pk_add_f32 " A ", " A ", %8\n"
#define I_PKMUL32(A) "v_pk_mul_f32 " A ", " A ", %8\n"

DEF_KERNEL64(k_mad64, I_MAD64, unsigned long long, INIT_U64)
DEF_KERNEL64(k_lshl64, I_LSHL64, unsigned long long, INIT_U64)
DEF_KERNEL64(k_fma64, I_FMA64, double, INIT_F64)
DEF_KERNEL64(k_mul64, I_MUL64, double, INIT_F64)
DEF_KERNEL64(k_add64, I_ADD64, double, INIT_F64)
DEF_KERNEL64(k_rnd64, I_RND64, double, INIT_F64)
DEF_KERNEL64(k_pkfma32, I_PKFMA32, double, INIT_F64)
DEF_KERNEL64(k_pkadd32, I_PKADD32, double, INIT_F64)
DEF_KERNEL64(k_pkmul32, I_PKMUL32, double, INIT_F64)

// dependent-chain variant for mad64: one accumulator, 16 dependent mads per body
__global__ void k_mad64_dep(unsigned* out, unsigned seed) {
  unsigned long long a0 = INIT_U64(1);
  unsigned m = seed * 2654435761u + threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
    asm volatile(
      I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0")
      I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0")
      I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0")
      I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0") I_MAD64S("%0")
      : "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0)
      : "v"(m), "v"(m), "v"(m) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)a0;
}
__global__ void k_add_dep(unsigned* out, unsigned seed) {
  unsigned a0 = seed + threadIdx.x, b = seed * 7u + 1;
  for (int it = 0; it < ITERS; ++it) {
    asm volatile(
      I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0")
      I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0") I_ADD("%0")
      : "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0), "+v"(a0)
      : "v"(b), "v"(b) : "vcc");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0;
}

__global__ void k_clock(unsigned long long* out) {
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  unsigned a = threadIdx.x;
  for (int i = 0; i < 200000; ++i) asm volatile("v_add_u32 %0, %0, %0\n" : "+v"(a));
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = a; }
}

typedef void (*kern_t)(unsigned*, unsigned);
struct Entry { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  printf("device %s CUs=%d clockRate=%d kHz lds/block=%zu regs/block=%d\n", prop.name, cus,
         prop.clockRate, prop.sharedMemPerBlock, prop.regsPerBlock);
  unsigned long long* dclk; CHECK(hipMalloc(&dclk, 64));
  k_clock<<<1, 64>>>(dclk); CHECK(hipDeviceSynchronize());
  unsigned long long hclk[3]; CHECK(hipMemcpy(hclk, dclk, 24, hipMemcpyDeviceToHost));
  double ghz = (double)hclk[0] / ((double)hclk[1] / 100e6) / 1e9;   // wall clock = 100 MHz
  printf("shader clock (single wave, idle chip): %.3f GHz\n", ghz);

  unsigned* out; CHECK(hipMalloc(&out, sizeof(unsigned) * cus * 8 * 256 * 4));
  std::vector<Entry> es = {
    {"v_add_u32", k_add}, {"v_add_u32(dep chain)", k_add_dep}, {"v_add_co_u32", k_addco}, {"v_addc_co_u32", k_addc},
    {"v_subb_co_u32", k_subb},
    {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd}, {"v_bfe_u32", k_bfe}, {"v_alignbit_b32", k_alignbit},
    {"v_cndmask_b32(vcc never set)", k_cndmask}, {"v_cndmask_b32(vcc by s_mov)", k_cndmask_vccset},
    {"v_cndmask_b32_e64(sgpr mask)", k_cndmask_sgpr}, {"v_cmp+v_cndmask (x2 instr)", k_cndmask_vcmp},
    {"v_mov_b32_dpp", k_dpp},
    {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_mad_u64_u32", k_mad64},
    {"v_mad_u64_u32(dep chain)", k_mad64_dep},
    {"v_mul_u32_u24", k_mul24}, {"v_mad_u32_u24", k_mad24}, {"v_mul_hi_u32_u24", k_mulhi24},
    {"v_lshlrev_b64", k_lshl64},
    {"v_fma_f32", k_fma32}, {"v_pk_fma_f32", k_pkfma32}, {"v_pk_add_f32", k_pkadd32}, {"v_pk_mul_f32", k_pkmul32},
    {"v_fma_f64", k_fma64}, {"v_mul_f64", k_mul64}, {"v_add_f64", k_add64}, {"v_rndne_f64", k_rnd64},
  };
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("%-30s %6s %12s %14s %16s\n", "instr", "w/SIMD", "ms", "ns/winstr/SIMD", "cyc@2.4GHz");
  for (auto& e : es) {
    for (int wps : {1, 2, 4, 8}) {
      // one block of 256 threads = 1 wave per SIMD on a CU; wps blocks per CU
      int blocks = cus * wps;
      // warm-up: one launch is 0.07-2.5 ms; the chip only reaches its sustained clocks after tens of
      // milliseconds of work (round 1 timed the second launch and read every rate ~15 % too slow)
      for (int i = 0; i < kWarmupLaunches; ++i) e.k<<<blocks, 256>>>(out, 1);
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < kTimedLaunches; ++i) e.k<<<blocks, 256>>>(out, 2);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      ms /= kTimedLaunches;
      const bool two = std::string(e.name).find("x2 instr") != std::string::npos;
      double winstr_per_simd = (double)ITERS * UNROLL * wps * (two ? 2 : 1);   // wave-instructions issued per SIMD
      double ns = ms * 1e6 / winstr_per_simd;
      printf("%-30s %6d %12.4f %14.3f %16.2f\n", e.name, wps, ms, ns, ns * 2.4);
    }
  }
  return 0;
}

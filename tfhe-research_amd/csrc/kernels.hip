// kernels.hip -- gfx950 (CDNA4 / MI355X) kernels of the TFHE bootstrapping hot path.
//
// Execution shape: one workgroup = one TEAM of K+1 groups of G wavefronts = one LWE sample; group c
// owns GLWE polynomial c and output column c (pbs_wave.h).  G = 1 up to N = 1024 (transforms are
// wave-local, no barrier inside), G = 4 at N = 2048 (one cross-wave transpose per transform).  The
// team meets at a workgroup barrier once or twice per gadget level (two or one LDS exchange
// buffers per group) to hand the digit spectra around.
#include <atomic>
#include <cstdlib>

#include "launch.h"

namespace tfhe {
namespace {

// waves per polynomial: one, except N = 2048, where four waves hold 8 elements per array each.
// (With two waves -- 16 elements, 256 VGPRs -- a k = 2 team is 6 waves on a CU that then has room
// for 8: SIMDs end up with 2, 2, 1, 1 waves and the team runs at the pace of the shared ones.  Four
// waves per polynomial fit 170 VGPRs, so the 12-wave team sits 3, 3, 3, 3.)
#ifndef TFHE_GROUP_N2048
#define TFHE_GROUP_N2048 4
#endif
#ifndef TFHE_GROUP_N2048_FFT
#define TFHE_GROUP_N2048_FFT 4
#endif
// (N = 1024 over two waves was measured too: 97.9 ms against 64.0 ms per cfg2 batch -- a fourth
// register pass and cross-wave barriers cost more than the third wave per SIMD gives.)
// (experiment flag: the complex transform at N = 1024 over two waves per polynomial -- 4 elements per lane, registers
// for two samples per team; profiles/r03_kernel_ab.txt)
#ifndef TFHE_GROUP_N1024_FFT
#define TFHE_GROUP_N1024_FFT 1
#endif
template <class F, int LOGN>
struct GroupOf {
  // the complex transform has N/2 elements of 16 bytes: two waves per polynomial at N = 2048 hold 8 of
  // them per lane and array (the register footprint of one wave at N = 1024)
  static constexpr int value = (LOGN >= 11) ? (F::kLogShrink ? TFHE_GROUP_N2048_FFT : TFHE_GROUP_N2048)
                               : (LOGN == 10 && F::kLogShrink) ? TFHE_GROUP_N1024_FFT : 1;
};

// Shapes a transform policy is instantiated for.  The complex transform (field_fft.h) holds two
// coefficients per element, so its N/2-point transform has 8 elements per lane at N = 1024 with one wave
// per polynomial and 4 at N = 512 (four register passes of two bits); N = 2048 is bit-exact both over two waves per
// polynomial (8 elements per lane, one 6-wave team per CU: 137 ms per 1024 cfg5 bootstraps) and over four (4 elements
// per lane, five register passes, 12 waves: 79.2 ms) but does not beat the 42-bit field's 77.2 ms -- a 12-wave team
// is bound by its barriers, not its arithmetic (profiles/r02_kernel_ab.txt) -- so it is not instantiated.
// the complex transform at N = 2048: over four waves per polynomial (4 elements per lane, five register passes) with
// TWO samples per team -- 57.4 ms per 1024 cfg5 bootstraps against 77.5 ms for the 42-bit field and 80.1 ms with one
// sample per team (profiles/r03_kernel_ab.txt); 0: not instantiated (round 2's state)
#ifndef TFHE_FFT_N2048
#define TFHE_FFT_N2048 1
#endif
template <class F, int LOGN>
constexpr bool field_shape_ok() {
  return F::kLogShrink == 0 || LOGN == 10 || LOGN == 9 || (TFHE_FFT_N2048 && LOGN == 11);
}

// bytes of the twiddle table of a field at ring degree 2^LOGN
template <class F, int LOGN>
constexpr size_t twiddle_bytes() {
  return (size_t)ntt_twiddle_words(1 << (LOGN - F::kLogShrink)) * sizeof(typename F::elem);
}
// entries of it the working copy in LDS needs: the fused-stage constants behind psi_rev[N) are read through the
// uniform pointer only, and fields without fused stages do not have them at all -- 288 bytes that decide whether
// a fourth team fits a CU at N = 512, k = 2 and whether two samples fit the one team of N = 2048
template <class F, int LOGN>
constexpr int staged_twiddle_words() {
  return F::kFuseFirstTwo ? ntt_twiddle_words(1 << (LOGN - F::kLogShrink)) : (1 << (LOGN - F::kLogShrink));
}

template <class Elem, int G, int EXB = 1>
struct DeviceWave {
  unsigned char* team_base_;  // LDS of group 0 of this team (its exchange buffer in use)
  Elem* scratch_;             // my group's exchange / transpose buffer in use
  u32* acc_;
  const Elem* tw_;
  const Elem* twg_;           // the natural-order table in global memory (kernel argument: uniform)
  int group_;                 // polynomial / output column of my group
  unsigned group_stride_;     // bytes of LDS per group
  unsigned buffer_bytes_;     // bytes of one exchange buffer (EXB of them per group, back to back)
  unsigned acc_words_;        // u32 words of one accumulator polynomial (one per sample of the team, back to back)
  // number of exchange buffers per group and a copy of this context that works in buffer i
  // (pbs_wave.h::external_product_team); buffer 0 is the one selected at construction
  __device__ __forceinline__ int exchange_buffers() const { return EXB; }
  __device__ __forceinline__ DeviceWave with_exchange_buffer(int i) const {
    DeviceWave w = *this;
    if (EXB > 1) {
      w.team_base_ += (size_t)i * buffer_bytes_;
      w.scratch_ = reinterpret_cast<Elem*>(reinterpret_cast<unsigned char*>(scratch_) + (size_t)i * buffer_bytes_);
    }
    return w;
  }
  // thread index inside my polynomial's group of G waves
  __device__ __forceinline__ int tid() const { return (int)(threadIdx.x & (64u * G - 1u)); }
  __device__ __forceinline__ int group() const { return group_; }
  // Orders LDS stores of my polynomial's group before its later LDS loads.  G == 1: all 64 lanes run
  // in lockstep and the LDS pipeline is in-order per wave, so only the compiler has to be fenced.
  // G > 1: the group spans waves, so this is the workgroup barrier (every wave of the workgroup runs
  // the same transforms, hence the same number of barriers).
  __device__ __forceinline__ void poly_sync() const {
    if (G == 1) {
      wave_sync();
    } else {
      __syncthreads();
    }
  }
  // Orders the LDS stores of this wave's lanes before their later LDS loads (wave-local exchanges)
  __device__ __forceinline__ void wave_sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // workgroup barrier: the team of K+1 groups that shares one sample IS the workgroup
  // (timing experiments with the barriers compiled out: cfg2 -1 %, cfg3 20 % SLOWER -- there they cost
  // next to nothing and keep the team's key accesses together; cfg5, one 12-wave team per CU and 13
  // workgroup barriers per product: 78.6 -> 67.2 ms, i.e. 14.5 % of that kernel is barrier wait.  A
  // barrier scoped to the 4 waves of one polynomial group (LDS arrival counter + bounded spin) for the
  // 8 barriers that only order a group's cross-wave transposes was built and measured: bit-exact, same
  // time (78.88 vs 78.95 ms) -- the wait is the drift between waves that share SIMDs three at a time,
  // whatever the scope.  profiles/r02_kernel_ab.txt)
  // (TFHE_PROBE_NO_TEAM_SYNC: dev_switches.h -- a WRONG-BITS timing probe, TFHE_DEV_BUILD only)
  __device__ __forceinline__ void team_sync() const {
    if (!TFHE_PROBE_NO_TEAM_SYNC) __syncthreads();
  }
  __device__ __forceinline__ Elem* scratch() const { return scratch_; }
  __device__ __forceinline__ const Elem* scratch_of(int s) const {
    return reinterpret_cast<const Elem*>(team_base_ + (size_t)s * group_stride_);
  }
  __device__ __forceinline__ u32* acc(int s = 0) const { return acc_ + (size_t)s * acc_words_; }
  __device__ __forceinline__ const Elem* twiddles() const { return tw_; }
  __device__ __forceinline__ const Elem* twiddles_uniform() const { return twg_; }
  __device__ __forceinline__ u32 uniform(u32 v) const { return __builtin_amdgcn_readfirstlane(v); }
  // *p += v on an LDS word this lane owns: one ds_add_u32 (no return value) instead of a read, an add and a write
#ifndef TFHE_LDS_ADD
#define TFHE_LDS_ADD 1
#endif
  __device__ __forceinline__ void lds_add(u32* p, u32 v) const {
    if (TFHE_LDS_ADD) (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else *p += v;
  }
  // compiler-only barrier: memory operations are not moved across it
  __device__ __forceinline__ void compiler_fence() const { asm volatile("" ::: "memory"); }
};

#ifndef TFHE_WAVES_PER_SIMD_GL
#define TFHE_WAVES_PER_SIMD_GL 2
#endif
#ifndef TFHE_WAVES_PER_SIMD_FP
#define TFHE_WAVES_PER_SIMD_FP 2
#endif
#ifndef TFHE_WAVES_PER_SIMD_E8  // shapes with 8 ring coefficients per lane and array
#define TFHE_WAVES_PER_SIMD_E8 3
#endif

// One workgroup = one team = K+1 polynomial groups of G waves = one LWE sample.
// LDS (dynamic, 16-B aligned base, no static LDS): [ twiddles (N+2) x 8 B ][ group c: EXB transpose/
// exchange buffers of N x 8 B | accumulator polynomial N x 4 B ] for c = 0..K
// Exchange buffers per group: two (one team barrier per gadget level instead of two) at N = 2048,
// where one 12-wave team per CU leaves LDS to spare and the barriers of a 768-thread workgroup are
// dear (+5 % at cfg5).  One at N = 1024 (4 teams per CU would need 192 KiB with two) and at N = 512
// (two fit, 139 KiB, but measured no gain at cfg3: 114.6 vs 116.1 ms per 4096 gates).
#ifndef TFHE_EXB_N512
#define TFHE_EXB_N512 1
#endif
template <int LOGN>
struct ExchangeBuffersOf {
  static constexpr int value = (LOGN >= 11) ? 2 : (LOGN == 9) ? TFHE_EXB_N512 : 1;
};

// (Two and four teams per workgroup -- one barrier sequence, one twiddle table, the second wave that asks for
// a key tile finding it in the CU's vector L1 -- were measured for the complex transform, whose kernel draws
// 13 TB/s of key from L2: 41.5 and 46.4 ms against 37.1 ms with one team per workgroup.  Teams that are free
// to drift fill each other's stalls; coupling them costs more than the L1 hits give.)
// Samples per team in the blind rotation (pbs_wave.h::external_product_team_multi): two where the registers allow it
// (the complex transform with 4 elements per lane) AND the kernel waits on its key stream or its barriers -- the
// twelve-wave team of N = 2048 and the many digit rows of N = 512, k = 2.  Measured per shape, profiles/r03_kernel_ab.txt.
#ifndef TFHE_NS_N2048
#define TFHE_NS_N2048 2
#endif
#ifndef TFHE_NS_N512_K2
#define TFHE_NS_N512_K2 2
#endif
#ifndef TFHE_NS_N512_K1
#define TFHE_NS_N512_K1 1
#endif
#ifndef TFHE_NS_N1024
#define TFHE_NS_N1024 1
#endif
template <class F, int LOGN, int K>
struct SamplesPerTeam {
  static constexpr int value = !F::kLogShrink ? 1 : LOGN == 11 ? TFHE_NS_N2048 : LOGN == 9 ? (K == 2 ? TFHE_NS_N512_K2 : TFHE_NS_N512_K1)
                               : LOGN == 10 ? TFHE_NS_N1024 : 1;
};

template <class F, int LOGN, int K, int NS_ = SamplesPerTeam<F, LOGN, K>::value>
struct TeamCfg {
  static constexpr int N = 1 << LOGN;
  static constexpr int G = GroupOf<F, LOGN>::value;
  static constexpr int S = NS_;  // samples per team: each has an exchange buffer and an accumulator per group
  static constexpr int EXB = ExchangeBuffersOf<LOGN>::value > S ? ExchangeBuffersOf<LOGN>::value : S;
  static constexpr int kWaves = (K + 1) * G;
  static constexpr int kThreads = kWaves * 64;
  static constexpr unsigned kGroupLds = (unsigned)N * 8u * EXB + (unsigned)N * 4u * S;
  // the working copy of the table: (N + 18) 8-byte elements, or N/2 16-byte ones
  static constexpr size_t kTwBytes = (size_t)staged_twiddle_words<F, LOGN>() * sizeof(typename F::elem);  // multiple of 16
  static constexpr size_t kLds = kTwBytes + (size_t)(K + 1) * kGroupLds;
  static constexpr int kMinWavesGl = (NttShape<LOGN, G>::kE == 8) ? TFHE_WAVES_PER_SIMD_E8 : TFHE_WAVES_PER_SIMD_GL;
  static constexpr int kMinWavesFp = (NttShape<LOGN, G>::kE == 8) ? TFHE_WAVES_PER_SIMD_E8 : TFHE_WAVES_PER_SIMD_FP;
};

template <class F, int LOGN, int K, int NS = SamplesPerTeam<F, LOGN, K>::value>
__device__ __forceinline__ DeviceWave<typename F::elem, GroupOf<F, LOGN>::value, TeamCfg<F, LOGN, K, NS>::EXB>
make_wave(unsigned char* smem, const typename F::elem* tw_global) {
  typedef typename F::elem elem;
  using C = TeamCfg<F, LOGN, K, NS>;
  static_assert(C::kTwBytes % 16 == 0, "twiddle table");
  elem* tw = reinterpret_cast<elem*>(smem);
  ntt_stage_twiddles<LOGN - F::kLogShrink, C::G, elem, staged_twiddle_words<F, LOGN>()>(tw, tw_global, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  DeviceWave<elem, C::G, C::EXB> w;
  w.group_ = (int)(threadIdx.x / (64u * C::G));
  w.group_stride_ = C::kGroupLds;
  w.buffer_bytes_ = (unsigned)C::N * 8u;
  w.acc_words_ = (unsigned)C::N;
  w.team_base_ = smem + C::kTwBytes;
  unsigned char* base = w.team_base_ + (size_t)w.group_ * C::kGroupLds;
  w.tw_ = tw;
  w.twg_ = tw_global;
  w.scratch_ = reinterpret_cast<elem*>(base);
  w.acc_ = reinterpret_cast<u32*>(base + (size_t)C::N * 8 * C::EXB);
  return w;
}

extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];

// ------------------------------------------------------------------------------ bsk_prepare
// one polynomial per group of G waves; kPolysPerBlock groups per workgroup
template <class F, int LOGN>
__global__ void __launch_bounds__(256) bsk_prepare_kernel(const typename F::elem* __restrict__ tw,
                                                         const u32* __restrict__ polys,
                                                         size_t poly_count,
                                                         typename F::elem* __restrict__ spectra,
                                                         typename F::elem n_inv, int layout_e) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int G = GroupOf<F, LOGN>::value;
  elem* twl = reinterpret_cast<elem*>(g_smem);
  ntt_stage_twiddles<LOGN - F::kLogShrink, G>(twl, tw, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  const int group = (int)(threadIdx.x / (64u * G));
  const int groups = (int)(blockDim.x / (64u * G));
  // For G > 1 the transform contains workgroup barriers, so every group of the block has to run it:
  // a group past the end of the list redoes the last polynomial (same values to the same addresses).
  size_t poly = (size_t)blockIdx.x * groups + group;
  if (poly >= poly_count) {
    if (G == 1) return;
    poly = poly_count - 1;
  }
  DeviceWave<elem, G> w;
  w.group_ = 0;
  w.group_stride_ = 0;
  w.buffer_bytes_ = 0;
  w.acc_words_ = 0;
  w.team_base_ = nullptr;
  w.tw_ = twl;
  w.twg_ = tw;
  w.scratch_ = reinterpret_cast<elem*>(g_smem + twiddle_bytes<F, LOGN>() + (size_t)group * N * 8);
  w.acc_ = nullptr;
  // the key's layout (pbs_wave.h::key_layout_e): the pair kernel's at N = 512, k = 1 in the complex transform
  if constexpr (key_layout_e<F, LOGN, 1>() != 0) {
    if (layout_e == key_layout_e<F, LOGN, 1>()) {
      bsk_prepare_wave<F, LOGN, G, key_layout_e<F, LOGN, 1>()>(w, polys + poly * N, spectra + poly * (N >> F::kLogShrink) * F::kParts, n_inv);
      return;
    }
  }
  bsk_prepare_wave<F, LOGN, G>(w, polys + poly * N, spectra + poly * (N >> F::kLogShrink) * F::kParts, n_inv);
}

// ------------------------------------------------------------------------------ blind rotation
template <class F, int LOGN, int K>
__global__ void __launch_bounds__((TeamCfg<F, LOGN, K>::kThreads),
                                  (F::kId == FpField::kId || F::kId == Fp49Field::kId || F::kId == FftField::kId
                                       ? TeamCfg<F, LOGN, K>::kMinWavesFp
                                       : TeamCfg<F, LOGN, K>::kMinWavesGl))
blind_rotate_kernel(PbsParams P, const typename F::elem* __restrict__ tw,
                    const u32* __restrict__ lwe_in, size_t batch, const u32* __restrict__ tv,
                    size_t tv_stride, const typename F::elem* __restrict__ bsk,
                    u32* glwe_out /* may be glwe_state (the caller's output parks the accumulators): no restrict */,
                    u32* __restrict__ lwe_extracted,
                    u32 i_begin, u32 i_end, u32* glwe_state /* [batch][K+1][N]: accumulators between segments */) {
  using C = TeamCfg<F, LOGN, K>;
  constexpr int N = C::N;
  constexpr int G = C::G;
  constexpr int NS = C::S;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  auto w = make_wave<F, LOGN, K>(g_smem, tw);
  // grid = ceil(batch / NS) teams of NS samples; every wave runs every barrier.  An odd batch leaves the last team
  // one sample short: it redoes the last sample in the free slot (same inputs, no output)
  size_t sample[NS];
  const u32* lwes[NS];
  const u32* tvs[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const size_t idx = (size_t)blockIdx.x * NS + s;
    sample[s] = idx < batch ? idx : batch - 1;
    lwes[s] = lwe_in + sample[s] * (P.n + 1);
    tvs[s] = tv + sample[s] * tv_stride;
  }

  // (TFHE_STAGGER: experiment -- teams that are likely to share a CU start a fraction of a level apart, so that one team's
  // multiply-accumulate (key fill) overlaps another's transforms (VALU); units of 64 cycles per phase step)
#ifndef TFHE_STAGGER
#define TFHE_STAGGER 0
#endif
#ifndef TFHE_STAGGER_SHIFT
#define TFHE_STAGGER_SHIFT 8
#endif
  if (TFHE_STAGGER > 0) {
    const unsigned phase = (blockIdx.x >> TFHE_STAGGER_SHIFT) & 3u;
    for (unsigned i = 0; i < phase; ++i) __builtin_amdgcn_s_sleep(TFHE_STAGGER);
  }
  const u32* resume[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) resume[s] = i_begin > 0 ? glwe_state + sample[s] * (size_t)(K + 1) * N : nullptr;  // (null state: one launch)
  blind_rotate_team_multi<F, LOGN, K, G, NS>(w, P, lwes, tvs, bsk, i_begin, i_end, resume);

  const int tid = w.tid();
  if (i_end < P.n) {  // not the last segment: park the accumulators
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      if ((size_t)blockIdx.x * NS + s >= batch) break;
      u32* dst = glwe_state + (sample[s] * (size_t)(K + 1) + w.group()) * N;
#pragma unroll
      for (int r = 0; r < E; ++r) dst[r * T + tid] = w.acc(s)[r * T + tid];
    }
    return;
  }
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    if ((size_t)blockIdx.x * NS + s >= batch) break;  // the duplicate of an odd batch's last sample
    if (glwe_out) {
      u32* dst = glwe_out + (sample[s] * (size_t)(K + 1) + w.group()) * N;
#pragma unroll
      for (int r = 0; r < E; ++r) dst[r * T + tid] = w.acc(s)[r * T + tid];
    }
    if (lwe_extracted) sample_extract_team<LOGN, K, G>(w, lwe_extracted + sample[s] * ((size_t)K * N + 1), s);
  }
}

// ------------------------------------------------------------------------------ blind rotation, pair kernel
// ONE wavefront per sample at k = 1, N = 512 in the complex transform (pbs_wave.h::blind_rotate_pair): polynomial c in lanes
// 32 c .. 32 c + 31, 8 elements per lane, no workgroup barrier.  The shape's throughput kernel (the launch plan -- key
// slices, two streams -- is the team kernel's: launch_blind_rotate); same arguments as blind_rotate_kernel.
// LDS: [ twiddles ][ half 0: buffer N x 4 B... | half 1 ] = 4 KiB + 2 x 4 KiB buffers + 2 x 2 KiB accumulator polynomials.
template <class F, int LOGN, int K>
constexpr bool pair_shape() {
  return F::kLogShrink == 1 && F::kParts == 2 && LOGN == 9 && K == 1;
}
template <class Elem>
struct PairWave {
  unsigned char* buffers_;  // half 0's transpose / exchange buffer; half 1's follows
  u32* accs_;               // accumulator polynomial 0; polynomial 1 follows
  const Elem* tw_;
  const Elem* twg_;
  unsigned buffer_bytes_, acc_words_;
  __device__ __forceinline__ int tid() const { return (int)(threadIdx.x & 31u); }
  __device__ __forceinline__ int group() const { return (int)((threadIdx.x >> 5) & 1u); }
  __device__ __forceinline__ void wave_sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void poly_sync() const { wave_sync(); }
  __device__ __forceinline__ void team_sync() const { wave_sync(); }  // the team is this wave
  __device__ __forceinline__ Elem* scratch() const { return reinterpret_cast<Elem*>(buffers_ + (size_t)group() * buffer_bytes_); }
  __device__ __forceinline__ const Elem* scratch_of(int half) const { return reinterpret_cast<const Elem*>(buffers_ + (size_t)half * buffer_bytes_); }
  __device__ __forceinline__ u32* acc(int = 0) const { return accs_ + (size_t)group() * acc_words_; }
  __device__ __forceinline__ const Elem* twiddles() const { return tw_; }
  __device__ __forceinline__ const Elem* twiddles_uniform() const { return twg_; }
  __device__ __forceinline__ u32 uniform(u32 v) const { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ void lds_add(u32* p, u32 v) const {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void compiler_fence() const { asm volatile("" ::: "memory"); }
};

template <class F, int LOGN>
struct PairCfg {
  static constexpr int N = 1 << LOGN;
  static constexpr int kThreads = 64;
  static constexpr size_t kTwBytes = (size_t)staged_twiddle_words<F, LOGN>() * sizeof(typename F::elem);
  static constexpr unsigned kBufferBytes = (unsigned)N * 8u;
  static constexpr size_t kLds = kTwBytes + 2 * (size_t)kBufferBytes + 2 * (size_t)N * 4;
};

#ifndef TFHE_PAIR_MIN_WAVES
#define TFHE_PAIR_MIN_WAVES 2
#endif
template <class F, int LOGN>
__global__ void __launch_bounds__(64, TFHE_PAIR_MIN_WAVES)
blind_rotate_pair_kernel(PbsParams P, const typename F::elem* __restrict__ tw, const u32* __restrict__ lwe_in, size_t batch,
                         const u32* __restrict__ tv, size_t tv_stride, const typename F::elem* __restrict__ bsk,
                         u32* glwe_out, u32* __restrict__ lwe_extracted, u32 i_begin, u32 i_end, u32* glwe_state) {
  typedef typename F::elem elem;
  using C = PairCfg<F, LOGN>;
  constexpr int N = C::N;
  constexpr int T = 32;
  constexpr int EC = N / T;
  constexpr int K = 1;
  elem* twl = reinterpret_cast<elem*>(g_smem);
  ntt_stage_twiddles<LOGN - F::kLogShrink, 0, elem, staged_twiddle_words<F, LOGN>()>(twl, tw, (int)threadIdx.x, 64);
  __syncthreads();
  PairWave<elem> w;
  w.buffers_ = g_smem + C::kTwBytes;
  w.buffer_bytes_ = C::kBufferBytes;
  w.accs_ = reinterpret_cast<u32*>(w.buffers_ + 2 * (size_t)C::kBufferBytes);
  w.acc_words_ = (unsigned)N;
  w.tw_ = twl;
  w.twg_ = tw;
  const size_t sample = blockIdx.x;  // grid = batch
  (void)batch;
  const u32* resume = i_begin > 0 ? glwe_state + sample * (size_t)(K + 1) * N : nullptr;
  blind_rotate_pair<F, LOGN>(w, P, lwe_in + sample * (P.n + 1), tv + sample * tv_stride, bsk, i_begin, i_end, resume);
  const int tid = w.tid(), me = w.group();
  const u32* acc = w.acc();
  if (i_end < P.n) {  // not the last segment: park the accumulators
    u32* dst = glwe_state + (sample * (size_t)(K + 1) + me) * N;
#pragma unroll
    for (int r = 0; r < EC; ++r) dst[r * T + tid] = acc[r * T + tid];
    return;
  }
  if (glwe_out) {
    u32* dst = glwe_out + (sample * (size_t)(K + 1) + me) * N;
#pragma unroll
    for (int r = 0; r < EC; ++r) dst[r * T + tid] = acc[r * T + tid];
  }
  if (lwe_extracted) {  // sample_extract at index 0 (bootstrapping.rs:122-156)
    u32* out = lwe_extracted + sample * ((size_t)K * N + 1);
    if (me < K) {
#pragma unroll
      for (int r = 0; r < EC; ++r) {
        const int x = r * T + tid;
        out[me * N + x] = (x == 0) ? acc[0] : (0u - acc[N - x]);
      }
    } else if (tid == 0) {
      out[K * N] = acc[0];
    }
  }
}

// The kernel that rotates a shape's large batches, with what the launcher needs to know about it
template <class F, int LOGN, int K>
struct RotateKernel {
  using C = TeamCfg<F, LOGN, K>;
  static constexpr int kThreads = C::kThreads;
  static constexpr size_t kLds = C::kLds;
  static constexpr int kSamples = C::S;   // samples per workgroup
  static constexpr int kWaves = C::kWaves;
  static auto get() { return blind_rotate_kernel<F, LOGN, K>; }
};
template <int LOGN, int K>
struct RotateKernelPair {
  using C = PairCfg<FftField, LOGN>;
  static constexpr int kThreads = C::kThreads;
  static constexpr size_t kLds = C::kLds;
  static constexpr int kSamples = 1;
  static constexpr int kWaves = 1;
  static auto get() { return blind_rotate_pair_kernel<FftField, LOGN>; }
};
#ifndef TFHE_PAIR_KERNEL
#define TFHE_PAIR_KERNEL 1  // 0: A/B builds that keep the two-wave team at N = 512, k = 1 (the key keeps the pair layout)
#endif
template <class F, int LOGN, int K>
using PairOrTeamKernel = typename std::conditional<(TFHE_PAIR_KERNEL && pair_shape<F, LOGN, K>()), RotateKernelPair<LOGN, K>, RotateKernel<F, LOGN, K>>::type;

// ------------------------------------------------------------------------------ blind rotation, wide team
// The latency shape (pbs_wave.h::blind_rotate_team_wide): 2 (K+1) waves per sample -- wave (c, q) transforms half of
// polynomial c's digit levels, accumulates key part q of column c over all rows and inverse-transforms that one
// accumulator.  The complex transform up to N = 1024.  Picked by the launcher for batches that leave most CUs idle.
// LDS: [ twiddles ][ (K+1) l row buffers of N x 8 B ][ 2 (K+1) transpose buffers, one per wave ][ K+1 accumulator polynomials of N x 4 B ]
template <class Elem>
struct WideWave {
  unsigned char* rows_;  // row buffer 0
  Elem* scratch_;        // my own transpose buffer
  u32* acc_;             // my polynomial's accumulator
  const Elem* tw_;
  const Elem* twg_;
  unsigned row_bytes_;
  int wave_;             // 2 c + q
  __device__ __forceinline__ int tid() const { return (int)(threadIdx.x & 63u); }
  __device__ __forceinline__ int group() const { return wave_ >> 1; }
  __device__ __forceinline__ int half() const { return wave_ & 1; }
  __device__ __forceinline__ Elem* row_buffer(int r) const { return reinterpret_cast<Elem*>(rows_ + (size_t)r * row_bytes_); }
  __device__ __forceinline__ void wave_sync() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  __device__ __forceinline__ void poly_sync() const { wave_sync(); }  // one wave per transform
  __device__ __forceinline__ void team_sync() const { __syncthreads(); }
  __device__ __forceinline__ Elem* scratch() const { return scratch_; }
  __device__ __forceinline__ u32* acc(int = 0) const { return acc_; }
  __device__ __forceinline__ const Elem* twiddles() const { return tw_; }
  __device__ __forceinline__ const Elem* twiddles_uniform() const { return twg_; }
  __device__ __forceinline__ u32 uniform(u32 v) const { return __builtin_amdgcn_readfirstlane(v); }
  __device__ __forceinline__ void lds_add(u32* p, u32 v) const {
    (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void compiler_fence() const { asm volatile("" ::: "memory"); }
};

template <class F, int LOGN, int K>
struct WideCfg {
  static constexpr int N = 1 << LOGN;
  static constexpr int kWaves = 2 * (K + 1);
  static constexpr int kThreads = kWaves * 64;
  static constexpr size_t kTwBytes = (size_t)staged_twiddle_words<F, LOGN>() * sizeof(typename F::elem);
  static constexpr unsigned kRowBytes = (unsigned)N * 8u;
  static __host__ __device__ size_t row_buffers(u32 levels) { return (size_t)(K + 1) * levels + kWaves; }  // rows, then one per wave
  static __host__ __device__ size_t lds(u32 levels) { return kTwBytes + row_buffers(levels) * kRowBytes + (size_t)(K + 1) * N * 4; }
};

// Key ring of the wide team (pbs_wave.h::WideKeyRing): rows of a wave's key tiles held in registers ahead of time.
// LEVELS > 0 instantiates the kernel for that level count (the ring's slots are compile-time); LEVELS = 0 is the generic
// kernel (run-time level count, key chunks one ahead).  Register budget: a k = 1 team is four waves, one per SIMD, and
// one team per CU is all its LDS allows -- a wave may use the SIMD's whole file (512 registers: the allocator takes
// AGPRs beyond 256): 224 for the ring; a k = 2 team's six waves put two on two of the SIMDs: 256 each, 112 for the ring.
#ifndef TFHE_WIDE_MIN_WAVES_K2
#define TFHE_WIDE_MIN_WAVES_K2 2
#endif
template <class F, int LOGN, int K, int LEVELS>
__global__ void __launch_bounds__((WideCfg<F, LOGN, K>::kThreads), (K == 1 && LEVELS > 0 ? 1 : TFHE_WIDE_MIN_WAVES_K2))
blind_rotate_wide_kernel(PbsParams P, const typename F::elem* __restrict__ tw, const u32* __restrict__ lwe_in, size_t batch,
                         const u32* __restrict__ tv, size_t tv_stride, const typename F::elem* __restrict__ bsk,
                         u32* glwe_out, u32* __restrict__ lwe_extracted, u32 i_begin, u32 i_end, u32* glwe_state) {
  typedef typename F::elem elem;
  using C = WideCfg<F, LOGN, K>;
  constexpr int N = C::N;
  constexpr int EC = N / 64;
  elem* twl = reinterpret_cast<elem*>(g_smem);
  ntt_stage_twiddles<LOGN - F::kLogShrink, 1, elem, staged_twiddle_words<F, LOGN>()>(twl, tw, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  WideWave<elem> w;
  w.wave_ = (int)(threadIdx.x >> 6);
  w.rows_ = g_smem + C::kTwBytes;
  w.row_bytes_ = C::kRowBytes;
  w.scratch_ = reinterpret_cast<elem*>(w.rows_ + ((size_t)(K + 1) * P.levels + w.wave_) * C::kRowBytes);
  w.acc_ = reinterpret_cast<u32*>(w.rows_ + C::row_buffers(P.levels) * C::kRowBytes) + (size_t)w.group() * N;
  w.tw_ = twl;
  w.twg_ = tw;
  const size_t sample = blockIdx.x;  // grid = batch
  const u32* resume = i_begin > 0 ? glwe_state + sample * (size_t)(K + 1) * N : nullptr;
  blind_rotate_team_wide<F, LOGN, K, LEVELS, wide_ring_rows<LOGN, K, LEVELS>()>(w, P, lwe_in + sample * (P.n + 1), tv + sample * tv_stride, bsk,
                                                                               i_begin, i_end, resume);
  (void)batch;
  // each half stores the words it loaded: registers [q EC/2, (q + 1) EC/2) of polynomial c
  const int tid = w.tid(), me = w.group(), q = w.half();
  const u32* acc = w.acc();
  if (i_end < P.n) {  // not the last segment: park the accumulators
    u32* dst = glwe_state + (sample * (size_t)(K + 1) + me) * N;
#pragma unroll
    for (int r = 0; r < EC / 2; ++r) dst[(r + q * (EC / 2)) * 64 + tid] = acc[(r + q * (EC / 2)) * 64 + tid];
    return;
  }
  if (glwe_out) {
    u32* dst = glwe_out + (sample * (size_t)(K + 1) + me) * N;
#pragma unroll
    for (int r = 0; r < EC / 2; ++r) dst[(r + q * (EC / 2)) * 64 + tid] = acc[(r + q * (EC / 2)) * 64 + tid];
  }
  if (lwe_extracted) {  // sample_extract at index 0 (bootstrapping.rs:122-156), the words split between the halves
    u32* out = lwe_extracted + sample * ((size_t)K * N + 1);
    if (me < K) {
#pragma unroll
      for (int r = 0; r < EC / 2; ++r) {
        const int x = (r + q * (EC / 2)) * 64 + tid;
        out[me * N + x] = (x == 0) ? acc[0] : (0u - acc[N - x]);
      }
    } else if (q == 0 && tid == 0) {
      out[K * N] = acc[0];
    }
  }
}

// unrolled blind rotation (two key bits per step, pbs_wave.h::blind_rotate_bmmp_team); offered where a
// lane holds 8 elements per array and one wave owns a polynomial: N = 512
#ifndef TFHE_BMMP_MIN_WAVES
#define TFHE_BMMP_MIN_WAVES 3
#endif
template <class F, int LOGN, int K>
__global__ void __launch_bounds__((TeamCfg<F, LOGN, K, 1>::kThreads), TFHE_BMMP_MIN_WAVES)
blind_rotate_bmmp_kernel(PbsParams P, const typename F::elem* __restrict__ tw,
                         const u32* __restrict__ lwe_in, size_t batch, const u32* __restrict__ tv,
                         size_t tv_stride, const typename F::elem* __restrict__ bsk,
                         u32* __restrict__ glwe_out, u32* __restrict__ lwe_extracted) {
  using C = TeamCfg<F, LOGN, K, 1>;
  constexpr int N = C::N;
  constexpr int G = C::G;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  static_assert(G == 1 && C::EXB == 1, "one wave per polynomial, one exchange buffer");
  auto w = make_wave<F, LOGN, K, 1>(g_smem, tw);
  const size_t sample = blockIdx.x;
  blind_rotate_bmmp_team<F, LOGN, K, G>(w, P, lwe_in + sample * (P.n + 1), tv + sample * tv_stride, bsk);
  const int tid = w.tid();
  if (glwe_out) {
    u32* dst = glwe_out + (sample * (size_t)(K + 1) + w.group()) * N;
#pragma unroll
    for (int r = 0; r < E; ++r) dst[r * T + tid] = w.acc()[r * T + tid];
  }
  if (lwe_extracted) sample_extract_team<LOGN, K, G>(w, lwe_extracted + sample * ((size_t)K * N + 1));
}

// ------------------------------------------------------------------------------ external product
// Persistent grid: the launcher starts as many teams as are resident at once, so the twiddle table is
// staged into LDS once per team.  Short batches are handed out by stride (blockIdx.x, + gridDim.x, ...),
// long ones through a work queue.  Why both: a grid that exactly fills the chip with a FIXED share per
// team ends when its slowest CU does -- the same team code ran 7-12 % slower per product that way than
// under the dispatcher's dynamic placement of a larger grid -- so at 64 samples per team drawing tickets
// from a device counter is 5.7 % faster; with a handful of samples per team a queue only makes some
// teams take one more than the others (4 per team: +15 %, and +5 % even when only the tail is queued),
// so the launcher uses it from 16 samples per team (profiles/r02_external_product_experiments.txt).
// queue[0] hands out tickets; the launcher zeroes it on the stream right before the kernel (a memset node
// when the call is captured into a graph, so replays start from zero too).  Both forms of the reference call go through ONE
// instantiation of the team code: the plain product (ggsw.rs:132-161) and the CMUX form
// (ggsw.rs:164-178: ct1 -= ct0 is written back, the product is added to ct0); `cmux_ct0` is a kernel
// argument, so the selects below are wave-uniform branches.  (Measured and dropped: fetching the next
// sample's operand early -- by LDS-DMA or into registers behind the inverse transforms -- and starting
// the teams out of phase left the time per product unchanged within 2 %; with loads and stores compiled
// out the kernel is only 4 % faster.)
template <class F, int LOGN, int K>
__global__ void __launch_bounds__((TeamCfg<F, LOGN, K>::kThreads),
                                  (F::kId == FpField::kId || F::kId == Fp49Field::kId || F::kId == FftField::kId
                                       ? TeamCfg<F, LOGN, K, 1>::kMinWavesFp
                                       : TeamCfg<F, LOGN, K, 1>::kMinWavesGl))
external_product_kernel(PbsParams P, const typename F::elem* __restrict__ tw,
                        const typename F::elem* __restrict__ ggsw, size_t ggsw_stride_words,
                        const u32* glwe_in, u32* ct1_inout, const u32* cmux_ct0, size_t batch,
                        u32* glwe_out, unsigned long long* queue /* null: by stride */) {
  using C = TeamCfg<F, LOGN, K, 1>;  // one product per team at a time (the samples have their own GGSWs in general)
  constexpr int N = C::N;
  constexpr int G = C::G;
  auto w = make_wave<F, LOGN, K, 1>(g_smem, tw);
  const bool is_cmux = cmux_ct0 != nullptr;
  // the ticket of the team's current sample, published through LDS (behind the team's arrays)
  unsigned long long* ticket = reinterpret_cast<unsigned long long*>(g_smem + C::kLds);
  for (size_t turn = 0;; ++turn) {
    size_t sample;
    if (queue == nullptr) {
      sample = blockIdx.x + turn * gridDim.x;
    } else {
      if (threadIdx.x == 0) *ticket = atomicAdd(queue, 1ull);
      __syncthreads();  // every wave of the team sees the same ticket: the barriers below stay matched
      sample = (size_t)*ticket;
    }
    if (sample >= batch) break;
    const size_t poly = (sample * (size_t)(K + 1) + w.group()) * N;  // my polynomial / my output column
    const typename F::elem* g = ggsw + sample * ggsw_stride_words;
    const u32* in = is_cmux ? ct1_inout + poly : glwe_in + poly;
    const u32* c0 = cmux_ct0 + poly;
    u32* c1 = ct1_inout + poly;
    u32* dst = glwe_out + poly;
    // CMUX: each coefficient is read and written by the one lane that owns index j, so the in-place
    // update of ct1 (*glwe_ciphertext1 -= glwe_ciphertext0, ggsw.rs:171) is race free
    auto src = [&](int j) -> u32 {
      u32 d = __builtin_nontemporal_load(&in[j]);
      if (is_cmux) {
        d -= c0[j];
        c1[j] = d;
      }
      return d;
    };
    auto out = [&](int j, u32 v) { __builtin_nontemporal_store(is_cmux ? v + c0[j] : v, &dst[j]); };
    external_product_team<F, LOGN, K, G>(w, P, g, src, out);
    // (thread 0 overwrites the ticket only after the product's team barriers, which every wave passes
    // after it has read the ticket)
  }
}

// ------------------------------------------------------------------------------ key switch
// out[b][c] = -sum_{i<big_n, l<levels} digit_l(lwe[b][i]) * ksk[i*levels + l][c];  out[b][n] += b
// Tiled as a wrapping-u32 GEMM: a workgroup owns kKsSamples samples x 128 output columns and walks
// key rows in chunks; digits of the chunk are produced once into LDS.  The key rows are staged through
// LDS as well, kKsKeyRows at a time and double-buffered: the four waves of a workgroup all need the same
// 128 columns of every row (they differ in the samples they own), and reading them straight from global
// memory quadrupled the L1 traffic -- 512 B per wave and row at ~75 cycles per row and ten waves per CU is
// more than the 64 B/clk a CU's vector L1 delivers, which is what bounded the kernel (1.30 ms against a
// 0.39 ms multiply-add floor at cfg2).  Small batches do not give enough (sample, column) tiles to fill
// 256 CUs, so the big_n mask words are also split over gridDim.z: with more than one split the partial
// sums go to the (pre-zeroed) output with u32 atomic adds -- wrapping addition is associative and
// commutative, the bits do not depend on the order.
constexpr int kKsSamples = 32;     // samples per workgroup
constexpr int kKsColsPerLane = 2;  // output columns per lane
constexpr int kKsCols = 64 * kKsColsPerLane;  // output columns per workgroup
constexpr int kKsWords = 8;        // mask words decomposed per chunk
constexpr int kKsPerThread = 8;    // samples per thread (kKsSamples / 4 waves)
constexpr int kKsKeyRows = 16;     // key rows staged in LDS per step (x 128 columns x 4 B = 8 KiB, two buffers)
constexpr int kKsLoads = kKsKeyRows * kKsCols / 256;  // key words each thread moves per step
// workgroups a launch should reach before the mask words stop being split over gridDim.z: 16 per CU
// (cfg2, batch 4096: 640 tiles -> 0.90 ms unsplit, 0.70 ms at 3 splits, 0.59 ms at 6; flat beyond)
#ifndef TFHE_KS_TARGET_WGS
#define TFHE_KS_TARGET_WGS 4096u
#endif

__global__ void __launch_bounds__(256) key_switch_kernel(KsParams Kp, u32 big_n, u32 n,
                                                         const u32* __restrict__ lwe_in, size_t batch,
                                                         const u32* __restrict__ ksk,
                                                         u32* __restrict__ lwe_out, u32 words_per_split) {
  const u32 levels = Kp.levels;
  u32* dig = reinterpret_cast<u32*>(g_smem);                 // [kKsWords*levels][kKsSamples]
  u32* ktile = dig + (size_t)kKsWords * levels * kKsSamples;  // [2][kKsKeyRows][kKsCols]
  const int tx = (int)(threadIdx.x & 63u);
  const int ty = (int)(threadIdx.x >> 6);
  const size_t s0 = (size_t)blockIdx.y * kKsSamples;
  const u32 width = n + 1;
  const u32 col0 = blockIdx.x * kKsCols;
  u32 col[kKsColsPerLane];
  bool col_ok[kKsColsPerLane];
#pragma unroll
  for (int c = 0; c < kKsColsPerLane; ++c) {
    col[c] = col0 + c * 64 + tx;
    col_ok[c] = col[c] < width;
  }
  // staging slot of this thread: column lc of rows lr, lr + 2, ... of a step (a wave covers 64
  // consecutive columns of one row: coalesced)
  const u32 lc = threadIdx.x & (kKsCols - 1);
  const u32 lr = threadIdx.x / kKsCols;
  const bool lc_ok = col0 + lc < width;

  // 64-bit accumulators so that every multiply-add is ONE v_mad_u64_u32; only bits 31..0 are kept
  u64 acc[kKsColsPerLane][kKsPerThread];
#pragma unroll
  for (int c = 0; c < kKsColsPerLane; ++c)
#pragma unroll
    for (int s = 0; s < kKsPerThread; ++s) acc[c][s] = 0;

  const u32 w_begin = blockIdx.z * words_per_split;  // multiple of kKsWords
  const u32 w_end = (w_begin + words_per_split < big_n) ? w_begin + words_per_split : big_n;
  for (u32 w0 = w_begin; w0 < w_end; w0 += kKsWords) {
    const u32 rows = ((w_end - w0 < (u32)kKsWords) ? (w_end - w0) : (u32)kKsWords) * levels;
    const u32* krow = ksk + (size_t)w0 * levels * width + col0;
    u32 stage[kKsLoads];
    auto fetch = [&](u32 first_row) {  // rows first_row + lr + 2 i of this chunk -> registers
#pragma unroll
      for (int i = 0; i < kKsLoads; ++i) {
        const u32 r = first_row + lr + (256 / kKsCols) * i;
        stage[i] = (lc_ok && r < rows) ? krow[(size_t)r * width + lc] : 0u;
      }
    };
    fetch(0);
    // 256 threads decompose 32 samples x 8 words: thread -> (sample = tid / 8, word = tid % 8)
    {
      const int sl = (int)(threadIdx.x >> 3);
      const u32 wl = threadIdx.x & 7u;
      const size_t sample = s0 + sl;
      const u32 word = w0 + wl;
      u32 v = 0;
      if (sample < batch && word < w_end) v = lwe_in[sample * ((size_t)big_n + 1) + word];
      v = round_value(v, Kp.ignored_bits);
      u32 carry = 0;
      for (u32 t = 0; t < levels; ++t) {  // LSB -> MSB, level index counts from the MSB
        const u32 d = decompose_limb(v, Kp.first_shift + Kp.log_base * t, Kp.log_base, carry);
        dig[(wl * levels + (levels - 1 - t)) * kKsSamples + sl] = d;
      }
    }
    // one barrier per step: a step writes buffer (step & 1), which was last read two steps ago, and
    // every thread has passed the barrier of the step in between since (the first barrier of a chunk
    // also publishes the digits; the chunk's last one, below, protects them from the next chunk)
    for (u32 r0 = 0, step = 0; r0 < rows; r0 += kKsKeyRows, ++step) {
      u32* kt = ktile + (step & 1u) * (kKsKeyRows * kKsCols);
#pragma unroll
      for (int i = 0; i < kKsLoads; ++i) kt[(lr + (256 / kKsCols) * i) * kKsCols + lc] = stage[i];
      __syncthreads();
      if (r0 + kKsKeyRows < rows) fetch(r0 + kKsKeyRows);  // in flight while this step computes
      const u32 here = rows - r0 < (u32)kKsKeyRows ? rows - r0 : (u32)kKsKeyRows;
#pragma unroll 4
      for (u32 r = 0; r < here; ++r) {
        u32 kv[kKsColsPerLane];
#pragma unroll
        for (int c = 0; c < kKsColsPerLane; ++c) kv[c] = kt[r * kKsCols + c * 64 + tx];
        const u32* d = dig + (r0 + r) * kKsSamples + ty * kKsPerThread;
#pragma unroll
        for (int s = 0; s < kKsPerThread; ++s) {
          const u32 ds = d[s];  // same address in every lane of the wave: LDS broadcast
#pragma unroll
          for (int c = 0; c < kKsColsPerLane; ++c) acc[c][s] = (u64)ds * kv[c] + acc[c][s];
        }
      }
    }
    __syncthreads();
  }

#pragma unroll
  for (int c = 0; c < kKsColsPerLane; ++c) {
    if (!col_ok[c]) continue;
#pragma unroll
    for (int s = 0; s < kKsPerThread; ++s) {
      const size_t sample = s0 + ty * kKsPerThread + s;
      if (sample >= batch) continue;
      u32 v = 0u - (u32)acc[c][s];
      if (col[c] == n && blockIdx.z == 0) v += lwe_in[sample * ((size_t)big_n + 1) + big_n];
      if (gridDim.z == 1) lwe_out[sample * width + col[c]] = v;
      else atomicAdd(&lwe_out[sample * width + col[c]], v);
    }
  }
}

// ------------------------------------------------------------------------------ elementwise
__global__ void decompose_words_kernel(u32 log_base, u32 levels, u32 ignored_bits, u32 first_shift,
                                       const u32* __restrict__ values, size_t count,
                                       u32* __restrict__ digits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (size_t)gridDim.x * blockDim.x) {
    const u32 v = round_value(values[i], ignored_bits);
    u32 carry = 0;
    for (u32 t = 0; t < levels; ++t)
      digits[i * levels + (levels - 1 - t)] = decompose_limb(v, first_shift + log_base * t, log_base, carry);
  }
}

// glwe [batch][polys][N] -> digits [batch][polys*levels][N]
__global__ void decompose_glwe_kernel(u32 log_base, u32 levels, u32 ignored_bits, u32 first_shift,
                                      u32 polys, u32 n_coeff, const u32* __restrict__ glwe,
                                      size_t total /* batch*polys*N */, u32* __restrict__ digits) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t poly = i / n_coeff;  // global polynomial index = b*polys + p
    const u32 j = (u32)(i % n_coeff);
    const u32 v = round_value(glwe[i], ignored_bits);
    u32 carry = 0;
    for (u32 t = 0; t < levels; ++t)
      digits[(poly * levels + (levels - 1 - t)) * n_coeff + j] =
          decompose_limb(v, first_shift + log_base * t, log_base, carry);
  }
}

__global__ void switch_modulus_kernel(const u32* __restrict__ values, size_t count, u32 log_from,
                                      u32 log_to, u32* __restrict__ out) {
  // utils.rs:13-33
  const u32 sh = log_from - log_to;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (size_t)gridDim.x * blockDim.x) {
    u32 v = values[i];
    if (sh == 0) {
      // divisor 1: rational = v, fractional = 0
    } else {
      v = (v >> sh) + ((v >> (sh - 1)) & 1u);
    }
    out[i] = (log_to >= 32) ? v : (v & ((1u << log_to) - 1u));
  }
}

__global__ void glwe_mul_monomial_kernel(u32 log_n, u32 polys, const u32* __restrict__ glwe,
                                         size_t total, const i64* __restrict__ monomial_index,
                                         u32* __restrict__ out) {
  const u32 N = 1u << log_n;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t poly = i >> log_n;
    const u32 j = (u32)(i & (N - 1));
    // `monomial_index as usize % (2*n)` (utils.rs:186): two's complement residue mod 2N
    const u32 m = (u32)((u64)monomial_index[poly / polys] & (u64)(2 * N - 1));
    const u32 deg = m & (N - 1);
    const u32 flip = (m >> log_n) & 1u;
    const u32 v = glwe[(poly << log_n) + ((j - deg) & (N - 1))];
    out[i] = (flip ^ (u32)(j < deg)) ? (0u - v) : v;
  }
}

__global__ void sample_extract_kernel(u32 log_n, u32 k, const u32* __restrict__ glwe, size_t batch,
                                      u32 sample_index, u32* __restrict__ lwe_out) {
  // bootstrapping.rs:122-156: per mask polynomial p[idx], p[idx-1], ..., p[0], -p[N-1], ..., -p[idx+1]
  const u32 N = 1u << log_n;
  const size_t width = (size_t)k * N + 1;
  const size_t total = batch * width;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t b = i / width;
    const size_t x = i % width;
    const u32* ct = glwe + b * (size_t)(k + 1) * N;
    u32 v;
    if (x == (size_t)k * N) {
      v = ct[(size_t)k * N + sample_index];
    } else {
      const u32 p = (u32)(x >> log_n);
      const u32 pos = (u32)(x & (N - 1));
      v = (pos <= sample_index) ? ct[(size_t)p * N + (sample_index - pos)]
                                : (0u - ct[(size_t)p * N + (N + sample_index - pos)]);
    }
    lwe_out[i] = v;
  }
}

// out = c0*ct0 + c1*ct1, plus b_add on the last word of every ciphertext of words_per_ct words
// (words_per_ct == 0: no constant term)
__global__ void lwe_linear_kernel(u32 c0, const u32* ct0, u32 c1, const u32* ct1, size_t words,
                                  size_t words_per_ct, u32 b_add, u32* out) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words;
       i += (size_t)gridDim.x * blockDim.x) {
    u32 v = c0 * ct0[i] + (ct1 ? c1 * ct1[i] : 0u);
    if (words_per_ct && (i % words_per_ct) == words_per_ct - 1) v += b_add;
    out[i] = v;
  }
}

// ------------------------------------------------------------------------------ encryption side
// dst[row][j] = body_in[row][j] +/- sum_i masks[row][i] (*) sk[i]: one GLWE row [k+1][N] per group
// of G waves, 4/G rows per workgroup.  Encrypt (glwe.rs:190-209): body_in holds the caller's error
// samples, dst = the body itself.  Decrypt (glwe.rs:245-265): dst = plaintext rows, negate = 1.
template <class F, int LOGN>
__global__ void __launch_bounds__(256) glwe_body_kernel(const typename F::elem* __restrict__ tw, u32 k,
                                                       const u32* rows, size_t row_count,
                                                       const u32* __restrict__ sk, u32* dst,
                                                       size_t dst_stride, u32 negate,
                                                       typename F::elem n_inv) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int G = GroupOf<F, LOGN>::value;
  elem* twl = reinterpret_cast<elem*>(g_smem);
  ntt_stage_twiddles<LOGN - F::kLogShrink, G>(twl, tw, (int)threadIdx.x, (int)blockDim.x);
  __syncthreads();
  const int group = (int)(threadIdx.x / (64u * G));
  const int groups = (int)(blockDim.x / (64u * G));
  // G > 1: the transforms contain workgroup barriers, so a group past the end redoes the last row
  // (same values to the same addresses) instead of leaving -- as in bsk_prepare_kernel
  size_t row = (size_t)blockIdx.x * groups + group;
  bool live = true;
  if (row >= row_count) {
    if (G == 1) return;
    row = row_count - 1;
    live = false;
  }
  DeviceWave<elem, G> w;
  w.group_ = 0;
  w.group_stride_ = 0;
  w.buffer_bytes_ = 0;
  w.acc_words_ = 0;
  w.team_base_ = nullptr;
  w.tw_ = twl;
  w.twg_ = tw;
  w.scratch_ = reinterpret_cast<elem*>(g_smem + twiddle_bytes<F, LOGN>() + (size_t)group * N * 8);
  w.acc_ = nullptr;
  const u32* masks = rows + row * (size_t)(k + 1) * N;
  const u32* body = masks + (size_t)k * N;
  u32* d = dst + row * dst_stride;
  auto out = [&](int j, u32 dot) {
    // body[j] is read before d[j] is written by the same lane, so d may alias body
    const u32 b = body[j];
    if (live) d[j] = negate ? b - dot : b + dot;
  };
  glwe_mask_dot_key<F, LOGN, G>(w, k, masks, sk, n_inv, out);
}

// dst[row] = lwe[row][n] +/- <lwe[row][0..n), sk> (+ plaintext[row]): one LWE row per wavefront.
// Encrypt (lwe.rs:117-160, and the rows of generate_ksk key_switching.rs:41-45): the b slot holds
// the caller's error sample, dst = the b slot.  Decrypt (lwe.rs:162-173): negate = 1.
__global__ void __launch_bounds__(256) lwe_body_kernel(const u32* rows, size_t row_count, u32 n,
                                                      const u32* __restrict__ sk,
                                                      const u32* __restrict__ plaintext, u32* dst,
                                                      size_t dst_stride, u32 negate) {
  const size_t row = (size_t)blockIdx.x * (blockDim.x / 64u) + threadIdx.x / 64u;
  if (row >= row_count) return;
  const u32 lane = threadIdx.x & 63u;
  const u32* ct = rows + row * ((size_t)n + 1);
  u32 dot = 0;
  for (u32 j = lane; j < n; j += 64u) dot += ct[j] * sk[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
  if (lane == 0) {
    const u32 b = ct[n];
    u32 v = negate ? b - dot : b + dot;
    if (plaintext) v += plaintext[row];
    dst[row * dst_stride] = v;
  }
}

// ggsw.rs:96-103: row = poly_index*levels + level of GGSW g gets
// messages[g] * 2^{gadget_top - log_base*(level+1)} added to coefficient 0 of polynomial poly_index
// (after the zero encryption was formed from the unmodified masks); gadget_top =
// log_base*floor(32/log_base) for the reference's literal decomposer
__global__ void ggsw_add_gadget_kernel(u32* ggsw, size_t ggsw_count, u32 k, u32 log_n, u32 levels,
                                       u32 log_base, u32 gadget_top, const u32* __restrict__ messages) {
  const size_t rows = (size_t)(k + 1) * levels;
  const size_t total = ggsw_count * rows;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t g = i / rows;
    const u32 row = (u32)(i % rows);
    const u32 poly_index = row / levels, level = row % levels;
    const u32 factor = messages[g] * (1u << (gadget_top - log_base * (level + 1)));
    ggsw[(((g * rows + row) * (k + 1)) + poly_index) << log_n] += factor;
  }
}

// HBM roofline probe: 16-byte-per-lane stream copy (the "float4 copy" of MI355X_MICROARCH.md).
// A workgroup moves one contiguous 16 KiB piece: every lane issues its 4 loads before the first
// store, so 64 B per lane are in flight; loads and stores are non-temporal (nothing is reused).
constexpr int kCopyUnroll = 4;
typedef unsigned int copy_vec4 __attribute__((ext_vector_type(4)));  // 16 bytes, what the builtins accept
__global__ void __launch_bounds__(256) stream_copy_kernel(const copy_vec4* __restrict__ src,
                                                         copy_vec4* __restrict__ dst, size_t count) {
  const size_t base = (size_t)blockIdx.x * (256 * kCopyUnroll) + threadIdx.x;
  copy_vec4 v[kCopyUnroll];
#pragma unroll
  for (int i = 0; i < kCopyUnroll; ++i)
    if (base + (size_t)i * 256 < count) v[i] = __builtin_nontemporal_load(&src[base + (size_t)i * 256]);
#pragma unroll
  for (int i = 0; i < kCopyUnroll; ++i)
    if (base + (size_t)i * 256 < count) __builtin_nontemporal_store(v[i], &dst[base + (size_t)i * 256]);
}

inline int grid_for(size_t work, int block) {
  size_t g = (work + block - 1) / block;
  if (g > 2048) g = 2048;  // 256 CUs x 8: grid-stride the rest
  if (g == 0) g = 1;
  return (int)g;
}

// Dynamic LDS above 64 KiB has to be enabled per kernel; done once per kernel and device (the
// attribute call is not something to repeat on every launch of a short kernel).  `done` is a bit
// mask over devices owned by the calling launcher, which is a distinct function (and so a distinct
// flag) per template instantiation; two host threads racing here both set the same attribute to
// the same value, which is harmless.
template <typename Kern>
hipError_t allow_lds(Kern kern, size_t bytes, std::atomic<unsigned long long>& done) {
  if (bytes <= 64 * 1024) return hipSuccess;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  const unsigned long long bit = 1ull << dev;
  if (done.load(std::memory_order_acquire) & bit) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) done.fetch_or(bit, std::memory_order_release);
  return e;
}

// How a batch's blind rotations go out (launch_blind_rotate).
//   chunk     samples per group of launches
//   segments  launches per rotation: each walks key rows [i0, i1) only and parks the accumulators in global memory
//   streams   2: the two halves of a chunk go out on two streams, segment by segment in turn
// Why.  A team reads GGSW_i once per iteration (cfg2: 197 KB, 124 MB per rotation; the reference's defaults: 442 KB,
// 319 MB; N = 2048 with k = 2: 1.2 MB, 743 MB) and the complex transform's kernels draw 13-15 TB/s of it: that is L2
// bandwidth, and only while the teams that run at the same time read the SAME rows.  A launch of 4,096 samples is four
// rounds of the 1,024 teams the chip holds, and the dispatcher refills them one by one: teams of different rounds are at
// different iterations, most of the key is in use somewhere, and it comes from the Infinity Cache -- or, when it is larger
// than that (> 256 MiB), from HBM, again and again (PMC, cfg3: 258 GB per launch = 4 TB/s).  Round 2 kept launches short
// for that reason (4,096 samples: 109 k PBS/s at cfg2 against 75 k in one launch of 131,072).  Round 3 blocks the key
// instead (profiles/r03_kernel_ab.txt):
//   - SEGMENT launches: all teams of a launch, whatever their round, read the same slice of the key; a slice of a few MB
//     stays in the L2s.  Batch 4,096 on one stream: cfg3 62.2 k -> 71.2 k PBS/s from 32 segments on, cfg5 18.8 k -> 20.95 k
//     from 4 on (slices of <= 186 MB: the Infinity Cache holds them; no more from smaller ones), cfg2 and cfg1 (124 and
//     33 MB: they fit as they are) nothing: what the slices gain, every launch's tail -- the chip draining while the last
//     teams finish -- takes back.
//   - TWO STREAMS: the halves of the batch alternate, so one half's tail is filled by the other half's next launch.
//     Batch 4,096 with slices of 2 MiB: cfg2 117.6 k -> 125.2 k, cfg3 71.2 k -> 77.9 k, cfg1 358 k -> 387 k; it pays
//     as soon as the batch exceeds the teams the chip holds at once (at exactly that many it loses 2-4 %); three and
//     four streams add nothing; cfg5's 12-wave teams (one per CU) gain nothing either.
//   - with L2-sized slices launches can be LONG: cfg4's 131,072: 118.3 k in launches of 4,096 -> 128-130 k in one group.
//   (Measured and dropped: the same blocking inside ONE persistent launch -- teams that are never re-dispatched never
//   re-synchronise, 48 k at cfg3.)
// Hence, at N <= 1024: batches above what the chip holds at once go out in groups of up to 131,072 samples over key slices
// of half an L2 (l2_key_slice: 2 MiB on gfx950), on two streams; smaller ones in one launch.  That holds for every field (two streams + slices at batch 4,096:
// fp64-p49 at the reference's defaults 54.2 k -> 60.7 k PBS/s, Goldilocks at cfg2 29.2 k -> 30.8 k, fp64-p42 at cfg2 75.6 k
// -> 77.0 k).  At N = 2048: slices of 128 MiB if the key exceeds the Infinity Cache, one stream; the complex transform's
// long batches in groups of 16,384.  Without a place to park accumulators (state == nullptr): whole rotations, 4,096
// samples per launch for the complex transform (round 2: 109 k against 75 k PBS/s in one launch of 131,072), one launch
// for the prime fields (VALU-bound, they lose 2.5 % to launch boundaries).
// At least 4 iterations per segment.  TFHE_BR_CHUNK, TFHE_BR_SEGMENTS and TFHE_BR_STREAMS override (tests, sweeps).
struct BlindRotatePlan {
  size_t chunk;
  u32 segments;
  int streams;
};
inline long env_number(const char* name) {
  const char* env = std::getenv(name);
  const long v = env ? std::atol(env) : 0;
  return v > 0 ? v : 0;
}
constexpr int kMaxDevices = 64;
inline int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
// The key slice a segment launch walks: HALF of one L2 of the current device (an XCD's L2 on gfx950: 4 MiB, so 2 MiB -- the
// size the sweeps of round 3 settled on: the other half holds what the teams stream beside the key), asked once per device.
inline size_t l2_key_slice() {
  static std::atomic<size_t> cached[kMaxDevices];
  const int dev = current_device_slot();
  size_t slice = cached[dev].load(std::memory_order_acquire);
  if (slice == 0) {
    int l2 = 0;
    if (hipDeviceGetAttribute(&l2, hipDeviceAttributeL2CacheSize, dev) != hipSuccess || l2 <= 0) l2 = 4 << 20;
    slice = (size_t)l2 / 2;
    if (slice < ((size_t)1 << 20)) slice = (size_t)1 << 20;
    if (slice > ((size_t)16 << 20)) slice = (size_t)16 << 20;
    cached[dev].store(slice, std::memory_order_release);
  }
  return slice;
}
template <class F>
inline BlindRotatePlan blind_rotate_plan(size_t batch, bool can_park, bool have_side, u32 n, size_t key_bytes, u32 log_n,
                                         size_t resident_samples) {
  static const long env_chunk = env_number("TFHE_BR_CHUNK"), env_segments = env_number("TFHE_BR_SEGMENTS"),
                    env_streams = env_number("TFHE_BR_STREAMS");
  BlindRotatePlan plan{F::kLogShrink ? (size_t)4096 : (size_t)1 << 20, 1u, 1};
  if (can_park) {
    const bool fits = key_bytes <= ((size_t)224 << 20);  // the Infinity Cache, with room for the accumulators
    size_t slice = 0;
    if (log_n >= 11) {
      slice = fits ? 0 : (size_t)128 << 20;
      if (F::kLogShrink && batch >= 16384) plan.chunk = 16384;
    } else if (batch > resident_samples) {
      slice = l2_key_slice();
      plan.chunk = 131072;
      plan.streams = have_side ? 2 : 1;
    } else {
      plan.chunk = batch ? batch : 1;
    }
    if (env_streams) plan.streams = have_side && env_streams >= 2 ? 2 : 1;
    if (plan.streams == 2 && slice == 0) slice = l2_key_slice();
    plan.segments = slice ? (u32)((key_bytes + slice - 1) / slice) : 1u;
    if (plan.segments > n / 4) plan.segments = n / 4;
    if (env_segments) plan.segments = (u32)env_segments;
    if (plan.segments > n) plan.segments = n;
    if (plan.segments < 1) plan.segments = 1;
    const u32 per = (n + plan.segments - 1) / plan.segments;  // iterations per launch ...
    plan.segments = (n + per - 1) / per;                      // ... and the launches that makes
  }
  if (env_chunk) plan.chunk = (size_t)env_chunk;
  return plan;
}

// teams of blind_rotate_kernel<F, LOGN, K> the CURRENT device holds at once (occupancy x CUs), asked once per
// instantiation and device (a pool spans devices: nothing about one device is assumed of another)
template <class RK>
hipError_t resident_teams(unsigned* out) {
  static std::atomic<unsigned> cached[kMaxDevices];
  const int dev = current_device_slot();
  unsigned capacity = cached[dev].load(std::memory_order_acquire);
  if (capacity == 0) {
    auto kern = RK::get();
    int per_cu = 0, cus = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, RK::kThreads, RK::kLds);
    if (e != hipSuccess) return e;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
    capacity = (unsigned)(per_cu > 0 ? per_cu : 1) * (unsigned)(cus > 0 ? cus : 256);
    cached[dev].store(capacity, std::memory_order_release);
  }
  *out = capacity;
  return hipSuccess;
}

// Which kernel rotates a batch of the PAIR shape (N = 512, k = 1, complex transform) that is too large for the wide team:
// one wave per sample takes ~2x the time of a two-wave team to finish ONE sample, and 2,048 of them fill the chip -- so up
// to what the chip holds as two-wave teams (1,536 samples) the team is faster (1,024 samples: 2.96 against 4.2 ms); above,
// the pair kernel's third fewer LDS cycles and 8 instead of 6 samples per CU win (4,096: 9.2 against 10.0 ms;
// profiles/r04_kernel_ab.txt).  TFHE_BR_PAIR_MIN overrides the threshold (batches >= it take the pair kernel).
template <class F, int LOGN, int K>
bool use_pair_kernel(size_t batch) {
  if constexpr (!(TFHE_PAIR_KERNEL && pair_shape<F, LOGN, K>())) {
    return false;
  } else {
    static const long env_min = std::getenv("TFHE_BR_PAIR_MIN") ? std::atol(std::getenv("TFHE_BR_PAIR_MIN")) : -1;
    if (env_min >= 0) return batch >= (size_t)env_min;
    unsigned team_capacity = 0;
    if (resident_teams<RotateKernel<F, LOGN, K>>(&team_capacity) != hipSuccess) return true;
    return batch > (size_t)team_capacity * RotateKernel<F, LOGN, K>::kSamples;
  }
}

// The wide team (blind_rotate_wide_kernel) is offered for the complex transform up to N = 1024 while its row buffers fit
// the CU's LDS; it is USED for batches up to `wide_max_batch`: below that the chip has CUs to spare and a sample's latency
// -- not the chip's throughput -- is what a caller waits for (profiles/r04_batch_sweep*.txt).  TFHE_BR_WIDE overrides the
// threshold (0: never; n: batches up to n).
template <class F, int LOGN, int K>
constexpr bool wide_shape_ok() {
  return F::kLogShrink == 1 && F::kParts == 2 && LOGN <= 10 && field_shape_ok<F, LOGN>();
}
template <class F, int LOGN, int K>
size_t wide_max_batch(const PbsParams& P, int shape) {
  if constexpr (!wide_shape_ok<F, LOGN, K>()) {
    return 0;
  } else {
    using W = WideCfg<F, LOGN, K>;
    if (W::lds(P.levels) > (size_t)160 * 1024 || shape == launch::kShapeTeam) return 0;
    if (shape == launch::kShapeWide) return ~(size_t)0;
    static const long env_wide = std::getenv("TFHE_BR_WIDE") ? std::atol(std::getenv("TFHE_BR_WIDE")) : -1;
    if (env_wide >= 0) return (size_t)env_wide;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, current_device_slot()) != hipSuccess || cus <= 0) cus = 256;
#ifndef TFHE_WIDE_BATCH_PER_CU
#define TFHE_WIDE_BATCH_PER_CU 1
#endif
    return (size_t)cus * TFHE_WIDE_BATCH_PER_CU;
  }
}

template <class F, int LOGN, int K>
hipError_t launch_blind_rotate_wide(hipStream_t s, const PbsParams& P, const typename F::elem* tw, const u32* lwe_in, size_t batch,
                                    const u32* tv, size_t tv_stride, const typename F::elem* bsk, u32* glwe_out,
                                    u32* lwe_extracted) {
  if constexpr (!wide_shape_ok<F, LOGN, K>()) {
    return hipErrorInvalidValue;
  } else {
    using W = WideCfg<F, LOGN, K>;
    // the level counts with a key-ring instantiation (the BASELINE configurations' 2, 3, 6 and the full-word 4); any other
    // count runs the generic kernel
    auto launch = [&](auto levels_c) -> hipError_t {
      constexpr int LEVELS = decltype(levels_c)::value;
      auto kern = blind_rotate_wide_kernel<F, LOGN, K, LEVELS>;
      static std::atomic<unsigned long long> lds_done{0};
      hipError_t e = allow_lds(kern, (size_t)160 * 1024, lds_done);  // the size depends on the level count: allow the CU's whole LDS
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(W::kThreads), W::lds(P.levels), s, P, tw, lwe_in, batch, tv, tv_stride,
                         bsk, glwe_out, lwe_extracted, 0u, P.n, static_cast<u32*>(nullptr));
      return hipGetLastError();
    };
#ifndef TFHE_WIDE_KEY_RING
#define TFHE_WIDE_KEY_RING 1
#endif
    if (TFHE_WIDE_KEY_RING) {
      switch (P.levels) {
        case 2: return launch(IntC<2>{});
        case 3: return launch(IntC<3>{});
        case 4: return launch(IntC<4>{});
        case 6: return launch(IntC<6>{});
        default: break;
      }
    }
    return launch(IntC<0>{});
  }
}

template <class RK, class F, int LOGN, int K>
hipError_t plan_blind_rotate_with(const PbsParams& P, size_t batch, bool can_park, bool have_side, launch::BlindRotatePlanInfo* out, int shape) {
  {
    using C = TeamCfg<F, LOGN, K>;
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(RK::get(), RK::kLds, lds_done);
    unsigned capacity = 0;
    if (e == hipSuccess) e = resident_teams<RK>(&capacity);
    if (e != hipSuccess) return e;
    const size_t key_bytes = (size_t)P.n * (K + 1) * P.levels * (K + 1) * F::kParts * C::N * 8;
    const BlindRotatePlan plan = blind_rotate_plan<F>(batch, can_park, have_side, P.n, key_bytes, (u32)LOGN, (size_t)capacity * RK::kSamples);
    out->chunk = plan.chunk;
    out->segments = plan.segments;
    out->streams = plan.streams == 2 && plan.segments > 1 ? 2 : 1;
    out->resident_samples = (size_t)capacity * RK::kSamples;
    out->samples_per_team = RK::kSamples;
    out->waves_per_sample = RK::kWaves;
    if (batch > 0 && batch <= wide_max_batch<F, LOGN, K>(P, shape)) {  // the wide team: one launch over the whole key
      out->chunk = batch;
      out->segments = 1;
      out->streams = 1;
      out->samples_per_team = 1;
      out->waves_per_sample = 2 * (K + 1);
    }
    return hipSuccess;
  }
}

template <class F, int LOGN, int K>
hipError_t plan_blind_rotate(const PbsParams& P, size_t batch, bool can_park, bool have_side, launch::BlindRotatePlanInfo* out, int shape) {
  if constexpr (!field_shape_ok<F, LOGN>()) {
    return hipErrorInvalidValue;
  } else {
    if (use_pair_kernel<F, LOGN, K>(batch))
      return plan_blind_rotate_with<PairOrTeamKernel<F, LOGN, K>, F, LOGN, K>(P, batch, can_park, have_side, out, shape);
    return plan_blind_rotate_with<RotateKernel<F, LOGN, K>, F, LOGN, K>(P, batch, can_park, have_side, out, shape);
  }
}

template <class RK, class F, int LOGN, int K>
hipError_t launch_blind_rotate_with(hipStream_t s, const PbsParams& P, const typename F::elem* tw, const u32* lwe_in,
                                    size_t batch, const u32* tv, size_t tv_stride, const typename F::elem* bsk,
                                    u32* glwe_out, u32* lwe_extracted, u32* state, const launch::SideStream* side) {
  {
    using C = TeamCfg<F, LOGN, K>;
    auto kern = RK::get();
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(kern, RK::kLds, lds_done);
    if (e != hipSuccess) return e;
    unsigned capacity = 0;
    if ((e = resident_teams<RK>(&capacity)) != hipSuccess) return e;
    // a second stream cannot be forked inside a stream capture that the caller ends on `s` alone without it joining;
    // it does join (below), but a capture is no place for a measured policy: one stream there
    bool have_side = side && side->stream && side->fork && side->join;
    if (have_side) {
      hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(s, &capturing) != hipSuccess || capturing != hipStreamCaptureStatusNone) have_side = false;
    }
    const size_t key_bytes = (size_t)P.n * (K + 1) * P.levels * (K + 1) * F::kParts * C::N * 8;
    const BlindRotatePlan plan =
        blind_rotate_plan<F>(batch, state != nullptr, have_side, P.n, key_bytes, (u32)LOGN, (size_t)capacity * RK::kSamples);
    const int parts = plan.streams == 2 && plan.segments > 1 ? 2 : 1;
    const u32 per = (P.n + plan.segments - 1) / plan.segments;
    if (parts == 2) {  // fork: the side stream starts after everything already on s
      if ((e = hipEventRecord(side->fork, s)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(side->stream, side->fork, 0)) != hipSuccess) return e;
    }
    for (size_t off = 0; off < batch; off += plan.chunk) {
      const size_t here = batch - off < plan.chunk ? batch - off : plan.chunk;
      // part q = samples [q * share, (q + 1) * share) of the chunk; share is a multiple of the samples per team
      const size_t share = (((here + parts - 1) / parts + RK::kSamples - 1) / RK::kSamples) * RK::kSamples;
      for (u32 i0 = 0; i0 < P.n; i0 += per) {
        const u32 i1 = i0 + per < P.n ? i0 + per : P.n;
        for (int q = 0; q < parts; ++q) {
          if ((size_t)q * share >= here) break;
          const size_t o = off + (size_t)q * share;
          const size_t cnt = here - (size_t)q * share < share ? here - (size_t)q * share : share;
          hipLaunchKernelGGL(kern, dim3((unsigned)((cnt + RK::kSamples - 1) / RK::kSamples)), dim3(RK::kThreads), RK::kLds,
                             q ? side->stream : s, P, tw, lwe_in + o * ((size_t)P.n + 1), cnt, tv + o * tv_stride,
                             tv_stride, bsk, glwe_out ? glwe_out + o * (size_t)(K + 1) * C::N : nullptr,
                             lwe_extracted ? lwe_extracted + o * ((size_t)K * C::N + 1) : nullptr, i0, i1,
                             state ? state + o * (size_t)(K + 1) * C::N : nullptr);
          e = hipGetLastError();
          if (e != hipSuccess) return e;
        }
      }
    }
    if (parts == 2) {  // join: whatever follows on s waits for the side stream
      if ((e = hipEventRecord(side->join, side->stream)) != hipSuccess) return e;
      if ((e = hipStreamWaitEvent(s, side->join, 0)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
}

template <class F, int LOGN, int K>
hipError_t launch_blind_rotate(hipStream_t s, const PbsParams& P, const void* tw_v, const u32* lwe_in,
                               size_t batch, const u32* tv, size_t tv_stride, const void* bsk_v,
                               u32* glwe_out, u32* lwe_extracted, u32* state, const launch::SideStream* side, int shape) {
  if constexpr (!field_shape_ok<F, LOGN>()) {
    return hipErrorInvalidValue;  // the context never picks such a field (capi.cpp)
  } else {
    auto tw = static_cast<const typename F::elem*>(tw_v);
    auto bsk = static_cast<const typename F::elem*>(bsk_v);
    if constexpr (wide_shape_ok<F, LOGN, K>()) {
      if (batch <= wide_max_batch<F, LOGN, K>(P, shape))
        return launch_blind_rotate_wide<F, LOGN, K>(s, P, tw, lwe_in, batch, tv, tv_stride, bsk, glwe_out, lwe_extracted);
    }
    if (use_pair_kernel<F, LOGN, K>(batch))
      return launch_blind_rotate_with<PairOrTeamKernel<F, LOGN, K>, F, LOGN, K>(s, P, tw, lwe_in, batch, tv, tv_stride, bsk, glwe_out,
                                                                               lwe_extracted, state, side);
    return launch_blind_rotate_with<RotateKernel<F, LOGN, K>, F, LOGN, K>(s, P, tw, lwe_in, batch, tv, tv_stride, bsk, glwe_out,
                                                                          lwe_extracted, state, side);
  }
}

template <class F, int LOGN, int K>
hipError_t launch_blind_rotate_bmmp(hipStream_t s, const PbsParams& P, const void* tw_v, const u32* lwe_in,
                                    size_t batch, const u32* tv, size_t tv_stride, const void* bsk_v,
                                    u32* glwe_out, u32* lwe_extracted) {
  // (TFHE_BMMP_FFT: A/B builds only -- the unrolled rotation in the complex transform, measured and not offered:
  // profiles/r04_kernel_ab.txt)
#ifndef TFHE_BMMP_FFT
#define TFHE_BMMP_FFT 0
#endif
  if constexpr (LOGN != 9 || !field_shape_ok<F, LOGN>() ||
                !(F::kId == GlField::kId || F::kId == Fp49Field::kId || (TFHE_BMMP_FFT && F::kId == FftField::kId))) {
    return hipErrorInvalidValue;  // shape_supported_bmmp() / field_supported_bmmp() keep callers away
  } else {
    using C = TeamCfg<F, LOGN, K, 1>;
    auto tw = static_cast<const typename F::elem*>(tw_v);
    auto bsk = static_cast<const typename F::elem*>(bsk_v);
    auto kern = blind_rotate_bmmp_kernel<F, LOGN, K>;
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(kern, C::kLds, lds_done);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(C::kThreads), C::kLds, s, P, tw, lwe_in, batch,
                       tv, tv_stride, bsk, glwe_out, lwe_extracted);
    return hipGetLastError();
  }
}

template <class F, int LOGN, int K>
hipError_t launch_external_product(hipStream_t s, const PbsParams& P, const void* tw_v,
                                   const void* ggsw_v, size_t ggsw_stride_words, const u32* glwe_in,
                                   u32* ct1_inout, const u32* cmux_ct0, size_t batch, u32* glwe_out,
                                   unsigned long long* queue) {
  if constexpr (!field_shape_ok<F, LOGN>()) {
    return hipErrorInvalidValue;
  } else {
    using C = TeamCfg<F, LOGN, K, 1>;
    constexpr size_t kLdsWithTicket = C::kLds + 16;
    auto tw = static_cast<const typename F::elem*>(tw_v);
    auto ggsw = static_cast<const typename F::elem*>(ggsw_v);
    auto kern = external_product_kernel<F, LOGN, K>;
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(kern, kLdsWithTicket, lds_done);
    if (e != hipSuccess) return e;
    // persistent grid: as many teams as the device keeps resident at once (LDS or registers decide)
    static std::atomic<int> resident_per_device[kMaxDevices];
    const int dev_slot = current_device_slot();
    std::atomic<int>& resident = resident_per_device[dev_slot];
    int teams = resident.load(std::memory_order_relaxed);
    if (teams == 0) {
      int dev = dev_slot, cus = 0, per_cu = 0;
      e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (e == hipSuccess)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), C::kThreads, kLdsWithTicket);
      if (e != hipSuccess) return e;
      teams = cus * (per_cu > 0 ? per_cu : 1);
      resident.store(teams, std::memory_order_relaxed);
    }
    const size_t grid = batch < (size_t)teams ? batch : (size_t)teams;
    // the work queue pays from 16 samples per team (see the kernel)
    unsigned long long* use_queue = batch / grid >= 16 ? queue : nullptr;
    if (use_queue && (e = hipMemsetAsync(use_queue, 0, sizeof(unsigned long long), s)) != hipSuccess) return e;
    // the stride arrives in 8-byte words; the kernel counts elements
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::kThreads), kLdsWithTicket, s, P, tw, ggsw,
                       ggsw_stride_words / (sizeof(typename F::elem) / 8), glwe_in, ct1_inout, cmux_ct0, batch, glwe_out, use_queue);
    return hipGetLastError();
  }
}

template <class F, int LOGN>
hipError_t launch_bsk_prepare(hipStream_t s, const void* tw_v, const u32* polys, size_t poly_count,
                              void* spectra_v, u32 k) {
  if constexpr (!field_shape_ok<F, LOGN>()) {
    return hipErrorInvalidValue;
  } else {
    constexpr int N = 1 << LOGN;
    constexpr int G = GroupOf<F, LOGN>::value;
    constexpr int groups = 4 / G;  // polynomials per 256-thread workgroup
    const size_t lds = twiddle_bytes<F, LOGN>() + (size_t)N * 8 * groups;
    auto tw = static_cast<const typename F::elem*>(tw_v);
    auto spectra = static_cast<typename F::elem*>(spectra_v);
    auto kern = bsk_prepare_kernel<F, LOGN>;
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(kern, lds, lds_done);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((poly_count + groups - 1) / groups);
    const int layout_e = k == 1 ? key_layout_e<F, LOGN, 1>() : k == 2 ? key_layout_e<F, LOGN, 2>() : 0;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, tw, polys, poly_count, spectra,
                       F::n_inv(LOGN - F::kLogShrink), layout_e);
    return hipGetLastError();
  }
}

template <class F, int LOGN>
hipError_t launch_glwe_body(hipStream_t s, const void* tw_v, u32 k, const u32* rows, size_t row_count,
                            const u32* sk, u32* dst, size_t dst_stride, u32 negate) {
  if constexpr (!field_shape_ok<F, LOGN>()) {
    return hipErrorInvalidValue;
  } else {
    constexpr int N = 1 << LOGN;
    constexpr int G = GroupOf<F, LOGN>::value;
    constexpr int groups = 4 / G;  // rows per 256-thread workgroup
    const size_t lds = twiddle_bytes<F, LOGN>() + (size_t)N * 8 * groups;
    auto tw = static_cast<const typename F::elem*>(tw_v);
    auto kern = glwe_body_kernel<F, LOGN>;
    static std::atomic<unsigned long long> lds_done{0};
    hipError_t e = allow_lds(kern, lds, lds_done);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((row_count + groups - 1) / groups);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, tw, k, rows, row_count, sk, dst, dst_stride,
                       negate, F::n_inv(LOGN - F::kLogShrink));
    return hipGetLastError();
  }
}

}  // namespace

namespace launch {

bool shape_supported(u32 log_n, u32 k) { return log_n >= 9 && log_n <= 11 && (k == 1 || k == 2); }

bool shape_supported_bmmp(u32 log_n, u32 k) { return log_n == 9 && (k == 1 || k == 2); }

// The unrolled blind rotation carries three accumulator sets per wave.  Offered where that fits the registers and the
// mode is at least even with the loop: Goldilocks (+10 % at the reference's default parameters) and the single-spectrum
// 49-bit field (-5 %, 17-23 spilled registers).  In the two-spectra fields it ran 1.9-3.4x slower than the loop on 50-172
// spilled registers (profiles/r02_kernel_ab.txt, r02_h_isa_resources_all_kernels.txt): not instantiated, refused at load.
bool field_supported_bmmp(int field) { return field == kFieldGoldilocks || field == kFieldFp49 || (TFHE_BMMP_FFT && field == kFieldFft); }

int field_parts(int field) { return (field == kFieldGoldilocks || field == kFieldFp49) ? 1 : 2; }

bool field_shape_supported(int field, u32 log_n) {
  return field != kFieldFft || log_n == 9 || log_n == 10 || (TFHE_FFT_N2048 && log_n == 11);
}

int samples_per_team(int field, u32 log_n, u32 k) {
  if (field != kFieldFft) return 1;
  if (log_n == 11) return k == 2 ? SamplesPerTeam<FftField, 11, 2>::value : SamplesPerTeam<FftField, 11, 1>::value;
  if (log_n == 9) return k == 2 ? SamplesPerTeam<FftField, 9, 2>::value : SamplesPerTeam<FftField, 9, 1>::value;
  return 1;
}

// (field, log_n, k) -> template instantiation.  TFHE_DEV_CFG2_ONLY builds just the BASELINE cfg2
// shape (N = 1024, k = 1) for fast iteration on the kernels.
#if defined(TFHE_DEV_CFG2_ONLY)
#ifndef TFHE_DEV_LOGN
#define TFHE_DEV_LOGN 10
#define TFHE_DEV_K 1
#endif
#define TFHE_DISPATCH_LOGN_K(log_n, k, CALL)                                  \
  if ((k) == TFHE_DEV_K && (log_n) == TFHE_DEV_LOGN) {                        \
    constexpr int KK = TFHE_DEV_K;                                            \
    constexpr int LL = TFHE_DEV_LOGN;                                         \
    return CALL;                                                              \
  }
#define TFHE_DISPATCH_LOGN(log_n, CALL)                                       \
  if ((log_n) == TFHE_DEV_LOGN) {                                             \
    constexpr int LL = TFHE_DEV_LOGN;                                         \
    return CALL;                                                              \
  }
#else
#define TFHE_DISPATCH_LOGN_K(log_n, k, CALL)                                  \
  if ((k) == 1) {                                                             \
    constexpr int KK = 1;                                                     \
    switch (log_n) {                                                          \
      case 9: { constexpr int LL = 9; return CALL; }                          \
      case 10: { constexpr int LL = 10; return CALL; }                        \
      case 11: { constexpr int LL = 11; return CALL; }                        \
      default: break;                                                         \
    }                                                                         \
  } else if ((k) == 2) {                                                      \
    constexpr int KK = 2;                                                     \
    switch (log_n) {                                                          \
      case 9: { constexpr int LL = 9; return CALL; }                          \
      case 10: { constexpr int LL = 10; return CALL; }                        \
      case 11: { constexpr int LL = 11; return CALL; }                        \
      default: break;                                                         \
    }                                                                         \
  }
#define TFHE_DISPATCH_LOGN(log_n, CALL)                                       \
  switch (log_n) {                                                            \
    case 9: { constexpr int LL = 9; return CALL; }                            \
    case 10: { constexpr int LL = 10; return CALL; }                          \
    case 11: { constexpr int LL = 11; return CALL; }                          \
    default: break;                                                           \
  }
#endif

#if defined(TFHE_DEV_FIELD_FP_ONLY) || defined(TFHE_DEV_FIELD_FP49_ONLY) || defined(TFHE_DEV_FIELD_FFT_ONLY)  // dev builds: one field (fast iteration)
#if defined(TFHE_DEV_FIELD_FP49_ONLY)
#define TFHE_DEV_FIELD_ID kFieldFp49
#define TFHE_DEV_FIELD_T Fp49Field
#elif defined(TFHE_DEV_FIELD_FFT_ONLY)
#define TFHE_DEV_FIELD_ID kFieldFft
#define TFHE_DEV_FIELD_T FftField
#else
#define TFHE_DEV_FIELD_ID kFieldFp64
#define TFHE_DEV_FIELD_T FpField
#endif
#define TFHE_DISPATCH_FIELD(field, BODY)                                      \
  do {                                                                        \
    if ((field) == TFHE_DEV_FIELD_ID) {                                       \
      typedef TFHE_DEV_FIELD_T FF;                                            \
      BODY                                                                    \
    }                                                                         \
    return hipErrorInvalidValue;                                              \
  } while (0)
#else
#define TFHE_DISPATCH_FIELD(field, BODY)                                      \
  do {                                                                        \
    if ((field) == kFieldGoldilocks) {                                        \
      typedef GlField FF;                                                     \
      BODY                                                                    \
    } else if ((field) == kFieldFp64) {                                       \
      typedef FpField FF;                                                     \
      BODY                                                                    \
    } else if ((field) == kFieldGoldilocksSplit) {                            \
      typedef GlSplitField FF;                                                \
      BODY                                                                    \
    } else if ((field) == kFieldFp49) {                                       \
      typedef Fp49Field FF;                                                   \
      BODY                                                                    \
    } else if ((field) == kFieldFft) {                                        \
      typedef FftField FF;                                                    \
      BODY                                                                    \
    }                                                                         \
    return hipErrorInvalidValue;                                              \
  } while (0)
#endif

hipError_t bsk_prepare(hipStream_t s, int field, u32 log_n, u32 k, const void* tw, const u32* polys,
                       size_t poly_count, void* spectra) {
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN(log_n, (launch_bsk_prepare<FF, LL>(s, tw, polys, poly_count, spectra, k))));
}

hipError_t blind_rotate(hipStream_t s, int field, const PbsParams& P, const void* tw,
                        const u32* lwe_in, size_t batch, const u32* tv, size_t tv_stride,
                        const void* bsk, u32* glwe_out, u32* lwe_extracted, u32* state, const SideStream* side, int shape) {
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN_K(P.log_n, P.k,
                      (launch_blind_rotate<FF, LL, KK>(s, P, tw, lwe_in, batch, tv, tv_stride, bsk,
                                                       glwe_out, lwe_extracted, state, side, shape))));
}

hipError_t blind_rotate_plan(int field, const PbsParams& P, size_t batch, bool can_park, bool have_side,
                             BlindRotatePlanInfo* out, int shape) {
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN_K(P.log_n, P.k, (plan_blind_rotate<FF, LL, KK>(P, batch, can_park, have_side, out, shape))));
}

hipError_t blind_rotate_bmmp(hipStream_t s, int field, const PbsParams& P, const void* tw,
                             const u32* lwe_in, size_t batch, const u32* tv, size_t tv_stride,
                             const void* bsk, u32* glwe_out, u32* lwe_extracted) {
  if (!shape_supported_bmmp(P.log_n, P.k) || (P.n & 1u)) return hipErrorInvalidValue;
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN_K(P.log_n, P.k,
                      (launch_blind_rotate_bmmp<FF, LL, KK>(s, P, tw, lwe_in, batch, tv, tv_stride, bsk,
                                                            glwe_out, lwe_extracted))));
}

hipError_t external_product(hipStream_t s, int field, const PbsParams& P, const void* tw,
                            const void* ggsw, size_t ggsw_stride_words, const u32* glwe_in,
                            u32* ct1_inout, const u32* cmux_ct0, size_t batch, u32* glwe_out,
                            unsigned long long* queue) {
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN_K(P.log_n, P.k,
                      (launch_external_product<FF, LL, KK>(s, P, tw, ggsw, ggsw_stride_words, glwe_in,
                                                           ct1_inout, cmux_ct0, batch, glwe_out, queue))));
}

hipError_t key_switch(hipStream_t s, const KsParams& K, u32 big_n, u32 n, const u32* lwe_in,
                      size_t batch, const u32* ksk, u32* lwe_out) {
  const size_t lds = ((size_t)kKsWords * K.levels * kKsSamples + 2 * (size_t)kKsKeyRows * kKsCols) * sizeof(u32);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  dim3 grid((n + 1 + kKsCols - 1) / kKsCols, (unsigned)((batch + kKsSamples - 1) / kKsSamples));
  // aim at TFHE_KS_TARGET_WGS workgroups; a split covers a multiple of kKsWords words, at least 64
  unsigned splits = TFHE_KS_TARGET_WGS / (grid.x * grid.y);
  const unsigned max_splits = (big_n + 63u) / 64u;
  if (splits > max_splits) splits = max_splits;
  if (splits > 32u) splits = 32u;
  if (splits < 1u) splits = 1u;
  u32 words_per_split = ((big_n + splits - 1) / splits + kKsWords - 1) / kKsWords * kKsWords;
  splits = (big_n + words_per_split - 1) / words_per_split;
  grid.z = splits;
  if (splits > 1) {
    hipError_t e = hipMemsetAsync(lwe_out, 0, batch * ((size_t)n + 1) * sizeof(u32), s);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(key_switch_kernel, grid, dim3(256), lds, s, K, big_n, n, lwe_in, batch, ksk,
                     lwe_out, words_per_split);
  return hipGetLastError();
}

hipError_t decompose_words(hipStream_t s, u32 log_base, u32 levels, u32 fs, const u32* values,
                           size_t count, u32* digits) {
  const u32 ig = 32 - log_base * levels;
  hipLaunchKernelGGL(decompose_words_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, log_base,
                     levels, ig, fs, values, count, digits);
  return hipGetLastError();
}

hipError_t decompose_glwe(hipStream_t s, u32 log_base, u32 levels, u32 fs, u32 polys_per_ct,
                          u32 n_coeff, const u32* glwe, size_t batch, u32* digits) {
  const u32 ig = 32 - log_base * levels;
  const size_t total = batch * polys_per_ct * n_coeff;
  hipLaunchKernelGGL(decompose_glwe_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, log_base,
                     levels, ig, fs, polys_per_ct, n_coeff, glwe, total, digits);
  return hipGetLastError();
}

hipError_t switch_modulus(hipStream_t s, const u32* values, size_t count, u32 log_from, u32 log_to,
                          u32* out) {
  hipLaunchKernelGGL(switch_modulus_kernel, dim3(grid_for(count, 256)), dim3(256), 0, s, values,
                     count, log_from, log_to, out);
  return hipGetLastError();
}

hipError_t glwe_mul_monomial(hipStream_t s, u32 log_n, u32 polys_per_ct, const u32* glwe,
                             size_t batch, const i64* monomial_index, u32* out) {
  const size_t total = (batch * polys_per_ct) << log_n;
  hipLaunchKernelGGL(glwe_mul_monomial_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, log_n,
                     polys_per_ct, glwe, total, monomial_index, out);
  return hipGetLastError();
}

hipError_t sample_extract(hipStream_t s, u32 log_n, u32 k, const u32* glwe, size_t batch,
                          u32 sample_index, u32* lwe_out) {
  const size_t total = batch * (((size_t)k << log_n) + 1);
  hipLaunchKernelGGL(sample_extract_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, log_n, k,
                     glwe, batch, sample_index, lwe_out);
  return hipGetLastError();
}

hipError_t stream_copy(hipStream_t s, const void* src, void* dst, size_t bytes) {
  const size_t count = bytes / sizeof(copy_vec4);
  const size_t per_block = 256 * kCopyUnroll;
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)((count + per_block - 1) / per_block)), dim3(256), 0, s,
                     static_cast<const copy_vec4*>(src), static_cast<copy_vec4*>(dst), count);
  return hipGetLastError();
}

hipError_t fft_margin(double* worst, bool reset) {
#if defined(TFHE_FFT_TRACK_ERROR)
  unsigned long long bits = 0;
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(&bits, HIP_SYMBOL(g_fft_margin_bits), sizeof(bits));
  if (e != hipSuccess) return e;
  if (worst) __builtin_memcpy(worst, &bits, sizeof(bits));
  if (reset) {
    const unsigned long long zero = 0;
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_fft_margin_bits), &zero, sizeof(zero));
  }
  return e;
#else
  (void)worst;
  (void)reset;
  return hipErrorNotSupported;
#endif
}

hipError_t lwe_linear(hipStream_t s, u32 c0, const u32* ct0, u32 c1, const u32* ct1, size_t words,
                      u32* out, size_t words_per_ct, u32 b_add) {
  hipLaunchKernelGGL(lwe_linear_kernel, dim3(grid_for(words, 256)), dim3(256), 0, s, c0, ct0, c1, ct1,
                     words, words_per_ct, b_add, out);
  return hipGetLastError();
}

hipError_t glwe_body(hipStream_t s, int field, u32 log_n, const void* tw, u32 k, const u32* rows,
                     size_t row_count, const u32* sk, u32* dst, size_t dst_stride, bool negate) {
  if (row_count == 0) return hipSuccess;
  TFHE_DISPATCH_FIELD(field, TFHE_DISPATCH_LOGN(log_n, (launch_glwe_body<FF, LL>(s, tw, k, rows, row_count, sk, dst, dst_stride, negate ? 1u : 0u))));
}

hipError_t lwe_body(hipStream_t s, const u32* rows, size_t row_count, u32 n, const u32* sk,
                    const u32* plaintext, u32* dst, size_t dst_stride, bool negate) {
  if (row_count == 0) return hipSuccess;
  const unsigned grid = (unsigned)((row_count + 3) / 4);
  hipLaunchKernelGGL(lwe_body_kernel, dim3(grid), dim3(256), 0, s, rows, row_count, n, sk, plaintext,
                     dst, dst_stride, negate ? 1u : 0u);
  return hipGetLastError();
}

hipError_t ggsw_add_gadget(hipStream_t s, u32* ggsw, size_t ggsw_count, u32 k, u32 log_n, u32 levels,
                           u32 log_base, u32 gadget_top, const u32* messages) {
  const size_t total = ggsw_count * (size_t)(k + 1) * levels;
  hipLaunchKernelGGL(ggsw_add_gadget_kernel, dim3(grid_for(total, 256)), dim3(256), 0, s, ggsw,
                     ggsw_count, k, log_n, levels, log_base, gadget_top, messages);
  return hipGetLastError();
}

}  // namespace launch
}  // namespace tfhe

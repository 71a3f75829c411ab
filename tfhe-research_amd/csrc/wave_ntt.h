// wave_ntt.h -- negacyclic transform of one polynomial held by a GROUP of G 64-lane wavefronts
// (G = 1: one wave per polynomial, no barrier inside a transform; G = 4 for N = 2048 so that a lane
// holds only 8 elements per array and three waves fit a SIMD; G = 2 is kept and tested as well).
// The same passes serve the exact NTTs over the prime fields (8-byte elements, one per ring coefficient) and
// the complex FFT of field_fft.h (16-byte elements, one per PAIR of coefficients: LOGN below is then the log
// of the transform size N/2, and "psi" stands for the table of roots of i described there).
//
// N = 2^LOGN elements live in E = N/(64 G) registers per lane; below "tid" is the
// thread index inside the group (0 .. 64G-1) and TB = log2(64 G).  The transform is the
// merged-psi Cooley-Tukey NTT (natural order in, bit-reversed order out) and its Gentleman-Sande
// inverse, executed as three to five "register passes" (ceil(LOGN / e)).  A pass owns a window of e = log2(E) index bits:
// in window [LO, LO+e) a lane holds the E indices that differ only in those bits,
//     j = (hi << (LO+e)) | (r << LO) | lo,   lane = (hi << LO) | lo,   r = register number,
// so every butterfly of the stages on those bits is lane-local.  Between passes the polynomial
// is transposed through a wave-private LDS buffer (N u64) addressed with an XOR swizzle that
// makes all ds_write_b64 / ds_read_b64 of the transposes bank-conflict free on gfx950
// (tools/ntt_model.py proves layout and conflict-freedom for LOGN = 9, 10, 11).
//
//   forward : window [TB,TB+e) (j = r*64G + tid, coalesced) -> [TB-e,TB) -> [0,e)
//   inverse : the mirror image, ends in [TB,TB+e) again.
// (in the formulas above read "lane" as tid and 6 as TB.)  For G > 1 the transpose next to window
// [TB,TB+e) crosses waves and uses Ctx::poly_sync() (a workgroup barrier); the other one stays
// inside each wave's own part of the buffer and only needs Ctx::wave_sync() (see ntt_transpose).
//
// In window [0,e) position pos = tid*E + r of the bit-reversed-order spectrum sits in register r.
// The inverse is NOT scaled by N^-1: the bootstrapping key is pre-scaled instead (bsk_prepare).
//
// One twiddle table serves both directions: psi_rev[k] = psi^bitrev(k), and
// psi^-bitrev(h+i) = -psi_rev[2h-1-i], so the inverse butterfly is (V-U) * psi_rev[2h-1-i].
// The table has kTwiddleWords = N + 2 elements: [N], [N+1] hold psi_rev[1]*psi_rev[2] and
// psi_rev[1]*psi_rev[3] for the fused first two stages.
//
// The arithmetic is a policy class F (field_gl.h: Goldilocks u64; field_fp.h / field_fp49.h: 42- and 49-bit
// primes in fp64; field_fft.h: complex numbers); a polynomial takes 8 N bytes in every one of them, so the LDS
// budgets are identical; swizzles and the twiddle order are chosen per element size.
//
// Ctx (GPU: DeviceWave in kernels.hip; CPU tests: the SIMT emulator in tests/emu) provides
//   int tid() const;  void poly_sync() const;  void wave_sync() const;  elem* scratch() const;
//   const elem* twiddles() const;          working copy of the table (ntt_twiddle_slot order; LDS)
//   const elem* twiddles_uniform() const;  the natural-order table behind a wave-uniform pointer
// and, for the team code in pbs_wave.h, int exchange_buffers() const (1 or 2 LDS buffers per group)
// and Ctx with_exchange_buffer(int i) const (a copy whose scratch()/scratch_of() use buffer i).
#pragma once
#include <type_traits>

#include "field_fft.h"
#include "field_fp.h"
#include "field_fp49.h"
#include "field_gl.h"

namespace tfhe {

// elements of the twiddle table of a ring of degree n: psi_rev[n], the two products of the fused
// first two stages and the 16 coefficients of the fused third one (F::radix8_small_v)
constexpr int ntt_twiddle_words(int n) { return n + 18; }

// G = 0: HALF a wave per polynomial (32 lanes): two polynomials side by side in one wavefront, each in its own half --
// lanes 32 c .. 32 c + 31 hold polynomial c, the same instruction stream transforms both (pbs_wave.h: the pair kernel).
template <int LOGN, int G = 1>
struct NttShape {
  static_assert(G == 0 || G == 1 || G == 2 || G == 4, "waves per polynomial (0: half a wave)");
  static constexpr int kLogN = LOGN;
  static constexpr int kN = 1 << LOGN;
  static constexpr int kG = G;
  static constexpr int kThreads = G == 0 ? 32 : 64 * G;
  static constexpr int kTBits = (G == 0) ? 5 : (G == 1) ? 6 : (G == 2) ? 7 : 8;
  static constexpr int kEBits = LOGN - kTBits;
  static constexpr int kE = 1 << kEBits;
  static_assert(kEBits >= 2 && kEBits <= 5, "4..32 elements per lane (4: the complex transform at N = 512, four register passes)");
  // a register pass covers e index bits: three passes (most shapes), four (N = 2048 over 4 waves: e = 3; the
  // complex transform at N = 512: 256 points, e = 2) or five (the complex transform at N = 2048 over 4 waves:
  // 1024 points, e = 2); the last pass takes whatever low bits remain
  static constexpr int kPasses = (LOGN + kEBits - 1) / kEBits;
  static_assert(kPasses >= 3 && kPasses <= 5, "three to five register passes");
  // window lows of the passes (forward order): pass p works in window [lo(p), lo(p) + e)
  static constexpr int lo(int p) { return LOGN - p * kEBits > 0 ? LOGN - p * kEBits : 0; }
  static constexpr int kLo1 = kTBits;
  static constexpr int kLo2 = lo(2);
  static constexpr int kLo3 = lo(3);
  static constexpr int kLo4 = lo(4);
  static constexpr int kLo5 = 0;
};

// Where the working copy of the twiddle table (LDS on the GPU) keeps psi_rev[idx].  Stage b of a
// transform reads psi_rev[m + hi * cnt + i] with m = N >> (b+1), hi = tid >> LO (LO = low bit of the
// stage's register window), i < cnt = 2^(LO+e-b-1) the twiddle of register pair group i.  In table
// order the lanes of one read are cnt * 8 bytes apart -- a power of two, up to 8-way bank conflicts in
// the low windows (PMC round 1: a third of the LDS-active cycles were conflict cycles).  The working
// copy therefore stores the [H][cnt] block of every stage transposed, at m + i * H + hi with
// H = 2^(TB-LO): one read touches H consecutive 8-byte words (lanes with equal hi share a word), which
// is conflict free, and the address is one per-pass lane term plus an immediate.  The host-side table
// (F::fill_twiddles) stays in natural order; the permutation is applied while staging it
// (ntt_stage_twiddles).  Entries 0 and >= N (the fused-stage constants) stay where they are.
template <int LOGN, int G>
TFHE_HD int ntt_stage_window_lo(int b) {
  using S = NttShape<LOGN, G>;
  return b >= S::kLo1 ? S::kLo1 : b >= S::kLo2 ? S::kLo2 : b >= S::kLo3 ? S::kLo3 : b >= S::kLo4 ? S::kLo4 : S::kLo5;
}

#ifndef TFHE_TW_TRANSPOSED
#define TFHE_TW_TRANSPOSED 0
#endif
#ifndef TFHE_RADIX8
#define TFHE_RADIX8 1
#endif
// (the transposed layout costs the 8-byte fields their ds_read_b128 of two adjacent twiddles and measured 6 %
// SLOWER there, so they keep table order; the complex transform's 16-byte entries are one ds_read_b128 either
// way and it gains 1.2 % -- a third of its LDS-active cycles were conflict cycles: profiles/r02_kernel_ab.txt)
template <int ELEM_BYTES>
constexpr bool ntt_twiddles_transposed() {
  return TFHE_TW_TRANSPOSED || ELEM_BYTES == 16;
}

template <int LOGN, int G, bool TRANSPOSED = (TFHE_TW_TRANSPOSED != 0)>
TFHE_HD int ntt_twiddle_slot(int idx) {
  using S = NttShape<LOGN, G>;
  if (!TRANSPOSED || idx <= 0 || idx >= S::kN) return idx;
  const int fl = 31 - __builtin_clz((unsigned)idx);  // m = 2^fl
  const int b = LOGN - 1 - fl;
  const int lo = ntt_stage_window_lo<LOGN, G>(b);
  const int cnt_bits = lo + S::kEBits - b - 1;  // log2 of twiddles per lane at this stage
  const int h_bits = S::kTBits - lo;            // log2 of distinct lane terms
  const int local = idx - (1 << fl);
  const int hi = local >> cnt_bits, i = local & ((1 << cnt_bits) - 1);
  return (1 << fl) + (i << h_bits) + hi;
}

// copy the natural-order table `src` (ntt_twiddle_words(N) elements) into the working copy `dst`
// (eight loads in flight per thread: one load, wait, store per trip made the copy a chain of nine global
// round trips at the start of every team -- several microseconds, which the persistent external-product
// kernel pays once per four products at batch 4096)
// (W: entries to copy -- the whole table, or only psi_rev[0 .. N) for a field without fused-stage constants)
template <int LOGN, int G, class Elem, int W = ntt_twiddle_words(1 << LOGN)>
TFHE_HD void ntt_stage_twiddles(Elem* dst, const Elem* src, int tid, int nthreads) {
  constexpr int U = 8;
  for (int base = tid; base < W; base += U * nthreads) {
    Elem tmp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthreads;
      tmp[u] = src[i < W ? i : 0];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = base + u * nthreads;
      if (i < W) dst[ntt_twiddle_slot<LOGN, G, ntt_twiddles_transposed<(int)sizeof(Elem)>()>(i)] = tmp[u];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Lane-uniform constants of the TOP register window, fetched ahead of their use.
// In the top window hi = 0: every lane uses the same twiddles (and the same fused-stage constants), so
// on the GPU they are scalar operands (SGPRs) of the fp64 instructions.  Left to the compiler, their
// scalar loads sit inside the loops right in front of the first use -- the workgroup barriers forbid
// hoisting them -- and every `s_waitcnt lgkmcnt(0)` that follows also waits for whatever LDS reads of a
// transpose are in flight (scalar loads return out of order, so the counter has to drain): a dozen
// exposed round trips per wave and CMUX iteration (profiles/r02_*_isa_*).  TopConsts issues the loads
// explicitly, long before the pass that needs them (issue()), and waits once at a point where nothing
// else is outstanding (ready()).  The compiler does not know about the loads in flight; that is safe:
// an extra outstanding scalar load only makes the LDS waits it inserts itself more conservative, and no
// value of a block is used before ready() (the blocks are asm outputs of issue() and in/out operands of
// ready()).  The host emulator copies the values.
// ---------------------------------------------------------------------------------------------
template <class Elem, int CNT>
struct UniformBlock {
  static_assert(CNT == 2 || CNT == 4 || CNT == 8, "4, 8 or 16 dwords");
#if defined(__HIP_DEVICE_COMPILE__)
  typedef Elem vec __attribute__((ext_vector_type(CNT)));
  vec v;
  template <int FIRST>
  __device__ __forceinline__ void issue(const Elem* table) {
    if constexpr (CNT == 8)
      asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(table), "i"(FIRST * 8));
    else if constexpr (CNT == 4)
      asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(table), "i"(FIRST * 8));
    else
      asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(v) : "s"(table), "i"(FIRST * 8));
  }
  __device__ __forceinline__ void pin() { asm volatile("" : "+s"(v)); }
#else
  Elem v[CNT];
  template <int FIRST>
  void issue(const Elem* table) {
    for (int i = 0; i < CNT; ++i) v[i] = table[FIRST + i];
  }
  void pin() {}
#endif
  TFHE_HD Elem operator[](int i) const { return v[i]; }
};

TFHE_HD void uniform_wait() {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
}

// 8-byte word type an element is made of (one word, or re/im for the complex transform)
template <class Elem>
struct WordOf {
  typedef Elem type;
};
template <>
struct WordOf<cplx> {
  typedef double type;
};

// the top window's constants read straight from the table (the compiler places the loads)
template <class Elem>
struct TopFromTable {
  const Elem* twu;
  int n;
  TFHE_HD Elem tw(int i) const { return twu[i]; }
  TFHE_HD Elem fused(int i) const { return twu[n + i]; }
};

// ... prefetched.  FORWARD_SMALL: the forward transform of gadget digits, whose first two or three
// stages are fused in fields with kFuseFirstTwo (constants at [N ..]); otherwise plain stages only
// (any inverse transform; forward transforms of fields without fused stages).  Plain stage s of the
// window reads table elements [2^s, 2^(s+1)), s < e = log2(E).  The blocks are counted in 8-byte WORDS
// (an element is one word, or two for the complex transform): words [0, 4), [4, 8), [8, 16), ...
template <class F, int LOGN, int G, bool FORWARD_SMALL>
struct TopConsts {
  typedef typename F::elem elem;
  typedef typename WordOf<elem>::type word;
  static constexpr int W = (int)(sizeof(elem) / sizeof(word));  // words per element
  static constexpr int E = NttShape<LOGN, G>::kE;
  static constexpr int e = NttShape<LOGN, G>::kEBits;
  static constexpr int N = 1 << LOGN;
  static constexpr bool FUSE = FORWARD_SMALL && F::kFuseFirstTwo && e >= 2;
  static constexpr bool FUSE3 = TFHE_RADIX8 && FUSE && e >= 3;
  static_assert(!FUSE || W == 1, "fused stages: one-word elements");
  static constexpr int kFirstPlain = FUSE3 ? 8 : FUSE ? 4 : 1;  // lowest table index a plain stage reads
  static constexpr int kWords = E * W;                          // the window reads words [W, kWords)
  static_assert(kWords <= 32, "at most 32 words of top-window twiddles");
  UniformBlock<word, 4> lo;       // words [0, 4)
  UniformBlock<word, 4> mid;      // [4, 8)
  UniformBlock<word, 8> hi[3];    // [8, 16), [16, 24), [24, 32)
  UniformBlock<word, 8> fa, fb;   // fused-stage constants [N, N+8), [N+8, N+16)
  UniformBlock<word, 2> fc;       // [N+16, N+18)   (FUSE without the radix-8 step: fc holds [N, N+2))
  static constexpr bool kMid = kWords > 4 && kFirstPlain * W < 8;
  TFHE_HD void issue(const elem* table_e) {
    const word* table = reinterpret_cast<const word*>(table_e);
    lo.template issue<0>(table);
    if constexpr (kMid) mid.template issue<4>(table);
    if constexpr (kWords > 8) hi[0].template issue<8>(table);
    if constexpr (kWords > 16) {
      hi[1].template issue<16>(table);
      hi[2].template issue<24>(table);
    }
    if constexpr (FUSE3) {
      fa.template issue<N>(table);
      fb.template issue<N + 8>(table);
      fc.template issue<N + 16>(table);
    } else if constexpr (FUSE) {
      fc.template issue<N>(table);
    }
  }
  // before the first tw()/fused(): call where no LDS read is in flight that is not needed anyway
  TFHE_HD void ready() {
    uniform_wait();
    lo.pin();
    if constexpr (kMid) mid.pin();
    if constexpr (kWords > 8) hi[0].pin();
    if constexpr (kWords > 16) {
      hi[1].pin();
      hi[2].pin();
    }
    if constexpr (FUSE3) {
      fa.pin();
      fb.pin();
    }
    if constexpr (FUSE) fc.pin();
  }
  TFHE_HD word at(int w) const { return w < 4 ? lo[w] : w < 8 ? mid[w - 4] : hi[(w - 8) >> 3][(w - 8) & 7]; }
  TFHE_HD elem tw(int i) const {
    if constexpr (W == 1) {
      return at(i);
    } else {
      return elem{at(2 * i), at(2 * i + 1)};
    }
  }
  TFHE_HD elem fused(int i) const {
    if constexpr (W == 1) {
      if constexpr (FUSE3) return i < 8 ? fa[i] : i < 16 ? fb[i - 8] : fc[i - 16];
      return fc[i];
    } else {
      return F::zero();
    }
  }
};

// fields whose forward butterfly is one fused step (F::kFusedForwardButterfly + F::butterfly_forward)
template <class F, class = void>
struct FusedForwardButterfly {
  static constexpr bool value = false;
};
template <class F>
struct FusedForwardButterfly<F, decltype((void)F::kFusedForwardButterfly)> {
  static constexpr bool value = F::kFusedForwardButterfly;
};

// XOR swizzle of the transpose buffer (element = 8 bytes).  Each one makes every ds_write_b64 and
// ds_read_b64 of both transposes, in both directions, bank-conflict free (tools/ntt_model.py).
template <int LOGN, int G>
TFHE_HD int ntt_swizzle(int j) {
  // 256 16-byte elements over HALF a wave, 8 per lane (the pair kernel at N = 512): linear map of index bits 4..7 into the
  // 16-byte slot number, found by exhaustive search over such maps under the b128 banking rules restricted to a half's
  // lane groups: every read and the writes of windows [5,8) and [2,5) conflict free, the writes of window [0,3) 2-way
  // (16 LDS-array cycles against the ~14 the store's register transfer takes anyway)
  if (G == 0 && LOGN == 8) return j ^ (((j >> 4) & 1) * 1) ^ (((j >> 5) & 1) * 4) ^ (((j >> 7) & 1) * 10);
  if (G == 1 && LOGN == 10) return j ^ ((j >> 4) & 31);
  // 256 16-byte elements, 4 per lane (the complex transform at N = 512): best linear map of bits 4..7 into the
  // 16-byte slot number under the b128 banking rules -- every read and all but the last window's writes (2-way)
  // conflict free (tools/ntt_model.py rules; search in profiles/r02_kernel_ab.txt)
  if (G == 1 && LOGN == 8) return j ^ (((j >> 4) & 1) * 13) ^ (((j >> 5) & 1) * 4) ^ (((j >> 6) & 1) * 2);
  // 1024 16-byte elements over four waves, 4 per lane (the complex transform at N = 2048, emulator-tested, not
  // instantiated on the GPU): the same kind of map over index bits 4..9
  if (G == 4 && LOGN == 10)
    return j ^ (((j >> 4) & 1) * 13) ^ (((j >> 5) & 1) * 4) ^ (((j >> 6) & 1) * 2) ^ (((j >> 7) & 1) * 15) ^ (((j >> 8) & 1) * 2);
  if (G == 1 && LOGN == 9) return j ^ ((j >> 3) & 7) ^ (((j >> 6) & 3) << 3);
  if (G == 1 && LOGN == 11) return j ^ ((j >> 5) & 31);
  if (G == 2 && LOGN == 11) return j ^ ((j >> 1) & 7) ^ ((j >> 4) & 31);
  if (G == 4 && LOGN == 11) {
    // found by search over triangular XOR maps (tools/ntt_model.py::conflicts_grouped(11, 4))
    const int b4 = (j >> 4) & 1, b5 = (j >> 5) & 1, b6 = (j >> 6) & 1, b7 = (j >> 7) & 1;
    return j ^ b5 ^ ((b5 ^ b4) << 1) ^ ((b6 ^ b5) << 2) ^ ((b7 ^ b5 ^ b4) << 3) ^ (b7 << 4);
  }
  return j;  // correct for any shape, just not conflict-free
}

// element index held in register r of thread `tid` for window [LO, LO+e)
template <int LOGN, int G, int LO>
TFHE_HD int ntt_index(int tid, int r) {
  constexpr int e = NttShape<LOGN, G>::kEBits;
  return ((tid >> LO) << (LO + e)) | (r << LO) | (tid & ((1 << LO) - 1));
}

// memory position (in elements) of spectrum register r of thread `tid` inside one NTT-domain
// polynomial of the prepared bootstrapping key: pairs of registers are interleaved so that one
// global_load_dwordx4 per lane reads 64 x 16 B = 1 KiB contiguous per wave.
// (16-byte elements -- the complex transform -- are one global_load_dwordx4 each: register r of all
// threads is contiguous)
// LAYOUT_E: elements per lane of the shape the key was LAID OUT for.  A prepared key stores spectrum position pos (bit-reversed
// order; register r of thread tid holds pos = tid E + r at the end of a forward transform) at (pos mod E_l) (n / E_l) +
// pos / E_l, so that the kernel whose shape has E_l elements per lane reads register r of all its threads contiguously.  A
// kernel of another shape (E != E_l) reads the same key through the same formula: one context, one prepared key, whatever
// kernels run on it (pbs_wave.h::KeyLayout picks E_l per parameter set).
// (LAYOUT_E = 0: the reading kernel's own shape.)
template <int LOGN, int G, int ELEM_BYTES = 8, int LAYOUT_E = 0>
TFHE_HD int spectrum_slot(int tid, int r) {
  if (ELEM_BYTES == 16) {
    if (LAYOUT_E == 0 || LAYOUT_E == NttShape<LOGN, G>::kE) return r * NttShape<LOGN, G>::kThreads + tid;
    const int pos = tid * NttShape<LOGN, G>::kE + r;
    return (pos % LAYOUT_E) * ((1 << LOGN) / LAYOUT_E) + pos / LAYOUT_E;
  }
  return (r >> 1) * (2 * NttShape<LOGN, G>::kThreads) + tid * 2 + (r & 1);
}

// position of spectrum register r of thread `tid` in the group's LDS buffer when a transformed
// polynomial is handed to the other groups of the team: wave w of the group writes only inside
// [w*N/G, (w+1)*N/G), the same region its wave-local transposes use (see ntt_transpose), 64
// consecutive 8-byte words per register (conflict-free)
template <int LOGN, int G>
TFHE_HD int exchange_slot(int tid, int r) {
  if (G == 0) return r * 32 + tid;  // half a wave per polynomial: 32 consecutive 16-byte words per register
  return (tid >> 6) * (NttShape<LOGN, G>::kN / (G ? G : 1)) + r * 64 + (tid & 63);
}

// Transpose between two register windows through the group's LDS buffer.
// In a window with LO <= 6 a lane of wave w (= tid >> 6) holds only indices whose top log2(G) bits
// equal w, and the swizzles only permute low address bits, so its accesses stay inside the wave's
// own N/G-element region of the buffer; in the top window (LO = TB > 6) a lane touches every region.
//   both windows low : wave-local exchange, no workgroup barrier.
//   FROM is the top window (forward direction): the writes land in other waves' regions, so a
//     barrier first lets every wave finish reading its own region, a second one separates the
//     writes from the (own-region) reads.
//   TO is the top window (inverse direction): own-region writes, barrier, reads from every region,
//     and a barrier after them before anybody writes again.
// Invariant every user of the buffer keeps: reads that leave the own region (here, and the team's
// spectrum exchange in pbs_wave.h) are followed by a workgroup barrier before the next write.
// G == 1: everything is wave-local.
// SKIP_LEAD: the caller guarantees that a workgroup barrier already separates every earlier read of
// the buffer from this call (no read of it since), so the leading barrier of a cross-writing
// transpose is dropped.
// NS polynomials at once (ntt_transpose_multi): polynomial s goes through the buffer of context c[s]; the
// contexts differ only in the buffer they select, and the waits and barriers are shared -- all stores, one
// synchronisation, all loads -- so two polynomials cost one set of barriers (and one LDS round trip of latency).
// A transpose between ADJACENT windows exchanges register-index bit i with thread-index bit min(LO_FROM, LO_TO) + i.  Where
// those thread bits are lane bits 4 and 5 -- rows of 16 lanes and halves of the wave -- gfx950 does the exchange in
// registers: v_permlane16_swap_b32 a, b swaps the odd rows of a with the even rows of b, v_permlane32_swap_b32 the upper
// half of a with the lower half of b: exactly "the lanes whose bit is 1 hand their a over for the b of the lanes whose bit
// is 0", one instruction per dword pair and bit -- 16 VALU instructions for a two-bit window of four 16-byte elements
// instead of four stores and four loads through the LDS path (74 cycles of it, plus the round trip).  Taken where the
// kernel has VALU issue to spare and the LDS path is the busier pipe (TFHE_SWAP_TRANSPOSE: the two-bit windows of the
// complex transform at N = 512 with k = 2 and at N = 2048; profiles/r04_kernel_ab.txt).  The host emulator keeps the LDS
// path (same values).
#ifndef TFHE_SWAP_TRANSPOSE
#define TFHE_SWAP_TRANSPOSE 1
#endif
// Which lane bits may be exchanged in registers (TFHE_SWAP_LOW_BIT: the lowest one; 6 = never): bits 5 and 4 by the permlane
// swaps (1 instruction per dword pair), bits 3 and 2 by bank-masked DPP moves -- a row of 16 lanes is four banks of 4, lane bit
// 2 picks the odd banks, lane bit 3 the upper two, so "the lanes whose bit is 0 take their partner's a into b" is ONE
// v_mov_b32_dpp whose bank mask leaves the other lanes' b alone (2 instructions per dword pair) --, bits 1 and 0 by quad
// permutes and selects (4 per pair: measured slower than LDS, profiles/r04_kernel_ab.txt, so the default stops at bit 2).
// Per shape (TFHE_SWAP_E8: also for the three-bit windows of 8 elements per lane): see swap_transpose_low_bit.
#ifndef TFHE_SWAP_LOW_BIT
#define TFHE_SWAP_LOW_BIT 2
#endif
#ifndef TFHE_SWAP_E8
#define TFHE_SWAP_E8 0
#endif
template <int LOGN, int G, int LO_FROM, int LO_TO>
constexpr int swap_transpose_low_bit() {  // lowest exchanged lane bit, or -1: through LDS
  constexpr int e = NttShape<LOGN, G>::kEBits;
  constexpr int lo = LO_FROM < LO_TO ? LO_FROM : LO_TO, hi = LO_FROM < LO_TO ? LO_TO : LO_FROM;
  // adjacent windows of lane bits only (thread bit = lane bit below 6; half-wave groups: below 5)
  if (!TFHE_SWAP_TRANSPOSE || hi - lo != e || lo + e > (G == 0 ? 5 : 6)) return -1;
  if (e == 3 && !TFHE_SWAP_E8) return -1;
  if (e != 2 && e != 3) return -1;
  return lo >= TFHE_SWAP_LOW_BIT ? lo : -1;
}
template <int LOGN, int G, int LO_FROM, int LO_TO>
constexpr bool swap_transpose_shape() {
  return swap_transpose_low_bit<LOGN, G, LO_FROM, LO_TO>() >= 0;
}
#if defined(__HIP_DEVICE_COMPILE__)
// exchange register-index bit RB with lane bit LB for all E registers of 16- or 8-byte elements (the words go through u32
// copies: a cast of an element's address to u32* would be an aliasing violation; the swaps and moves are the compiler's
// builtins, not inline assembly -- the hazard recogniser has to see them: issued as asm statements the swaps returned wrong
// words at N = 512)
template <int RB, int LB, int E, int W>
__device__ __forceinline__ void exchange_register_bit_with_lane_bit(u32 (&w)[E][W], int lane) {
#pragma unroll
  for (int r0 = 0; r0 < E; ++r0) {
    if ((r0 >> RB) & 1) continue;
    const int r1 = r0 | (1 << RB);
#pragma unroll
    for (int i = 0; i < W; ++i) {
      const u32 a = w[r0][i], b = w[r1][i];  // lanes whose bit is 0 hand over b for their partner's a
      if constexpr (LB == 5) {
        const auto sw = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        w[r0][i] = sw[0];
        w[r1][i] = sw[1];
      } else if constexpr (LB == 4) {
        const auto sw = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        w[r0][i] = sw[0];
        w[r1][i] = sw[1];
      } else if constexpr (LB == 3) {  // row_ror:8 = lane ^ 8; banks 0,1 have bit 3 clear
        w[r1][i] = (u32)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x128, 0xF, 0x3, false);
        w[r0][i] = (u32)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x128, 0xF, 0xC, false);
      } else if constexpr (LB == 2) {  // row_shl:4 reads lane + 4 (banks 0,2: bit 2 clear), row_shr:4 lane - 4
        w[r1][i] = (u32)__builtin_amdgcn_update_dpp((int)b, (int)a, 0x104, 0xF, 0x5, false);
        w[r0][i] = (u32)__builtin_amdgcn_update_dpp((int)a, (int)b, 0x114, 0xF, 0xA, false);
      } else {  // quad_perm:[1,0,3,2] = lane ^ 1, quad_perm:[2,3,0,1] = lane ^ 2, and selects
        constexpr int ctrl = LB == 0 ? 0xB1 : 0x4E;
        const bool bit = ((lane >> LB) & 1) != 0;
        const u32 pa = (u32)__builtin_amdgcn_update_dpp(0, (int)a, ctrl, 0xF, 0xF, true);
        const u32 pb = (u32)__builtin_amdgcn_update_dpp(0, (int)b, ctrl, 0xF, 0xF, true);
        w[r1][i] = bit ? b : pa;
        w[r0][i] = bit ? pb : a;
      }
    }
  }
}
template <class Elem, int E, int LOW>
__device__ __forceinline__ void swap_transpose(Elem (&x)[E], int lane) {
  constexpr int W = (int)(sizeof(Elem) / 4);
  u32 w[E][W];
#pragma unroll
  for (int r = 0; r < E; ++r) __builtin_memcpy(w[r], &x[r], sizeof(Elem));
  exchange_register_bit_with_lane_bit<0, LOW, E, W>(w, lane);
  exchange_register_bit_with_lane_bit<1, LOW + 1, E, W>(w, lane);
  if constexpr (E == 8) exchange_register_bit_with_lane_bit<2, LOW + 2, E, W>(w, lane);
#pragma unroll
  for (int r = 0; r < E; ++r) __builtin_memcpy(&x[r], w[r], sizeof(Elem));
}
#endif

// Issue priority by phase (TFHE_PHASE_PRIORITY).  A wave drops to priority 0 while it MOVES data -- while it issues the
// stores and the loads of a transpose through LDS here (the s_waitcnt for the loaded registers comes later, where a wave
// issues nothing anyway); in the shapes with 8 ring coefficients per lane also for the multiply-accumulate with its key
// loads and spectrum reads (pbs_wave.h) -- and runs its register passes, the digit chain and everything else at priority
// 2: of the waves of a SIMD the one that has arithmetic to issue goes first.
// Measured per shape and field (blind rotation of 4,096, ms; profiles/r04_kernel_ab.txt section 11):
//   16 coefficients per lane (N = 1024; the pair kernel): transposes low: complex transform cfg2 28.74 -> 27.95 (aligned
//      32.23 -> 31.92), cfg1's pair kernel 9.56 -> 9.28, fp64-p42 at cfg2 52.65 -> 50.6, Goldilocks 131.6 -> 130.1; the
//      multiply-accumulate low as well: worse or level (28.6; 50.7; 131.4) -- there it stays at 2;
//    8 coefficients per lane (N = 512, N = 2048): transposes low alone: cfg3 level, cfg5 +1 %, fp64-p49 at cfg3 -1.2 %,
//      Goldilocks level; multiply-accumulate low alone: -1.3 % / -0.4 %; BOTH: cfg3 50.16 -> 49.34, cfg5 169.2 -> 164.4,
//      fp64-p49 67.75 -> 65.45, Goldilocks 221.0 -> 205.0.
// (Raising the priority INSIDE the transposes, the operand read and the final update low too, and fixed different priorities
// per wave slot: level or worse.)
#ifndef TFHE_PHASE_PRIORITY
#define TFHE_PHASE_PRIORITY 1
#endif
template <class Elem, int E>
constexpr bool transpose_lowers_priority() {
  return TFHE_PHASE_PRIORITY != 0;
}
template <class Elem, int E>
constexpr bool mac_lowers_priority() {  // 8 ring coefficients per lane: 4 complex elements or 8 field elements
  return TFHE_PHASE_PRIORITY && E * (int)sizeof(Elem) == 64;
}

template <class F, int LOGN, int G, int LO_FROM, int LO_TO, bool SKIP_LEAD = false, int NS, class Ctx>
TFHE_HD void ntt_transpose_multi(const Ctx (&c)[NS], typename F::elem (&x)[NS][NttShape<LOGN, G>::kE]) {
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr bool PRIO = transpose_lowers_priority<typename F::elem, E>();
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (swap_transpose_shape<LOGN, G, LO_FROM, LO_TO>()) {
    constexpr int LOW = swap_transpose_low_bit<LOGN, G, LO_FROM, LO_TO>();
#pragma unroll
    for (int s = 0; s < NS; ++s) swap_transpose<typename F::elem, E, LOW>(x[s], c[0].tid());
    return;
  }
#endif
  constexpr bool WRITES_CROSS = G > 1 && LO_FROM > 6;
  constexpr bool READS_CROSS = G > 1 && LO_TO > 6;
  static_assert(!(WRITES_CROSS && READS_CROSS), "one of the two windows is a low one");
  const int tid = c[0].tid();
  // (TFHE_PROBE_NO_TRANSPOSE: dev_switches.h -- a WRONG-BITS timing probe, TFHE_DEV_BUILD only)
  if (TFHE_PROBE_NO_TRANSPOSE) return;
  if (WRITES_CROSS && !SKIP_LEAD) c[0].poly_sync();
  if constexpr (PRIO) wave_priority<0>();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    typename F::elem* buf = c[s].scratch();
#pragma unroll
    for (int r = 0; r < E; ++r) buf[ntt_swizzle<LOGN, G>(ntt_index<LOGN, G, LO_FROM>(tid, r))] = x[s][r];
  }
  if (WRITES_CROSS || READS_CROSS) c[0].poly_sync(); else c[0].wave_sync();
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const typename F::elem* buf = c[s].scratch();
#pragma unroll
    for (int r = 0; r < E; ++r) x[s][r] = buf[ntt_swizzle<LOGN, G>(ntt_index<LOGN, G, LO_TO>(tid, r))];
  }
  if (READS_CROSS) c[0].poly_sync(); else c[0].wave_sync();
  if constexpr (PRIO) wave_priority<2>();
}

template <class F, int LOGN, int G, int LO_FROM, int LO_TO, bool SKIP_LEAD = false, class Ctx>
TFHE_HD void ntt_transpose(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE]) {
  const Ctx cs[1] = {c};
  ntt_transpose_multi<F, LOGN, G, LO_FROM, LO_TO, SKIP_LEAD, 1>(
      cs, reinterpret_cast<typename F::elem (&)[1][NttShape<LOGN, G>::kE]>(x));
}

// ---------------------------------------------------------------------------------------------
// Twiddles of a LOW register window, fetched ahead of the pass that uses them.
// In the low windows every lane has its own twiddles (hi = tid >> LO differs), read from the working copy in LDS.
// Left to the compiler each ds_read sits right in front of its butterfly -- the wave-level fences of the
// transposes forbid hoisting it -- behind an s_waitcnt that exposes the LDS round trip: a dozen of them per pass
// in the ISA of round 2's kernel.  PassTwiddles reads all of a pass's twiddles (1 + 2 + 4 for a three-bit window)
// in one clump BEFORE the transpose that precedes the pass: the LDS pipe is in order per wave, so they arrive
// before the transposed elements do, and the pass starts with every operand in registers.  Costs their registers
// (28 VGPRs for 16-byte elements at e = 3) across the transpose, so a field opts in (F::kPreloadTwiddles): the
// complex transform has them, the 8-byte fields at 16 elements per lane do not.
// ---------------------------------------------------------------------------------------------
#ifndef TFHE_PRELOAD_TWIDDLES
#define TFHE_PRELOAD_TWIDDLES 1
#endif
template <class F, class = void>
struct PreloadsTwiddles {
  static constexpr bool value = false;
};
template <class F>
struct PreloadsTwiddles<F, decltype((void)F::kPreloadTwiddles)> {
  static constexpr bool value = F::kPreloadTwiddles && TFHE_PRELOAD_TWIDDLES != 0;
};

// position in the working copy of the twiddle that stage b of window [LO, LO+e) uses for register pair group i
template <class F, int LOGN, int G, int LO, bool INVERSE>
TFHE_HD int ntt_low_twiddle_slot(int tid, int b, int i) {
  typedef typename F::elem elem;
  constexpr int e = NttShape<LOGN, G>::kEBits;
  constexpr int H = 1 << (NttShape<LOGN, G>::kTBits - LO);
  const int hi = tid >> LO;
  const int m = NttShape<LOGN, G>::kN >> (b + 1);
  const int cnt = 1 << (LO + e - b - 1);
  constexpr bool TR = ntt_twiddles_transposed<(int)sizeof(elem)>();
  if (!INVERSE || F::kLogShrink) return TR ? m + i * H + hi : m + hi * cnt + i;  // node m + hi cnt + i
  // prime fields, inverse: psi^-bitrev(m+j) = -psi_rev[2m-1-j]
  return TR ? m + (cnt - 1 - i) * H + (H - 1 - hi) : 2 * m - 1 - hi * cnt - i;
}

template <class F, int LOGN, int G, int LO, int BHI, int BLO, bool INVERSE>
struct PassTwiddles {
  typedef typename F::elem elem;
  static constexpr int E = NttShape<LOGN, G>::kE;
  static constexpr int count_from(int b) { return b > BHI ? 0 : (E >> (b - LO + 1)) + count_from(b + 1); }
  static constexpr int kCount = count_from(BLO);
  // stage b's twiddles start behind those of the stages above it (b + 1 .. BHI)
  static constexpr int offset(int b) { return count_from(b + 1); }
  elem w[kCount];
  template <class Ctx>
  TFHE_HD void load(const Ctx& c) {
    const elem* tw = c.twiddles();
    const int tid = c.tid();
#pragma unroll
    for (int b = BHI; b >= BLO; --b)
#pragma unroll
      for (int i = 0; i < (E >> (b - LO + 1)); ++i) w[offset(b) + i] = tw[ntt_low_twiddle_slot<F, LOGN, G, LO, INVERSE>(tid, b, i)];
  }
  TFHE_HD elem get(int b, int i) const { return w[offset(b) + i]; }
};
// the pass reads its twiddles itself, one by one (fields that do not preload, and every top window)
struct NoPassTwiddles {};

// forward stages on bits BHI..BLO (descending) of window [LO, LO+e).  SMALL_FIRST: the inputs of
// the first stage handled here are small integers (gadget digits), so its twiddle products may
// use F::mul_small (exact without reduction in the fp64 field).
template <class F, int LOGN, int G, int LO, int BHI, int BLO, bool SMALL_FIRST, class Ctx, class Top, class Pre = NoPassTwiddles>
TFHE_HD void ntt_pass_forward(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE], const Top& top,
                              const Pre& pre = Pre{}) {
  constexpr bool PRE = !std::is_base_of<NoPassTwiddles, Pre>::value;
  typedef typename F::elem elem;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int e = NttShape<LOGN, G>::kEBits;
  constexpr int N = NttShape<LOGN, G>::kN;
  constexpr int H = 1 << (NttShape<LOGN, G>::kTBits - LO);  // distinct lane terms (ntt_twiddle_slot)
  const elem* tw = c.twiddles();
  // Top window (LO = TB): hi = 0, every lane uses the same twiddles.  They are read from the
  // natural-order table through a wave-uniform pointer (global memory on the GPU: scalar loads into
  // SGPRs -- no LDS reads, no vector registers; the fp64 instructions take them as their one scalar
  // operand), the lower windows from the working copy (LDS).
  constexpr bool TOP = LO == NttShape<LOGN, G>::kTBits;
  const int hi = c.tid() >> LO;
  // The top stages of the whole transform on small inputs (gadget digits): in a field with
  // kFuseFirstTwo the first two collapse into one exact radix-4 step without any modular reduction
  // (F::radix4_small; twiddles psi_rev[1..3] and their products at [N], [N+1]), and the third one
  // takes its multiplied leg w3 * v straight from the four small inputs of v's radix-4 group with the
  // 16 pre-multiplied coefficients at [N+2 .. N+17] (F::radix8_small_v): 34 instructions per eight
  // elements instead of 52.  (The top window's twiddles are the same for every lane: hi = 0.)
  constexpr bool FUSE = SMALL_FIRST && F::kFuseFirstTwo && BHI == LOGN - 1 && BHI - BLO >= 1;
  constexpr bool FUSE3 = TFHE_RADIX8 && FUSE && BHI - BLO >= 2;
  if constexpr (FUSE) {
    constexpr int s1 = 1 << (BHI - LO), s2 = s1 >> 1, s3 = s2 >> 1;
    static_assert(!FUSE || TOP, "the fused stages are the top ones");
    const elem w1 = top.tw(1), w2a = top.tw(2), w2b = top.tw(3), w12a = top.fused(0), w12b = top.fused(1);
    if constexpr (FUSE3) {
#pragma unroll
      for (int r = 0; r < s3; ++r) {
        const elem va = x[r + s3], vb = x[r + s3 + s2], vc = x[r + s3 + s1], vd = x[r + s3 + s1 + s2];
        F::radix4_small(x[r], x[r + s2], x[r + s1], x[r + s1 + s2], w1, w2a, w2b, w12a, w12b);
        constexpr int pos[4] = {0, s2, s1, s1 + s2};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const elem z = F::radix8_small_v(va, vb, vc, vd, top.fused(2 + 4 * q), top.fused(3 + 4 * q),
                                           top.fused(4 + 4 * q), top.fused(5 + 4 * q));
          const elem u = x[r + pos[q]];
          x[r + pos[q]] = F::add(u, z);
          x[r + pos[q] + s3] = F::sub(u, z);
        }
      }
    } else {
#pragma unroll
      for (int r = 0; r < s2; ++r)
        F::radix4_small(x[r], x[r + s2], x[r + s1], x[r + s1 + s2], w1, w2a, w2b, w12a, w12b);
    }
  }
#pragma unroll
  for (int b = FUSE3 ? BHI - 3 : FUSE ? BHI - 2 : BHI; b >= BLO; --b) {
    const int rb = b - LO;
    const int m = N >> (b + 1);
#pragma unroll
    for (int r0 = 0; r0 < E; ++r0) {
      if ((r0 >> rb) & 1) continue;
      const int r1 = r0 | (1 << rb);
      elem w;
      if constexpr (TOP) w = top.tw(m + (r0 >> (rb + 1)));
      else if constexpr (PRE) w = pre.get(b, r0 >> (rb + 1));
      else w = ntt_twiddles_transposed<(int)sizeof(elem)>() ? tw[m + (r0 >> (rb + 1)) * H + hi]
                                                            : tw[m + (hi << (LO + e - b - 1)) + (r0 >> (rb + 1))];
      const elem u = x[r0];
      if constexpr (FusedForwardButterfly<F>::value) {
        // (u + w x1, u - w x1) in six fused multiply-adds instead of a product and two sums (field_fft.h)
        F::butterfly_forward(u, x[r1], w, x[r0], x[r1]);
      } else {
        const elem v = (SMALL_FIRST && b == BHI) ? F::mul_small(x[r1], w) : F::mul(x[r1], w);
        x[r0] = F::add(u, v);
        x[r1] = F::sub(u, v);
      }
    }
  }
}

// inverse stages on bits BLO..BHI (ascending) of window [LO, LO+e)
template <class F, int LOGN, int G, int LO, int BHI, int BLO, class Ctx, class Top, class Pre = NoPassTwiddles>
TFHE_HD void ntt_pass_inverse(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE], const Top& top,
                              const Pre& pre = Pre{}) {
  constexpr bool PRE = !std::is_base_of<NoPassTwiddles, Pre>::value;
  typedef typename F::elem elem;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int e = NttShape<LOGN, G>::kEBits;
  constexpr int H = 1 << (NttShape<LOGN, G>::kTBits - LO);
  constexpr bool TOP = LO == NttShape<LOGN, G>::kTBits;  // lane-uniform twiddles: see ntt_pass_forward
  const elem* tw = c.twiddles();
  const int hi = c.tid() >> LO;
#pragma unroll
  for (int b = BLO; b <= BHI; ++b) {
    const int rb = b - LO;
    const int h = NttShape<LOGN, G>::kN >> (b + 1);
    // psi^-bitrev(h+j) = -psi_rev[2h-1-j]: table index h + (H-1-hi) * cnt + (cnt-1-i) for register
    // pair group i, which the working copy keeps at h + (cnt-1-i) * H + (H-1-hi) (ntt_twiddle_slot)
    const int cnt = 1 << (LO + e - b - 1);
#pragma unroll
    for (int r0 = 0; r0 < E; ++r0) {
      if ((r0 >> rb) & 1) continue;
      const int r1 = r0 | (1 << rb);
      // (F::inverse_twiddle_index(h, i) = 2h - 1 - i in the prime fields, h + i for the complex transform,
      // whose mul_inverse conjugates the entry)
      elem w;
      if constexpr (TOP) w = top.tw(F::inverse_twiddle_index(h, r0 >> (rb + 1)));
      else if constexpr (PRE) w = pre.get(b, r0 >> (rb + 1));
      else w = ntt_twiddles_transposed<(int)sizeof(elem)>() ? (F::kLogShrink ? tw[h + (r0 >> (rb + 1)) * H + hi]  // node h + hi cnt + i
                                                           : tw[h + (cnt - 1 - (r0 >> (rb + 1))) * H + (H - 1 - hi)])
                                          : (F::kLogShrink ? tw[h + (hi << (LO + e - b - 1)) + (r0 >> (rb + 1))]
                                                           : tw[2 * h - 1 - (hi << (LO + e - b - 1)) - (r0 >> (rb + 1))]);
      const elem u = x[r0];
      const elem v = x[r1];
      if constexpr (FusedForwardButterfly<F>::value) {
        F::butterfly_inverse(u, v, w, x[r0], x[r1]);  // the same values without a negated result (field_fft.h)
      } else {
        x[r0] = F::add(u, v);
        x[r1] = F::mul_inverse(F::sub(v, u), w);
      }
    }
    // fields with little lazy headroom (F::kInverseSweepEvery > 0): the un-multiplied leg doubles
    // per stage, so everything is brought back to |.| <= p/2 after every kInverseSweepEvery-th stage
    // (stage number = b + 1; nothing to do after the last one, finish() reduces)
    if (F::kInverseSweepEvery > 0 && (b + 1) % (F::kInverseSweepEvery > 0 ? F::kInverseSweepEvery : 1) == 0 &&
        b + 1 < LOGN) {
#pragma unroll
      for (int r = 0; r < E; ++r) x[r] = F::reduce(x[r]);
    }
  }
}

// in: x[r] = a[r*64G + tid].  out: x[r] = A_bitrev[tid*E + r].
// SMALL_INPUT: every |x[r]| <= 2^F::kSmallBits on entry (gadget digits).
// AFTER_BARRIER: nobody has read the buffer since the last workgroup barrier (see ntt_transpose).
// top: where the top window's lane-uniform constants come from (TopFromTable, or a TopConsts the
// caller has issued earlier and on which ready() has been called).
// ntt_forward_multi / ntt_inverse_multi transform NS polynomials in step (polynomial s through the buffer of c[s]):
// every register pass is run for each of them in turn, the transposes are shared (ntt_transpose_multi), and a low
// window's twiddles are fetched once for all of them.
template <class F, int LOGN, int G, int LO, int BHI, int BLO, bool SMALL, int NS, class Ctx, class Top, class Pre>
TFHE_HD void ntt_pass_forward_each(const Ctx (&c)[NS], typename F::elem (&x)[NS][NttShape<LOGN, G>::kE], const Top& top,
                                   const Pre& pre) {
#pragma unroll
  for (int s = 0; s < NS; ++s) ntt_pass_forward<F, LOGN, G, LO, BHI, BLO, SMALL>(c[s], x[s], top, pre);
}
template <class F, int LOGN, int G, int LO, int BHI, int BLO, int NS, class Ctx, class Top, class Pre>
TFHE_HD void ntt_pass_inverse_each(const Ctx (&c)[NS], typename F::elem (&x)[NS][NttShape<LOGN, G>::kE], const Top& top,
                                   const Pre& pre) {
#pragma unroll
  for (int s = 0; s < NS; ++s) ntt_pass_inverse<F, LOGN, G, LO, BHI, BLO>(c[s], x[s], top, pre);
}

// the twiddles of a low pass: fetched ahead (PassTwiddles) by the fields that opt in, otherwise read by the pass itself
template <class F, int LOGN, int G, int LO, int BHI, int BLO, bool INVERSE, bool PRE>
struct LowPassTwiddles : PassTwiddles<F, LOGN, G, LO, BHI, BLO, INVERSE> {};
template <class F, int LOGN, int G, int LO, int BHI, int BLO, bool INVERSE>
struct LowPassTwiddles<F, LOGN, G, LO, BHI, BLO, INVERSE, false> : NoPassTwiddles {
  template <class Ctx>
  TFHE_HD void load(const Ctx&) {}
};

// before_first_store(): called once, after the first register pass and before the first store into the buffer -- the
// place for a barrier that only has to precede the buffer's reuse (pbs_wave.h, TFHE_LATE_BARRIER)
struct NothingBefore {
  TFHE_HD void operator()() const {}
};
template <class F, int LOGN, int G, bool SMALL_INPUT = false, bool AFTER_BARRIER = false, int NS, class Ctx, class Top,
          class Before = NothingBefore>
TFHE_HD void ntt_forward_multi(const Ctx (&c)[NS], typename F::elem (&x)[NS][NttShape<LOGN, G>::kE], const Top& top,
                               const Before& before_first_store = Before{}) {
  using S = NttShape<LOGN, G>;
  constexpr bool PRE = PreloadsTwiddles<F>::value;
  ntt_pass_forward_each<F, LOGN, G, S::kLo1, LOGN - 1, S::kTBits, SMALL_INPUT>(c, x, top, NoPassTwiddles{});
  // every low pass's twiddles are read before the transpose in front of it (PassTwiddles)
  LowPassTwiddles<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2, false, PRE> t2;
  t2.load(c[0]);
  before_first_store();
  ntt_transpose_multi<F, LOGN, G, S::kLo1, S::kLo2, AFTER_BARRIER>(c, x);
  ntt_pass_forward_each<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2, false>(c, x, top, t2);
  LowPassTwiddles<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3, false, PRE> t3;
  t3.load(c[0]);
  ntt_transpose_multi<F, LOGN, G, S::kLo2, S::kLo3>(c, x);
  ntt_pass_forward_each<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3, false>(c, x, top, t3);
  if constexpr (S::kPasses >= 4) {
    LowPassTwiddles<F, LOGN, G, S::kLo4, S::kLo3 - 1, S::kLo4, false, PRE> t4;
    t4.load(c[0]);
    ntt_transpose_multi<F, LOGN, G, S::kLo3, S::kLo4>(c, x);
    ntt_pass_forward_each<F, LOGN, G, S::kLo4, S::kLo3 - 1, S::kLo4, false>(c, x, top, t4);
  }
  if constexpr (S::kPasses == 5) {
    LowPassTwiddles<F, LOGN, G, S::kLo5, S::kLo4 - 1, 0, false, PRE> t5;
    t5.load(c[0]);
    ntt_transpose_multi<F, LOGN, G, S::kLo4, S::kLo5>(c, x);
    ntt_pass_forward_each<F, LOGN, G, S::kLo5, S::kLo4 - 1, 0, false>(c, x, top, t5);
  }
}

template <class F, int LOGN, int G, bool SMALL_INPUT = false, bool AFTER_BARRIER = false, class Ctx, class Top>
TFHE_HD void ntt_forward(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE], const Top& top) {
  const Ctx cs[1] = {c};
  ntt_forward_multi<F, LOGN, G, SMALL_INPUT, AFTER_BARRIER, 1>(
      cs, reinterpret_cast<typename F::elem (&)[1][NttShape<LOGN, G>::kE]>(x), top);
}

template <class F, int LOGN, int G, bool SMALL_INPUT = false, bool AFTER_BARRIER = false, class Ctx>
TFHE_HD void ntt_forward(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE]) {
  const TopFromTable<typename F::elem> top{c.twiddles_uniform(), 1 << LOGN};
  ntt_forward<F, LOGN, G, SMALL_INPUT, AFTER_BARRIER>(c, x, top);
}

// in: x[r] = A_bitrev[tid*E + r].  out: x[r] = N * a[r*64G + tid] (unscaled inverse).
// A prefetched `top` must be ready() before the call.
// (the first pass's twiddles are read on entry -- nothing to hide them behind but the caller's last
// instructions --, every later low pass's before the transpose in front of it)
template <class F, int LOGN, int G, int NS, class Ctx, class Top, class Before = NothingBefore>
TFHE_HD void ntt_inverse_multi(const Ctx (&c)[NS], typename F::elem (&x)[NS][NttShape<LOGN, G>::kE], const Top& top,
                               const Before& before_first_store = Before{}) {
  using S = NttShape<LOGN, G>;
  constexpr bool PRE = PreloadsTwiddles<F>::value;
  if constexpr (S::kPasses == 5) {
    LowPassTwiddles<F, LOGN, G, S::kLo5, S::kLo4 - 1, 0, true, PRE> t5;
    t5.load(c[0]);
    ntt_pass_inverse_each<F, LOGN, G, S::kLo5, S::kLo4 - 1, 0>(c, x, top, t5);
  }
  if constexpr (S::kPasses >= 4) {
    LowPassTwiddles<F, LOGN, G, S::kLo4, S::kLo3 - 1, S::kLo4, true, PRE> t4;
    t4.load(c[0]);
    if constexpr (S::kPasses == 5) {
      before_first_store();
      ntt_transpose_multi<F, LOGN, G, S::kLo5, S::kLo4>(c, x);
    }
    ntt_pass_inverse_each<F, LOGN, G, S::kLo4, S::kLo3 - 1, S::kLo4>(c, x, top, t4);
  }
  LowPassTwiddles<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3, true, PRE> t3;
  t3.load(c[0]);
  if constexpr (S::kPasses >= 4) {
    if constexpr (S::kPasses == 4) before_first_store();
    ntt_transpose_multi<F, LOGN, G, S::kLo4, S::kLo3>(c, x);
  }
  ntt_pass_inverse_each<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3>(c, x, top, t3);
  LowPassTwiddles<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2, true, PRE> t2;
  t2.load(c[0]);
  if constexpr (S::kPasses == 3) before_first_store();
  ntt_transpose_multi<F, LOGN, G, S::kLo3, S::kLo2>(c, x);
  ntt_pass_inverse_each<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2>(c, x, top, t2);
  ntt_transpose_multi<F, LOGN, G, S::kLo2, S::kLo1>(c, x);
  ntt_pass_inverse_each<F, LOGN, G, S::kLo1, LOGN - 1, S::kTBits>(c, x, top, NoPassTwiddles{});
}

template <class F, int LOGN, int G, class Ctx, class Top>
TFHE_HD void ntt_inverse(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE], const Top& top) {
  const Ctx cs[1] = {c};
  ntt_inverse_multi<F, LOGN, G, 1>(cs, reinterpret_cast<typename F::elem (&)[1][NttShape<LOGN, G>::kE]>(x), top);
}

template <class F, int LOGN, int G, class Ctx>
TFHE_HD void ntt_inverse(const Ctx& c, typename F::elem (&x)[NttShape<LOGN, G>::kE]) {
  const TopFromTable<typename F::elem> top{c.twiddles_uniform(), 1 << LOGN};
  ntt_inverse<F, LOGN, G>(c, x, top);
}

// Two polynomials through ONE wave-private buffer, half a step apart: pass(a); T(a); pass(b); T(b); pass'(a); ...  A
// transpose is eight stores and eight loads; the wave's LDS operations execute in order, so b's stores follow a's loads
// into the same buffer without a wait, and the round trip of a's transpose is covered by b's register pass (and the
// other way round) instead of by nothing.  Three-pass shapes with wave-local transposes only (G = 1).
template <class F, int LOGN, int G, class Ctx, class Top, class Before = NothingBefore>
TFHE_HD void ntt_inverse_pair(const Ctx& c, typename F::elem (&a)[NttShape<LOGN, G>::kE],
                              typename F::elem (&b)[NttShape<LOGN, G>::kE], const Top& top,
                              const Before& before_first_store = Before{}) {
  using S = NttShape<LOGN, G>;
  static_assert(G <= 1 && S::kPasses == 3, "wave-local transposes, three passes");
  constexpr bool PRE = PreloadsTwiddles<F>::value;
  LowPassTwiddles<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3, true, PRE> t3;
  t3.load(c);
  ntt_pass_inverse<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3>(c, a, top, t3);
  before_first_store();
  ntt_transpose<F, LOGN, G, S::kLo3, S::kLo2>(c, a);
  ntt_pass_inverse<F, LOGN, G, S::kLo3, S::kLo2 - 1, S::kLo3>(c, b, top, t3);
  LowPassTwiddles<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2, true, PRE> t2;
  t2.load(c);
  ntt_transpose<F, LOGN, G, S::kLo3, S::kLo2>(c, b);
  ntt_pass_inverse<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2>(c, a, top, t2);
  ntt_transpose<F, LOGN, G, S::kLo2, S::kLo1>(c, a);
  ntt_pass_inverse<F, LOGN, G, S::kLo2, S::kTBits - 1, S::kLo2>(c, b, top, t2);
  ntt_transpose<F, LOGN, G, S::kLo2, S::kLo1>(c, b);
  ntt_pass_inverse<F, LOGN, G, S::kLo1, LOGN - 1, S::kTBits>(c, a, top, NoPassTwiddles{});
  ntt_pass_inverse<F, LOGN, G, S::kLo1, LOGN - 1, S::kTBits>(c, b, top, NoPassTwiddles{});
}

}  // namespace tfhe

// wave_ntt.h -- exact negacyclic NTT of one polynomial held by ONE 64-lane wavefront.
//
// N = 2^LOGN coefficients live in E = N/64 registers per lane (u64 each).  The transform is the
// merged-psi Cooley-Tukey NTT (natural order in, bit-reversed order out) and its Gentleman-Sande
// inverse, executed as three "register passes".  A pass owns a window of e = log2(E) index bits:
// in window [LO, LO+e) a lane holds the E indices that differ only in those bits,
//     j = (hi << (LO+e)) | (r << LO) | lo,   lane = (hi << LO) | lo,   r = register number,
// so every butterfly of the stages on those bits is lane-local.  Between passes the polynomial
// is transposed through a wave-private LDS buffer (N u64) addressed with an XOR swizzle that
// makes all ds_write_b64 / ds_read_b64 of the transposes bank-conflict free on gfx950
// (tools/ntt_model.py proves layout and conflict-freedom for LOGN = 9, 10, 11).
//
//   forward : window [6,6+e) (j = r*64 + lane, coalesced) -> [6-e,6) -> [0,e)
//   inverse : the mirror image, ends in [6,6+e) again.
//
// In window [0,e) position pos = lane*E + r of the bit-reversed-order spectrum sits in register r.
// The inverse is NOT scaled by N^-1: the bootstrapping key is pre-scaled instead (bsk_prepare).
//
// One twiddle table serves both directions: psi_rev[k] = psi^bitrev(k), and
// psi^-bitrev(h+i) = -psi_rev[2h-1-i], so the inverse butterfly is (V-U) * psi_rev[2h-1-i].
//
// The arithmetic is a field policy F (field_gl.h: Goldilocks u64, field_fp.h: 42-bit prime in
// fp64); elements are 8 bytes in both, so layouts, swizzles and LDS budgets are identical.
//
// Ctx (GPU: DeviceWave in kernels.hip; CPU tests: the SIMT emulator in tests/emu) provides
//   int lane() const;  void sync() const;  elem* scratch() const;  const elem* twiddles() const;
#pragma once
#include "field_fp.h"
#include "field_gl.h"

namespace tfhe {

template <int LOGN>
struct NttShape {
  static_assert(LOGN >= 9 && LOGN <= 11, "one wavefront per polynomial supports N = 512..2048");
  static constexpr int kLogN = LOGN;
  static constexpr int kN = 1 << LOGN;
  static constexpr int kEBits = LOGN - 6;
  static constexpr int kE = 1 << kEBits;
  // window lows of the three passes (forward order)
  static constexpr int kLo1 = 6;
  static constexpr int kLo2 = 6 - kEBits;
  static constexpr int kLo3 = 0;
};

// XOR swizzle of the transpose buffer (element = u64).  See tools/ntt_model.py::conflicts.
template <int LOGN>
TFHE_HD int ntt_swizzle(int j) {
  if (LOGN == 10) return j ^ ((j >> 4) & 31);
  if (LOGN == 9) return j ^ ((j >> 3) & 7) ^ (((j >> 6) & 3) << 3);
  return j ^ ((j >> 5) & 31);
}

// element index held in register r of `lane` for window [LO, LO+e)
template <int LOGN, int LO>
TFHE_HD int ntt_index(int lane, int r) {
  constexpr int e = NttShape<LOGN>::kEBits;
  return ((lane >> LO) << (LO + e)) | (r << LO) | (lane & ((1 << LO) - 1));
}

// memory position (in u64 elements) of spectrum register r of `lane` inside one NTT-domain
// polynomial of the prepared bootstrapping key: pairs of registers are interleaved so that one
// global_load_dwordx4 per lane reads 64 x 16 B = 1 KiB contiguous.
template <int LOGN>
TFHE_HD int spectrum_slot(int lane, int r) {
  return (r >> 1) * 128 + lane * 2 + (r & 1);
}

template <class F, int LOGN, int LO_FROM, int LO_TO, class Ctx>
TFHE_HD void ntt_transpose(const Ctx& c, typename F::elem (&x)[NttShape<LOGN>::kE]) {
  constexpr int E = NttShape<LOGN>::kE;
  typename F::elem* buf = c.scratch();
  const int lane = c.lane();
#pragma unroll
  for (int r = 0; r < E; ++r) buf[ntt_swizzle<LOGN>(ntt_index<LOGN, LO_FROM>(lane, r))] = x[r];
  c.sync();
#pragma unroll
  for (int r = 0; r < E; ++r) x[r] = buf[ntt_swizzle<LOGN>(ntt_index<LOGN, LO_TO>(lane, r))];
  c.sync();
}

// forward stages on bits BHI..BLO (descending) of window [LO, LO+e).  SMALL_FIRST: the inputs of
// the first stage handled here are small integers (gadget digits), so its twiddle products may
// use F::mul_small (exact without reduction in the fp64 field).
template <class F, int LOGN, int LO, int BHI, int BLO, bool SMALL_FIRST, class Ctx>
TFHE_HD void ntt_pass_forward(const Ctx& c, typename F::elem (&x)[NttShape<LOGN>::kE]) {
  typedef typename F::elem elem;
  constexpr int E = NttShape<LOGN>::kE;
  constexpr int e = NttShape<LOGN>::kEBits;
  const elem* tw = c.twiddles();
  const int hi = c.lane() >> LO;
#pragma unroll
  for (int b = BHI; b >= BLO; --b) {
    const int rb = b - LO;
    const int m = NttShape<LOGN>::kN >> (b + 1);
    const int base = m + (hi << (LO + e - b - 1));
#pragma unroll
    for (int r0 = 0; r0 < E; ++r0) {
      if ((r0 >> rb) & 1) continue;
      const int r1 = r0 | (1 << rb);
      const elem w = tw[base + (r0 >> (rb + 1))];
      const elem u = x[r0];
      const elem v = (SMALL_FIRST && b == BHI) ? F::mul_small(x[r1], w) : F::mul(x[r1], w);
      x[r0] = F::add(u, v);
      x[r1] = F::sub(u, v);
    }
  }
}

// inverse stages on bits BLO..BHI (ascending) of window [LO, LO+e)
template <class F, int LOGN, int LO, int BHI, int BLO, class Ctx>
TFHE_HD void ntt_pass_inverse(const Ctx& c, typename F::elem (&x)[NttShape<LOGN>::kE]) {
  typedef typename F::elem elem;
  constexpr int E = NttShape<LOGN>::kE;
  constexpr int e = NttShape<LOGN>::kEBits;
  const elem* tw = c.twiddles();
  const int hi = c.lane() >> LO;
#pragma unroll
  for (int b = BLO; b <= BHI; ++b) {
    const int rb = b - LO;
    const int h = NttShape<LOGN>::kN >> (b + 1);
    // psi^-bitrev(h+i) = -psi_rev[2h-1-i]
    const int top = 2 * h - 1 - (hi << (LO + e - b - 1));
#pragma unroll
    for (int r0 = 0; r0 < E; ++r0) {
      if ((r0 >> rb) & 1) continue;
      const int r1 = r0 | (1 << rb);
      const elem w = tw[top - (r0 >> (rb + 1))];
      const elem u = x[r0];
      const elem v = x[r1];
      x[r0] = F::add(u, v);
      x[r1] = F::mul(F::sub(v, u), w);
    }
  }
}

// in: x[r] = a[r*64 + lane] (canonical field elements).  out: x[r] = A_bitrev[lane*E + r].
// SMALL_INPUT: every |x[r]| <= 2^F::kSmallBits on entry (gadget digits).
template <class F, int LOGN, bool SMALL_INPUT = false, class Ctx>
TFHE_HD void ntt_forward(const Ctx& c, typename F::elem (&x)[NttShape<LOGN>::kE]) {
  using S = NttShape<LOGN>;
  ntt_pass_forward<F, LOGN, S::kLo1, LOGN - 1, 6, SMALL_INPUT>(c, x);
  ntt_transpose<F, LOGN, S::kLo1, S::kLo2>(c, x);
  ntt_pass_forward<F, LOGN, S::kLo2, 5, S::kLo2, false>(c, x);
  ntt_transpose<F, LOGN, S::kLo2, S::kLo3>(c, x);
  ntt_pass_forward<F, LOGN, S::kLo3, S::kLo2 - 1, 0, false>(c, x);
}

// in: x[r] = A_bitrev[lane*E + r].  out: x[r] = N * a[r*64 + lane] (unscaled inverse).
template <class F, int LOGN, class Ctx>
TFHE_HD void ntt_inverse(const Ctx& c, typename F::elem (&x)[NttShape<LOGN>::kE]) {
  using S = NttShape<LOGN>;
  ntt_pass_inverse<F, LOGN, S::kLo3, S::kLo2 - 1, 0>(c, x);
  ntt_transpose<F, LOGN, S::kLo3, S::kLo2>(c, x);
  ntt_pass_inverse<F, LOGN, S::kLo2, 5, S::kLo2>(c, x);
  ntt_transpose<F, LOGN, S::kLo2, S::kLo1>(c, x);
  ntt_pass_inverse<F, LOGN, S::kLo1, LOGN - 1, 6>(c, x);
}

}  // namespace tfhe

// field_fft.h -- transform policy: the negacyclic product over Z through a complex FFT in fp64,
// exact by a proven rounding bound.  Two coefficients per element: a polynomial a(X) mod X^N + 1 with
// real coefficients is folded to M = N/2 complex numbers b_j = a_j + i a_{j+M} and evaluated at the M
// points zeta^(4k+1), zeta = exp(i pi / N) -- the M-th roots of i -- where the negacyclic product is
// pointwise (the other N/2 evaluations are the complex conjugates).  "b(Y) mod (Y^M - i)" splits exactly
// like "a(X) mod (X^N + 1)" does in the prime fields: Y^2h - c = (Y^h - sqrt c)(Y^h + sqrt c), so the
// same merged-twiddle Cooley-Tukey / Gentleman-Sande passes of wave_ntt.h run on it with the table
//     T[h + i] = exp(i pi (1 + 4 bitrev_s(i)) / (4h)),   h = 2^s, i < h,
// and no separate twist.  The inverse butterfly multiplies by -conj(T[h + i]) (|T| = 1).
//
// Why: a complex butterfly is 8 fp64 instructions like the 42-bit field's (6 in the forward direction, see
// butterfly_forward), but there are half as many (M/2 log2 M instead of N/2 log2 N per transform) and a
// multiply-accumulate term is 4 FMAs for TWO coefficients instead of 7 instructions for one -- 2,300 instead of
// 5,090 VALU instructions per wave and CMUX iteration at N = 1024, k = 1, l = 3.
//
// Exactness.  The values are NOT exact integers on the way; the final result is: the inverse transform
// returns z~_j = z_j + e_j with z_j the integer convolution value, and |e_j| < 1/2 makes
// rint(z~_j) = z_j.  Bound on e (Higham, Accuracy and Stability of Numerical Algorithms, 2nd ed., Thm 24.2:
// a radix-2 FFT of n stages computed with twiddles of absolute error <= mu satisfies
// ||X~ - X||_2 <= n eta / (1 - n eta) ||X||_2, eta = mu + gamma_4 (sqrt 2 + mu), gamma_4 = 4u / (1 - 4u),
// u = 2^-53; our twiddles are correctly rounded from long double, each component within u/2, |error| <= u / sqrt 2,
// so eta <= 7.1 u = 7.9e-16 with room to spare; the forward transforms here use a six-FMA butterfly whose own
// per-stage perturbation is 4.32 u, derived at FftField::butterfly_forward below; the inverse ones the textbook
// sum / difference-times-twiddle form; the theorem's mechanism is an induction that needs nothing but a per-stage
// perturbation bound: x~_k = A_k x~_(k-1) + f_k with ||f_k||_2 <= eps ||A_k x~_(k-1)||_2 and A_k / sqrt 2 unitary gives
// ||x~_n - x_n||_2 <= n eps ||x_n||_2 (1 + O(n eps)), whatever the order of the operations inside a butterfly).
// With x the folded digits (|x_j| <= sqrt 2 B), y the folded key half
// (|y_j| <= sqrt 2 2^15), X, Y their unnormalised transforms (||X||_2 <= M |x|max, ||Y||_inf <= M |y|max),
// R = (k+1) l rows accumulated in the transform domain, n = log2 M stages:
//   forward errors   ||X~ - X||_2 <= n eta M |x|,  the same for Y (the prepared key)
//   products         every row's product carries 2 n eta M^2 |x| |y| of forward error; on top of that the R terms
//                    are ACCUMULATED by a chain of FMAs (mul_add: two per component and row), and each of them
//                    rounds the RUNNING sum, not one term: adding row r changes a component by at most
//                    u (|acc| + |d k|) + u |acc'| <= 2 u (|acc| + |d| |k|), the complex value by sqrt 2 times that, and
//                    ||acc_(r-1)||_2 + ||X_r o Y_r||_2 <= r M^2 |x| |y|, so row r adds 2 sqrt 2 u r M^2 |x| |y| and the
//                    chain sum_{r <= R} of it sqrt 2 u R (R + 1) M^2 |x| |y| -- quadratic in R (round 2 charged 4u per
//                    row as if the sum were exact: too small from R = 2 on, by 3 % at cfg2, 15 % at R = 18):
//                    ||P~ - P||_2 <= R (2 n eta + sqrt 2 (R + 1) u) M^2 |x| |y|
//   inverse (x 1/M, exact power of two): ||z~ - z||_2 <= ||P~ - P||_2 / sqrt M + n eta ||z||_2,
//                    ||z||_2 <= sqrt M R M |x| |y|
//   ==> max_j |e_j| <= ||z~ - z||_2 <= (3 n eta + sqrt 2 (R + 1) u) R M^1.5 |x|max |y|max =: error_bound()
// (evaluated with eta = 7.1 u, sqrt 2 rounded up to 1.42 and 0.1 % on top for the second-order terms, which are
// below n eta = 1e-14 relative).
// cfg2 (N=1024, k=1, l=3, B=2^7): 0.0131; N=2048, k=2, l=4, B=2^8: 0.170; the reference's default
// (N=512, k=2, l=6, B=2^4): 0.0017.  The context admits this backend only below kMaxError = 1/4 -- half of the 1/2
// at which a bit would change; the largest admitted sets are N=1024, k=2, l=2, B=2^11 (0.209) and N=512, k=1, l=2,
// B=2^13 (0.174), and those two run in the rounding-margin tests (emulator and gfx950) with operands of maximal
// magnitude and random signs.  Measured errors are four to five orders of magnitude below the bound (random data:
// 5e-7 at cfg2), and the worst-case-magnitude
// tests (emulator and GPU) drive every digit to +B / -B/2 and every key word to 0x80008000 / 0x7FFF7FFF.
// The 32-bit key word is split into signed 16-bit halves like in field_fp.h (two spectra per key
// polynomial): with the word taken whole the bound is 2^16 times larger and fails.
#pragma once
#include <float.h>
#include <math.h>

#if defined(TFHE_FFT_TRACK_ERROR) && !defined(__HIPCC__)
#include <atomic>
#endif
#include <vector>

#include "platform.h"

namespace tfhe {

struct alignas(16) cplx {
  double re, im;
};

#if defined(TFHE_FFT_TRACK_ERROR) && !defined(__HIPCC__)
// Host emulator only: the largest |t - rint(t)| seen by FftField::to_u32 since the last reset -- the measured
// counterpart of error_bound() (tests/test_emu_kernels.py::test_fft_rounding_margin).
inline std::atomic<double>& fft_error_slot() {
  static std::atomic<double> worst{0.0};
  return worst;
}
inline void fft_track_error(double t) {
  const double e = fabs(t - rint(t));
  std::atomic<double>& w = fft_error_slot();
  double cur = w.load(std::memory_order_relaxed);
  while (e > cur && !w.compare_exchange_weak(cur, e, std::memory_order_relaxed)) {
  }
}
#endif

#if defined(TFHE_FFT_TRACK_ERROR) && defined(__HIPCC__)
// Probe build of the GPU library only (libtfhe_hip_probe.so, never the product): the bit pattern of the largest
// |t - rint(t)| any lane has lifted on gfx950 since the last reset (non-negative doubles order like their bit
// patterns).  One copy per translation unit; kernels.hip owns the one the kernels write (launch::fft_margin).
static __device__ unsigned long long g_fft_margin_bits = 0ull;
__device__ __forceinline__ void fft_track_error_device(double t) {
  const double e = fabs(t - rint(t));
  unsigned long long bits;
  __builtin_memcpy(&bits, &e, sizeof(bits));
  // plain read first: the maximum settles after a few values and the atomics stop
  if (bits > __atomic_load_n(&g_fft_margin_bits, __ATOMIC_RELAXED)) atomicMax(&g_fft_margin_bits, bits);
}
#endif

struct FftField {
  typedef cplx elem;
  static constexpr int kParts = 2;
  static constexpr int kId = 5;
  // two ring coefficients per transform element: the transform has N/2 points
  static constexpr int kLogShrink = 1;
  static constexpr int kCoeffs = 2;
  static constexpr int kInverseSweepEvery = 0;
  static constexpr bool kReduceSpectrum = false;
  template <int E>
  static constexpr bool split_accum() { return false; }
  TFHE_HD static elem accum_init() { return elem{0.0, 0.0}; }
  TFHE_HD static void mac(elem, elem, elem&, elem&) {}
  TFHE_HD static elem mac_finish(elem a, elem) { return a; }
  static constexpr int kMaxRows = 1 << 10;
  static constexpr double kMaxError = 0.25;

  TFHE_HD static elem zero() { return elem{0.0, 0.0}; }
  TFHE_HD static elem add(elem a, elem b) { return elem{a.re + b.re, a.im + b.im}; }
  TFHE_HD static elem sub(elem a, elem b) { return elem{a.re - b.re, a.im - b.im}; }
  // complex product, 2 multiplications + 2 FMAs
  TFHE_HD static elem mul(elem a, elem w) {
    return elem{__builtin_fma(a.re, w.re, -(a.im * w.im)), __builtin_fma(a.re, w.im, a.im * w.re)};
  }
  // The forward butterfly (u + w b, u - w b) in SIX fused multiply-adds instead of a complex product (4) and two
  // complex sums (4): y0 = u + w b is two chained FMAs per component, and y1 = u - w b = 2 u - y0 one more.
  // Its rounding stays inside the eta the bound above is evaluated with.  With a = u, U = 2^-53, A = |a|, B = |b|:
  //   y0.re = fl(fl(a.re + w.re b.re) - w.im b.im): |error| <= U (2 |a.re| + 2 |w.re b.re| + |w.im b.im|), the same
  //   shape for y0.im, so |y0~ - (a + w^ b)| <= U (2 A + sqrt 5 B); the stored twiddle w^ adds |w^ - w| B <= U B / sqrt 2:
  //   |y0~ - y0| <= U (2 A + 2.95 B);
  //   y1~ = fl(2 a - y0~) = (y1 - (y0~ - y0)) (1 + d): |y1~ - y1| <= |y0~ - y0| + U (A + B) <= U (3 A + 3.95 B).
  // The pair's error vector has 2-norm <= U sqrt((2A + 2.95B)^2 + (3A + 3.95B)^2) <= 6.11 U sqrt(A^2 + B^2) (largest
  // eigenvalue of [[13, 17.75], [17.75, 24.305]] is 37.28), and a stage multiplies the 2-norm of its input by exactly
  // sqrt 2: the stage's relative perturbation is <= 6.11 / sqrt 2 = 4.32 U < eta = 7.1 U -- Higham's Theorem 24.2 only
  // needs ||dA_k||_2 <= eta ||A_k||_2 per stage, whatever the order of the operations inside the butterfly.
#ifndef TFHE_FFT_FMA_BUTTERFLY
#define TFHE_FFT_FMA_BUTTERFLY 1  // 0: the 8-instruction product + two sums (A/B builds only)
#endif
  static constexpr bool kFusedForwardButterfly = TFHE_FFT_FMA_BUTTERFLY != 0;
  TFHE_HD static void butterfly_forward(elem u, elem b, elem w, elem& y0, elem& y1) {
    const double re = __builtin_fma(-b.im, w.im, __builtin_fma(b.re, w.re, u.re));
    const double im = __builtin_fma(b.im, w.re, __builtin_fma(b.re, w.im, u.im));
    y0 = elem{re, im};
    y1 = elem{__builtin_fma(2.0, u.re, -re), __builtin_fma(2.0, u.im, -im)};
  }
  // The inverse butterfly (u + v, (v - u) (-conj w)) written as (u - v) conj(w): the same value, and no result has to
  // be negated after the fact (a v_xor on the sign word each time: the FMA's output has no negation modifier).
  TFHE_HD static void butterfly_inverse(elem u, elem v, elem w, elem& y0, elem& y1) {
    const double dre = u.re - v.re, dim = u.im - v.im;
    y0 = elem{u.re + v.re, u.im + v.im};
    y1 = elem{__builtin_fma(dre, w.re, dim * w.im), __builtin_fma(dim, w.re, -(dre * w.im))};
  }
  // the low register windows' twiddles are read a transpose ahead of their pass (wave_ntt.h::PassTwiddles)
  static constexpr bool kPreloadTwiddles = true;
  // a * (-conj(w)): the inverse butterfly's twiddle
  TFHE_HD static elem mul_inverse(elem a, elem w) {
    return elem{-__builtin_fma(a.re, w.re, a.im * w.im), __builtin_fma(a.re, w.im, -(a.im * w.re))};
  }
  // acc + d * k: 4 FMAs
  static constexpr bool kFusedMac = true;
  TFHE_HD static elem mul_add(elem d, elem k, elem acc) {
    return elem{__builtin_fma(-d.im, k.im, __builtin_fma(d.re, k.re, acc.re)),
                __builtin_fma(d.im, k.re, __builtin_fma(d.re, k.im, acc.im))};
  }
  // table entry the inverse butterfly of node h + i reads (mul_inverse conjugates it)
  TFHE_HD static constexpr int inverse_twiddle_index(int h, int i) { return h + i; }
  static constexpr int kSmallBits = 31;
  static constexpr int kMaxLogBase = 16;  // far beyond what the error bound admits
  TFHE_HD static elem mul_small(elem a, elem w) { return mul(a, w); }
  static constexpr bool kFuseFirstTwo = false;
  TFHE_HD static void radix4_small(elem&, elem&, elem&, elem&, elem, elem, elem, elem, elem) {}
  TFHE_HD static elem radix8_small_v(elem, elem, elem, elem, elem, elem, elem, elem) { return zero(); }
  TFHE_HD static elem reduce(elem x) { return x; }
  TFHE_HD static elem before_inverse(elem a) { return a; }
  // element j of the folded polynomial: coefficients j and j + N/2
  TFHE_HD static elem from_digits(const u32 (&d)[kCoeffs]) { return elem{(double)(i32)d[0], (double)(i32)d[1]}; }
  TFHE_HD static double key_half(u32 w, int part) {
    const i32 lo = (i32)(int16_t)(w & 0xFFFFu);
    if (part == 0) return (double)lo;
    return (double)((i32)(w - (u32)lo) >> 16);  // see field_fp.h::from_key_word
  }
  TFHE_HD static elem from_key_words(const u32 (&w)[kCoeffs], int part) {
    return elem{key_half(w[0], part), key_half(w[1], part)};
  }
  // scaling of the prepared key by the inverse transform's 1/M: a power of two, exact
  TFHE_HD static elem scale_key(elem x, elem n_inv) { return elem{x.re * n_inv.re, x.im * n_inv.re}; }
  // |t| < 2^51, t within 1/4 of an integer -> that integer mod 2^32: t + 1.5 * 2^52 rounds to nearest in
  // [2^52, 2^53), where doubles are the integers; the low 32 bits of the sum's mantissa are the answer
  TFHE_HD static u32 to_u32(double t) {
#if defined(TFHE_FFT_TRACK_ERROR) && !defined(__HIPCC__)
    fft_track_error(t);  // tests/emu only: distance of every lifted value from the integer it rounds to
#elif defined(TFHE_FFT_TRACK_ERROR) && defined(__HIP_DEVICE_COMPILE__)
    fft_track_error_device(t);  // probe build only: the same measurement on gfx950
#endif
    const double shifted = t + 6755399441055744.0;  // 1.5 * 2^52
    u64 bits;
    __builtin_memcpy(&bits, &shifted, sizeof(bits));
    return (u32)bits;
  }
  TFHE_HD static void finish(const elem (&parts)[kParts], u32 (&out)[kCoeffs]) {
    out[0] = to_u32(parts[0].re) + (to_u32(parts[1].re) << 16);
    out[1] = to_u32(parts[0].im) + (to_u32(parts[1].im) << 16);
  }

  // ---- host-side constants ----
  // out: m + 18 elements (wave_ntt.h::ntt_twiddle_words) for a transform of m = 2^logm points
  static inline void fill_twiddles(int logm, elem* out) {
    const int m = 1 << logm;
    // the bound's twiddle accuracy (each component correctly rounded to double) rests on an 80-bit long double
#if !defined(__HIP_DEVICE_COMPILE__)  // (host function; the device pass of hipcc parses it with long double = double)
    static_assert(LDBL_MANT_DIG >= 64, "twiddles are rounded from a 64-bit-mantissa long double");
#endif
    const long double pi = 3.14159265358979323846264338327950288L;
    out[0] = elem{1.0, 0.0};
    for (int s = 0; (1 << s) < m; ++s) {
      const int h = 1 << s;
      for (int i = 0; i < h; ++i) {
        int rev = 0;
        for (int b = 0; b < s; ++b) rev |= ((i >> b) & 1) << (s - 1 - b);
        const long double angle = pi * (long double)(1 + 4 * rev) / (long double)(4 * h);
        out[h + i] = elem{(double)cosl(angle), (double)sinl(angle)};
      }
    }
    for (int i = 0; i < 18; ++i) out[m + i] = elem{0.0, 0.0};
  }
  static inline elem n_inv(int logm) { return elem{1.0 / (double)(1 << logm), 0.0}; }
  // worst-case |error| of one output coefficient before rounding (header comment); log_n = ring degree
  static inline double error_bound(int log_n, int rows, int log_base) {
    const double u = 1.1102230246251565e-16, eta = 7.1 * u;
    const double m = (double)(1 << (log_n - 1)), n = (double)(log_n - 1), r = (double)rows;
    const double x = 1.4142135623730951 * (double)(1u << log_base), y = 1.4142135623730951 * 32768.0;
    // transforms (forward digits, forward key, inverse) + the FMA chain that accumulates r rows
    return 1.001 * (3.0 * n * eta + 1.42 * (r + 1.0) * u) * r * m * sqrt(m) * x * y;
  }
  static inline double exact_bits() { return 0.0; }  // not used: error_bound() decides
  static inline double key_bits() { return 15.0; }
};

}  // namespace tfhe

// field_fp49.h -- field policy: F_p with p = 671317819555841 ~ 2^49.25 (prime,
// p - 1 = 2^15 * 5 * 1277 * 3208613, generator 3), elements are IEEE doubles holding exact integers,
// ONE spectrum per key polynomial.
//
// Why: the 42-bit field (field_fp.h) needs the key split into two 16-bit halves, which doubles the
// multiply-accumulate tiles and the inverse transforms.  Parameter sets with a small gadget base --
// the reference's default (N = 512, k = 2, l = 6, log_base = 4: R N B 2^31 = 2^48.17) -- fit a
// 49-bit prime with the key word taken whole (as a signed 32-bit integer).  The price is less
// room for lazy additions (2^53 / p = 13.4), paid with a few reduction sweeps.
//
// Exactness (all values are integers; |twiddle|, |key spectrum| <= p/2 < 2^48.26):
//   mul(a, w), |a| < 2^53: h = RN(a w), l = a w - h exactly (FMA); q = rint(RN(h/p)) is an integer
//     below 2^52.1 within 1 + 2^-50|q| of h/p, so h - q p is an integer of magnitude <= 1.6 p < 2^50
//     and the FMA returns it exactly; adding the integer l (|l| <= ulp(h)/2 <= 2^48) is exact.
//     |mul(a, w)| <= p/2 + 2 |a| p / 2^53 (conservative).
//   forward (Cooley-Tukey): A_{s+1} <= 1.149 A_s + p/2 from digits: 8.3 p after 9 stages, 10.1 p
//     after 10, 12.1 p after 11 -- below 2^53 = 13.4 p.  The spectrum is reduced to |.| <= p/2 before
//     it is published (kReduceSpectrum), so a MAC term is <= 0.58 p and R <= 20 of them stay exact.
//   inverse (Gentleman-Sande): sums double per stage, so every element is reduced after every 4th
//     stage (kInverseSweepEvery): p/2 -> 8 p, the operand of a multiplication (v - u) <= 8 p.
//   lift: |t| <= R N B 2^31 < p/2 (checked when the context picks the field), so the balanced
//     residue IS t.
#pragma once
#include <math.h>

#include <vector>

#include "platform.h"

namespace tfhe {

struct Fp49Field {
  typedef double elem;
  static constexpr int kParts = 1;
  static constexpr int kId = 4;

  static constexpr double P = 671317819555841.0;
  static constexpr double PINV = 1.0 / 671317819555841.0;
  static constexpr u64 P_INT = 671317819555841ull;

  // hooks read by wave_ntt.h / pbs_wave.h (0 / false in the other fields)
  static constexpr int kInverseSweepEvery = 4;
  static constexpr bool kReduceSpectrum = true;
  static constexpr int kMaxRows = 20;  // lazily accumulated MAC terms (see mac() below)
  // Multiply-accumulate without reducing every term: a product d*k (|d|, |k| <= p/2 = 2^48.26, so
  // below 2^96.6) is split EXACTLY into h, a multiple of 2^48 obtained by rounding against the
  // constant 1.5 * 2^100 (one FMA and one subtraction, both exact: the sum stays inside the binade
  // of the constant), and l = d*k - h (one FMA, |l| <= 2^47).  Up to 20 h's add exactly (multiples
  // of 2^48 below 2^101: 20 * 2^96.6 = 2^100.9) and so do 20 l's (below 2^51.4): 5 instructions
  // per term instead of 7.  mac_finish() folds the pair back.  reduce() is exact for sum_h too:
  // q = rint(sum_h / p) is an integer below 2^51.7 within 1 of the true quotient and sum_h - q p is
  // an integer below 1.5 p, so the FMA returns it exactly; adding sum_l (< 4.3 p) and reducing once
  // more gives |.| <= p/2.
  template <int E>
  static constexpr bool split_accum() { return true; }
  static constexpr double kMacMagic = 1.5 * 1267650600228229401496703205376.0;  // 1.5 * 2^100
  TFHE_HD static elem accum_init() { return 0.0; }
  TFHE_HD static void mac(elem d, elem k, elem& acc_h, elem& acc_l) {
    const double t = __builtin_fma(d, k, kMacMagic);
    const double h = t - kMacMagic;
    acc_l += __builtin_fma(d, k, -h);
    acc_h += h;
  }
  TFHE_HD static elem mac_finish(elem acc_h, elem acc_l) {
    return reduce(reduce(acc_h) + acc_l);
  }

  TFHE_HD static elem zero() { return 0.0; }
  TFHE_HD static elem add(elem a, elem b) { return a + b; }
  TFHE_HD static elem sub(elem a, elem b) { return a - b; }
  TFHE_HD static elem mul(elem a, elem w) {
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_rint(h * PINV);
    return __builtin_fma(-q, P, h) + l;
  }
  // no reduction-free small stage here: 2^4 * p/2 already touches 2^53
  static constexpr int kSmallBits = 31;
  // (k+1) l N B 2^31 < 2^48.25 with N >= 512 and (k+1) l >= 2 leaves B < 2^7.25
  static constexpr int kMaxLogBase = 8;
  TFHE_HD static elem mul_small(elem a, elem w) { return mul(a, w); }
  static constexpr bool kFuseFirstTwo = false;
  TFHE_HD static void radix4_small(elem&, elem&, elem&, elem&, elem, elem, elem, elem, elem) {}
  TFHE_HD static elem radix8_small_v(elem, elem, elem, elem, elem, elem, elem, elem) { return 0.0; }
  // x -> balanced residue, |x| < 2^53
  TFHE_HD static elem reduce(elem x) { return __builtin_fma(-__builtin_rint(x * PINV), P, x); }
  TFHE_HD static elem from_digit(u32 d) { return (double)(i32)d; }
  // the key word as a signed 32-bit integer: w mod 2^32 with |.| <= 2^31
  TFHE_HD static elem from_key_word(u32 w, int) { return (double)(i32)w; }
  TFHE_HD static elem before_inverse(elem a) { return reduce(a); }
  // exact integer t (|t| < 2^51) -> t mod 2^32: t + 1.5 * 2^52 lies in [2^52, 2^53), where doubles
  // are the integers, so the sum is exact and its 52 mantissa bits hold 2^51 + t; their low 32
  // bits are t mod 2^32 (two's complement for negative t).  One add instead of mul, floor, fma, cvt.
  TFHE_HD static u32 to_u32(elem t) {
    const double shifted = t + 6755399441055744.0;  // 1.5 * 2^52
    u64 bits;
    __builtin_memcpy(&bits, &shifted, sizeof(bits));
    return (u32)bits;
  }
  TFHE_HD static u32 finish(const elem (&parts)[kParts]) { return to_u32(reduce(parts[0])); }

  // ---- hooks shared with the complex transform (field_fft.h): one ring coefficient per element here
  static constexpr int kLogShrink = 0;
  static constexpr int kCoeffs = 1;
  static constexpr bool kFusedMac = false;
  TFHE_HD static elem mul_add(elem d, elem k, elem acc) { return add(acc, mul(d, k)); }
  TFHE_HD static elem mul_inverse(elem a, elem w) { return mul(a, w); }
  TFHE_HD static constexpr int inverse_twiddle_index(int h, int i) { return 2 * h - 1 - i; }
  TFHE_HD static elem from_digits(const u32 (&d)[kCoeffs]) { return from_digit(d[0]); }
  TFHE_HD static elem from_key_words(const u32 (&w)[kCoeffs], int part) { return from_key_word(w[0], part); }
  TFHE_HD static elem scale_key(elem x, elem n_inv) { return reduce(mul(x, n_inv)); }
  TFHE_HD static void finish(const elem (&parts)[kParts], u32 (&out)[kCoeffs]) { out[0] = finish(parts); }

  // ---- host-side constants (integer arithmetic mod p) ----
  static inline u64 mulmod_u64(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % P_INT); }
  static inline u64 powmod_u64(u64 b, u64 e) {
    u64 r = 1;
    while (e) {
      if (e & 1) r = mulmod_u64(r, b);
      b = mulmod_u64(b, b);
      e >>= 1;
    }
    return r;
  }
  static inline double balanced(u64 v) { return v > P_INT / 2 ? -(double)(P_INT - v) : (double)v; }
  // out: n + 18 elements (wave_ntt.h::ntt_twiddle_words); the extra ones are unused here
  static inline void fill_twiddles(int logn, elem* out) {
    const int n = 1 << logn;
    const u64 psi = powmod_u64(3, (P_INT - 1) >> (logn + 1));  // 3 generates F_p^*
    std::vector<u64> raw(n);
    u64 pw = 1;
    for (int k = 0; k < n; ++k) {
      int rev = 0;
      for (int b = 0; b < logn; ++b) rev |= ((k >> b) & 1) << (logn - 1 - b);
      raw[rev] = pw;
      out[rev] = balanced(pw);
      pw = mulmod_u64(pw, psi);
    }
    out[n] = balanced(mulmod_u64(raw[1], raw[2]));
    out[n + 1] = balanced(mulmod_u64(raw[1], raw[3]));
    for (int i = 2; i < 18; ++i) out[n + i] = 0.0;  // fused-stage constants of the 42-bit field: unused here
  }
  static inline elem n_inv(int logn) { return balanced(powmod_u64((u64)1 << logn, P_INT - 2)); }
  static inline double exact_bits() { return 48.25; }  // |t| < p/2 = 2^48.254
  static inline double key_bits() { return 31.0; }     // the whole word, signed
};

}  // namespace tfhe

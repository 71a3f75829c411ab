// field_fp.h -- field policy: F_p with p = 2^42 - 24575 (prime, p - 1 = 2^13 * 536870909),
// elements are IEEE doubles holding exact integers.
//
// Why: on gfx950 fp64 FMA/mul/add/rndne issue at the same ~4.5 cycles per wave as every integer
// op (profiles/r01_valu_issue_rates_gfx950.txt), and a modular butterfly costs 8 fp64
// instructions here against ~46 integer instructions for Goldilocks.
//
// Exactness (all values are integers, |twiddle|, |key spectrum| <= p/2 < 2^41):
//   mul(a, w), |a| < 2^53:  h = RN(a*w), l = a*w - h exactly (FMA), q = rint(RN(h/p)) is within 1
//     of h/p, so h - q*p is an integer of magnitude <= 1.5p + |a w| 2^-52 < 2^53: the FMA returns
//     it exactly, and adding the integer l (|l| <= 2^41) is exact too.  |mul| <= p/2 + 1.5|a|p/2^53.
//   forward (Cooley-Tukey) stage: values grow by at most |mul| ~ p/2 per stage: <= 6.1p after 11
//     stages; inverse (Gentleman-Sande) stage: sums double: p/2 -> 2^11 * p/2 = 2^52 for N = 2048.
//     Every add/sub result stays below 2^53, hence exact.  Accumulators (sum of R products, each
//     <= 0.51p) are reduced to |.| <= p/2 before the inverse transform.
//   lift: the true integer result t of one part satisfies |t| <= R*N*B*2^15 < p/2 (checked when the
//     context is created), so the balanced residue IS t.
// The 32-bit key word is split into two signed 16-bit halves w = lo + 2^16 hi (two spectra per key
// polynomial, two accumulators per output column); result = t_lo + 2^16 t_hi mod 2^32.
#pragma once
#include <math.h>

#include <vector>

#include "platform.h"

namespace tfhe {

struct FpField {
  typedef double elem;
  static constexpr int kParts = 2;
  static constexpr int kId = 2;
  // hooks of fields with less lazy headroom (field_fp49.h); 11 bits of it here: never needed
  static constexpr int kInverseSweepEvery = 0;
  static constexpr bool kReduceSpectrum = false;
  // (h, l) accumulator pairs (field_fp49.h) are not used here: summing unreduced products would cut the
  // multiply-accumulate from 7 to 4 instructions per term, but the second accumulator set does not fit
  // the register budget of any shape -- measured with the 4-instruction form (accumulate inside the
  // binade of 1.5 * 2^100): 83-85 spilled registers at N = 1024 (256-register budget; before and after the
  // scalar-twiddle change), 37-80 at N = 512 /
  // N = 2048 over four waves (168-register budget for three waves per SIMD), profiles/r02_*.
  template <int E>
  static constexpr bool split_accum() { return false; }
  TFHE_HD static elem accum_init() { return 0.0; }
  TFHE_HD static void mac(elem, elem, elem&, elem&) {}
  TFHE_HD static elem mac_finish(elem a, elem) { return a; }
  static constexpr int kMaxRows = 1 << 10;

  static constexpr double P = 4398046486529.0;  // 2^42 - 24575
  static constexpr double PINV = 1.0 / 4398046486529.0;
  static constexpr u64 P_INT = 4398046486529ull;
  static constexpr u64 ROOT_8192 = 0;  // unused; see root_of_unity()

  TFHE_HD static elem zero() { return 0.0; }
  TFHE_HD static elem add(elem a, elem b) { return a + b; }
  TFHE_HD static elem sub(elem a, elem b) { return a - b; }
  TFHE_HD static elem mul(elem a, elem w) {
    const double h = a * w;
    const double l = __builtin_fma(a, w, -h);
    const double q = __builtin_rint(h * PINV);
    return __builtin_fma(-q, P, h) + l;
  }
  // |a| <= 2^kSmallBits (a gadget digit, |d| <= B): a*w is an exact integer below 2^53, no
  // reduction needed.  The unreduced value (up to 2^52) only ever gets ADDED to later-stage terms
  // (<= +p/2 per stage) or passes through mul(), which accepts any |a| < 2^53.
  static constexpr int kSmallBits = 9;
  static constexpr int kMaxLogBase = kSmallBits;  // the context refuses larger bases for this field
  TFHE_HD static elem mul_small(elem a, elem w) { return a * w; }
  // The first two Cooley-Tukey stages on four small inputs |a|,|b|,|c|,|d| <= 2^9 as one exact
  // radix-4 step: every product is below 2^50 and every output is a sum of at most three products
  // plus one input, below 2^51.6 -- all exact integers in fp64, no reduction.  (a,c) and (b,d) are
  // the stage-1 pairs with twiddle w1; (a,b) uses w2a and (c,d) uses w2b in stage 2;
  // w12a = w1*w2a, w12b = w1*w2b (mod p, balanced).  10 instructions instead of 22.
  static constexpr bool kFuseFirstTwo = true;
  TFHE_HD static void radix4_small(elem& a, elem& b, elem& c, elem& d, elem w1, elem w2a, elem w2b,
                                   elem w12a, elem w12b) {
    const elem s = __builtin_fma(w1, c, a);      // a + w1 c
    const elem sp = __builtin_fma(-w1, c, a);    // a - w1 c
    const elem t = __builtin_fma(w12a, d, w2a * b);    // w2a (b + w1 d)
    const elem tp = __builtin_fma(-w12b, d, w2b * b);  // w2b (b - w1 d)
    a = s + t;
    b = s - t;
    c = sp + tp;
    d = sp - tp;
  }
  // Third stage fused onto radix4_small: the multiplied leg w3[q] * v_q of the stage-3 butterfly of
  // quarter q, where v_q is output q of the radix-4 step on the small inputs (a, b, c, d), taken
  // straight from those inputs with the four pre-multiplied coefficients k = w3[q] * (1, +-w2, +-w1,
  // +-w1 w2) (signs folded in, mod p, balanced; fill_twiddles stores them at [N+2+4q ..]).  Every
  // product is below 2^41 * 2^9 = 2^50, the sum of four below 2^52, and the butterfly's other leg
  // (a radix4_small output) below 2^51.6: outputs stay below 2^52.8 < 2^53 -- exact integers, no
  // reduction.  4 instructions.
  TFHE_HD static elem radix8_small_v(elem a, elem b, elem c, elem d, elem k0, elem k1, elem k2, elem k3) {
    return __builtin_fma(k3, d, __builtin_fma(k2, c, __builtin_fma(k1, b, k0 * a)));
  }
  // x -> balanced residue, |x| < 2^53
  TFHE_HD static elem reduce(elem x) { return __builtin_fma(-__builtin_rint(x * PINV), P, x); }
  TFHE_HD static elem from_digit(u32 d) { return (double)(i32)d; }
  // signed 16-bit halves of the key word taken as a signed 32-bit integer
  TFHE_HD static elem from_key_word(u32 w, int part) {
    const i32 lo = (i32)(int16_t)(w & 0xFFFFu);
    if (part == 0) return (double)lo;
    // exact: the difference is a multiple of 2^16; taken in wrapping u32 (w = 0x7FFFxxxx with a
    // negative low half gives 2^31, i.e. hi = -32768: |hi| <= 2^15 and lo + 2^16 hi = w mod 2^32)
    return (double)((i32)(w - (u32)lo) >> 16);
  }
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
  TFHE_HD static u32 finish(const elem (&parts)[kParts]) {
    return to_u32(reduce(parts[0])) + (to_u32(reduce(parts[1])) << 16);
  }

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
  // out: n + 18 elements (wave_ntt.h::ntt_twiddle_words)
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
    // radix8_small_v: quarter q of stage 3 has twiddle psi_rev[4+q]; its v leg is radix-4 output q,
    //   q=0: a + w2a b + w1 c + w12a d      q=1: a - w2a b + w1 c - w12a d
    //   q=2: a + w2b b - w1 c - w12b d      q=3: a - w2b b - w1 c + w12b d
    for (int q = 0; q < 4; ++q) {
      const u64 w3 = raw[4 + q], w2 = raw[q < 2 ? 2 : 3];
      const u64 w12 = mulmod_u64(raw[1], w2);
      const bool neg_b = q & 1, neg_c = q >= 2, neg_d = (q == 1 || q == 2);
      const u64 kb = mulmod_u64(w3, w2), kc = mulmod_u64(w3, raw[1]), kd = mulmod_u64(w3, w12);
      out[n + 2 + 4 * q + 0] = balanced(w3);
      out[n + 2 + 4 * q + 1] = balanced(neg_b ? (P_INT - kb) % P_INT : kb);
      out[n + 2 + 4 * q + 2] = balanced(neg_c ? (P_INT - kc) % P_INT : kc);
      out[n + 2 + 4 * q + 3] = balanced(neg_d ? (P_INT - kd) % P_INT : kd);
    }
  }
  static inline elem n_inv(int logn) { return balanced(powmod_u64((u64)1 << logn, P_INT - 2)); }
  static inline double exact_bits() { return 40.9; }  // |t| < p/2 = 2^41 with margin
  static inline double key_bits() { return 15.0; }    // each half is <= 2^15 in magnitude
};

}  // namespace tfhe

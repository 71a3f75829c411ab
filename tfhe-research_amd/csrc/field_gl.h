// field_gl.h -- field policy: Goldilocks prime p = 2^64 - 2^32 + 1, elements u64 (goldilocks.h).
//   GlField      (PARTS = 1): one spectrum per key polynomial; exact when R*N*B*2^32 < 2^62.
//   GlSplitField (PARTS = 2): the key word is split into signed 16-bit halves like in the fp64 field;
//                exact when R*N*B*2^15 < 2^62, i.e. for every decomposition base the reference can
//                express (log_base <= 31) at N <= 2048.  The last resort of the backend selection.
#pragma once
#include "goldilocks.h"

namespace tfhe {

template <int PARTS>
struct GlFieldT {
  typedef u64 elem;
  static constexpr int kParts = PARTS;  // spectra per bootstrapping-key polynomial
  static constexpr int kId = PARTS == 1 ? 1 : 3;
  // hooks of the lazily-reduced fp64 fields; canonical arithmetic never needs them
  static constexpr int kInverseSweepEvery = 0;
  static constexpr bool kReduceSpectrum = false;
  template <int E>
  static constexpr bool split_accum() { return false; }  // field_fp49.h / field_fp.h: (h, l) accumulator pairs
  TFHE_HD static elem accum_init() { return 0; }
  TFHE_HD static void mac(elem, elem, elem&, elem&) {}
  TFHE_HD static elem mac_finish(elem a, elem) { return a; }
  static constexpr int kMaxRows = 1 << 20;
  TFHE_HD static elem reduce(elem a) { return a; }

  TFHE_HD static elem zero() { return 0; }
  TFHE_HD static elem add(elem a, elem b) { return gl::add(a, b); }
  TFHE_HD static elem sub(elem a, elem b) { return gl::sub(a, b); }
  TFHE_HD static elem mul(elem a, elem w) { return gl::mul(a, w); }
  static constexpr int kSmallBits = 31;  // mul_small is the full product here: any digit qualifies
  static constexpr int kMaxLogBase = 31;
  TFHE_HD static elem mul_small(elem a, elem w) { return gl::mul(a, w); }
  static constexpr bool kFuseFirstTwo = false;  // canonical u64 arithmetic gains nothing from it
  TFHE_HD static void radix4_small(elem&, elem&, elem&, elem&, elem, elem, elem, elem, elem) {}
  TFHE_HD static elem radix8_small_v(elem, elem, elem, elem, elem, elem, elem, elem) { return 0; }
  // gadget digit (wrapped u32 holding a small signed integer) -> field element
  TFHE_HD static elem from_digit(u32 d) { return gl::from_i32(d); }
  // key word -> field element of spectrum `part`
  TFHE_HD static elem from_key_word(u32 w, int part) {
    if (PARTS == 1) return (elem)w;
    // signed 16-bit halves of the word taken as a signed 32-bit integer: w = lo + 2^16 hi (mod 2^32)
    const i32 lo = (i32)(int16_t)(w & 0xFFFFu);
    // (w - lo) in wrapping u32: for w = 0x7FFFxxxx with a negative low half the difference is 2^31,
    // i.e. hi = -32768 -- still |hi| <= 2^15 and lo + 2^16 hi = w (mod 2^32)
    return gl::from_i32((u32)(part == 0 ? lo : ((i32)(w - (u32)lo) >> 16)));
  }
  // called on every accumulator before the inverse transform
  TFHE_HD static elem before_inverse(elem a) { return a; }
  // inverse-transform outputs of all parts -> value mod 2^32
  TFHE_HD static u32 finish(const elem (&parts)[kParts]) {
    u32 v = gl::lift_mod_2_32(parts[0]);
    if (PARTS == 2) v += gl::lift_mod_2_32(parts[PARTS - 1]) << 16;
    return v;
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

  // ---- host-side constants ----
  // out: n + 18 elements (wave_ntt.h::ntt_twiddle_words)
  static inline void fill_twiddles(int logn, elem* out) {
    const int n = 1 << logn;
    const u64 psi = gl::root_of_unity(logn + 1);
    u64 pw = 1;
    for (int k = 0; k < n; ++k) {
      int rev = 0;
      for (int b = 0; b < logn; ++b) rev |= ((k >> b) & 1) << (logn - 1 - b);
      out[rev] = pw;  // psi_rev[j] = psi^bitrev(j)
      pw = gl::mul(pw, psi);
    }
    out[n] = gl::mul(out[1], out[2]);
    out[n + 1] = gl::mul(out[1], out[3]);
    for (int i = 2; i < 18; ++i) out[n + i] = 0;  // fused-stage constants of the 42-bit fp64 field: unused here
  }
  static inline elem n_inv(int logn) { return gl::inv((u64)1 << logn); }
  // log2 of the largest |integer convolution value| this field lifts exactly
  static inline double exact_bits() { return 62.0; }
  // log2 of the magnitude of one key operand as seen by the convolution
  static inline double key_bits() { return PARTS == 1 ? 32.0 : 15.0; }
};

typedef GlFieldT<1> GlField;
typedef GlFieldT<2> GlSplitField;

}  // namespace tfhe

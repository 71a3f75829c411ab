// pbs_wave.h -- per-wavefront bodies of the bootstrapping hot path (host/device source).
//
// One 64-lane wavefront owns one LWE sample for the whole blind rotation: the GLWE accumulator
// stays in wave-private LDS for all n CMUX iterations, nothing is exchanged with other waves, and
// the only global traffic inside the loop is the (batch-shared, L2/Infinity-Cache resident)
// NTT-domain bootstrapping key.  The same bodies are compiled by g++ for the SIMT emulator that
// the CPU tests use (tests/emu), so what is parity-tested on the CPU is the code that runs on the
// GPU.
//
// Reference behaviour restated here (file:line in /root/reference/src):
//   switch_modulus            utils.rs:13-33
//   encode + trivial encrypt  glwe.rs:141-151, 232-243
//   Monomial multiply         glwe.rs:20-34, utils.rs:183-207
//   round_value / decompose   decomposer.rs:27-80 (digits at bit offsets log_base*l from bit 0,
//                             a limb equal to B keeps B and emits no carry)
//   decompose_glwe_ciphertext glwe.rs:90-108 (row = poly*levels + level, level 0 = MSB)
//   external_product / cmux   ggsw.rs:132-178
//   blind-rotation loop       bootstrapping.rs:79-105
//   sample_extract (index 0)  bootstrapping.rs:122-156
#pragma once
#include "wave_ntt.h"

namespace tfhe {

template <int V>
struct IntC {
  static constexpr int value = V;
};

template <int I, int END, class F>
TFHE_HD void static_for(F&& f) {
  if constexpr (I < END) {
    f(IntC<I>{});
    static_for<I + 1, END>(f);
  }
}

// Plain-old-data view of the parameter set the kernels need (derived once on the host).
struct PbsParams {
  u32 n;            // LWE dimension
  u32 k;            // GLWE dimension
  u32 log_n;        // log2 of the ring degree
  u32 tv_shift;     // 32 - log_p - padding_bits (glwe.rs:145)
  u32 log_base;     // PBS decomposer
  u32 levels;
  u32 ignored_bits;  // 32 - log_base*levels        (decomposer.rs:28)
  u32 first_shift;   // log_base * (floor(32/log_base) - levels): bit offset of the lowest kept limb
};

struct KsParams {
  u32 log_base;
  u32 levels;
  u32 ignored_bits;
  u32 first_shift;
};

// decomposer.rs:27-40
TFHE_HD u32 round_value(u32 v, u32 ignored_bits) {
  if (ignored_bits == 0) return v;
  return ((v >> ignored_bits) + ((v >> (ignored_bits - 1)) & 1u)) << ignored_bits;
}

// One limb of decomposer.rs:53-65.  `v` is already rounded.  Returns the digit as a wrapped u32
// and updates the carry (0/1).  Limbs below first_shift are zero after rounding, so the chain can
// start at first_shift with carry 0.
TFHE_HD u32 decompose_limb(u32 v, u32 shift, u32 log_base, u32& carry) {
  const u32 res = ((v >> shift) & ((1u << log_base) - 1u)) + carry;
  const u32 carry_mask = res & (1u << (log_base - 1));
  carry = carry_mask >> (log_base - 1);
  return res - (carry_mask << 1);
}

// utils.rs:13-33 with log_from = 32: round(v * 2N / 2^32) mod 2N
TFHE_HD u32 switch_modulus_2n(u32 v, u32 log_n) {
  const u32 sh = 32u - (log_n + 1u);
  return ((v >> sh) + ((v >> (sh - 1u)) & 1u)) & ((2u << log_n) - 1u);
}

// coefficient j of X^m * poly, m in [0, 2N) (utils.rs:183-207)
template <int LOGN>
TFHE_HD u32 monomial_coeff(const u32* poly, int j, u32 m) {
  constexpr int N = 1 << LOGN;
  const int deg = (int)(m & (N - 1));
  const u32 flip = (m >> LOGN) & 1u;
  const u32 v = poly[(j - deg) & (N - 1)];
  const u32 negate = flip ^ (u32)(j < deg);
  return negate ? (0u - v) : v;
}

// ---------------------------------------------------------------------------------------------
// GGSW (NTT domain) x GLWE external product, accumulated in the NTT domain.
//   src(p, j)  -> coefficient j of polynomial p of the GLWE operand (a functor, lane-local)
//   ggsw       -> prepared GGSW: [R][K+1] spectra of N u64 in spectrum_slot order, pre-scaled by
//                 N^-1 (so the unscaled inverse NTT below lands on the true product)
//   out(p, j, value mod 2^32) is called once per output coefficient.
// ---------------------------------------------------------------------------------------------
template <int LOGN, int K, class Ctx, class Src, class Out>
TFHE_HD void external_product_wave(const Ctx& c, const PbsParams& P, const u64* ggsw, Src src,
                                   Out out) {
  constexpr int E = NttShape<LOGN>::kE;
  constexpr int N = 1 << LOGN;
  const int lane = c.lane();

  u64 accum[K + 1][E];
#pragma unroll
  for (int col = 0; col <= K; ++col)
#pragma unroll
    for (int r = 0; r < E; ++r) accum[col][r] = 0;

#pragma unroll 1
  for (int p = 0; p <= K; ++p) {
    u32 v[E];
#pragma unroll
    for (int r = 0; r < E; ++r) v[r] = round_value(src(p, r * 64 + lane), P.ignored_bits);
    u32 carry_bits = 0;  // bit r = carry of coefficient r
#pragma unroll 1
    for (u32 t = 0; t < P.levels; ++t) {  // limbs LSB -> MSB; level index counts from the MSB
      const u32 level = P.levels - 1 - t;
      const u32 shift = P.first_shift + P.log_base * t;
      u64 work[E];
#pragma unroll
      for (int r = 0; r < E; ++r) {
        u32 carry = (carry_bits >> r) & 1u;
        const u32 digit = decompose_limb(v[r], shift, P.log_base, carry);
        carry_bits = (carry_bits & ~(1u << r)) | (carry << r);
        work[r] = gl::from_i32(digit);
      }
      ntt_forward<LOGN>(c, work);
      const u64* row = ggsw + (size_t)(p * P.levels + level) * (K + 1) * N;
#pragma unroll
      for (int col = 0; col <= K; ++col) {
        const u64* spec = row + (size_t)col * N;
#pragma unroll
        for (int r = 0; r < E; ++r)
          accum[col][r] = gl::add(accum[col][r], gl::mul(work[r], spec[spectrum_slot<LOGN>(lane, r)]));
      }
    }
  }

  // compile-time loop: the body is too large for `#pragma unroll`, and a rolled loop would index
  // accum[] dynamically and push it to scratch memory
  static_for<0, K + 1>([&](auto col_c) {
    constexpr int col = decltype(col_c)::value;
    ntt_inverse<LOGN>(c, accum[col]);
#pragma unroll
    for (int r = 0; r < E; ++r) out(col, r * 64 + lane, gl::lift_mod_2_32(accum[col][r]));
  });
}

// ---------------------------------------------------------------------------------------------
// Blind rotation of ONE LWE sample (bootstrapping.rs:67-105), accumulator in c.acc() (LDS,
// (K+1) x N u32, natural order).  On return c.acc() holds the final GLWE accumulator.
// ---------------------------------------------------------------------------------------------
template <int LOGN, int K, class Ctx>
TFHE_HD void blind_rotate_wave(const Ctx& c, const PbsParams& P, const u32* lwe /* n+1 */,
                               const u32* tv /* N, un-encoded */, const u64* bsk /* prepared */) {
  constexpr int E = NttShape<LOGN>::kE;
  constexpr int N = 1 << LOGN;
  const int lane = c.lane();
  u32* acc = c.acc();

  // acc = X^{-b~} * (0, ..., 0, tv << tv_shift)
  {
    const u32 b_tilde = switch_modulus_2n(lwe[P.n], LOGN);
    const u32 m = (2u * N - b_tilde) & (2u * N - 1u);
    const int deg = (int)(m & (N - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int j = r * 64 + lane;
#pragma unroll
      for (int p = 0; p < K; ++p) acc[p * N + j] = 0;
      const u32 t = tv[(j - deg) & (N - 1)] << P.tv_shift;
      acc[K * N + j] = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
    }
    c.sync();
  }

  const size_t ggsw_words = (size_t)(K + 1) * P.levels * (K + 1) * N;
#pragma unroll 1
  for (u32 i = 0; i < P.n; ++i) {
    const u32 a_tilde = c.uniform(switch_modulus_2n(lwe[i], LOGN));
    // X^0 * acc - acc = 0: every digit is zero and the CMUX returns acc unchanged
    if (a_tilde == 0) continue;
    // cmux(ggsw_i, acc, X^{a~} * acc) = external_product(ggsw_i, X^{a~} acc - acc) + acc
    auto src = [&](int p, int j) -> u32 {
      return monomial_coeff<LOGN>(acc + p * N, j, a_tilde) - acc[p * N + j];
    };
    // all rotated reads happen before the first inverse NTT, so the in-place update is safe
    auto out = [&](int p, int j, u32 value) { acc[p * N + j] += value; };
    external_product_wave<LOGN, K>(c, P, bsk + (size_t)i * ggsw_words, src, out);
    c.sync();
  }
}

// sample_extract at index 0 (bootstrapping.rs:122-156) from the LDS accumulator
template <int LOGN, int K, class Ctx>
TFHE_HD void sample_extract_wave(const Ctx& c, u32* out /* K*N + 1 */) {
  constexpr int E = NttShape<LOGN>::kE;
  constexpr int N = 1 << LOGN;
  const int lane = c.lane();
  const u32* acc = c.acc();
#pragma unroll
  for (int p = 0; p < K; ++p)
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int x = r * 64 + lane;
      out[p * N + x] = (x == 0) ? acc[p * N] : (0u - acc[p * N + N - x]);
    }
  if (lane == 0) out[K * N] = acc[K * N];
}

// Forward NTT of one u32 polynomial of the bootstrapping key into the prepared layout,
// pre-scaled by N^-1.
template <int LOGN, class Ctx>
TFHE_HD void bsk_prepare_wave(const Ctx& c, const u32* poly, u64* spec, u64 n_inv) {
  constexpr int E = NttShape<LOGN>::kE;
  const int lane = c.lane();
  u64 x[E];
#pragma unroll
  for (int r = 0; r < E; ++r) x[r] = (u64)poly[r * 64 + lane];
  ntt_forward<LOGN>(c, x);
#pragma unroll
  for (int r = 0; r < E; ++r) spec[spectrum_slot<LOGN>(lane, r)] = gl::mul(x[r], n_inv);
}

}  // namespace tfhe

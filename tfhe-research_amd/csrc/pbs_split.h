// pbs_split.h -- EXPERIMENT (round 3, -DTFHE_SPLIT_N1024=1; not the shipped path): a blind rotation whose team of
// 2 (K+1) waves rotates TWO samples with the registers of the one-sample kernel.
//
// The one-wave-per-polynomial kernels of the complex transform at N = 1024 hold 8 elements per lane: the two key parts'
// accumulators are 64 of their 234 registers and a second sample's set does not fit (pbs_wave.h, two samples per
// team).  Here wave (c, q) splits the work the other way:
//   forward role   it decomposes and transforms polynomial c of SAMPLE q (l transforms per CMUX, as before),
//   product role   it accumulates output column c for key PART q only, but for BOTH samples (2 x 1 accumulator sets:
//                  the same 64 registers), reading every key chunk of its part once for two products,
//   inverse role   it inverse-transforms its part of column c for both samples (two transforms, as before) and adds
//                  to_u32(part) << (16 q) into the accumulator polynomial (c, sample) in LDS with ds_add_u32 -- the two
//                  parts of a word arrive from two waves, wrapping addition commutes.
// Per wave and CMUX iteration: the arithmetic of the shipped kernel, half its key bytes; per team: 4 waves at k = 1 meet
// at every barrier instead of 2.  Whether the halved key stream pays for the wider barriers is what the experiment
// measures (profiles/r03_kernel_ab.txt).
#pragma once
#include "pbs_wave.h"

namespace tfhe {

// Ctx: group() = 2 c + q; scratch() / scratch_of(g) = the 8 N-byte buffer of group g; acc(s) = accumulator polynomial c
// of sample s (owned -- initialised, rotated and read -- by wave (c, s), added to by waves (c, 0) and (c, 1)).
template <class F, int LOGN, int K, class Ctx>
TFHE_HD void blind_rotate_team_split(const Ctx& c, const PbsParams& P, const u32* const* lwe /* 2 x (n+1) */,
                                     const u32* const* tv /* 2 x N, un-encoded */, const typename F::elem* bsk) {
  typedef typename F::elem elem;
  static_assert(F::kParts == 2 && F::kLogShrink == 1 && F::kCoeffs == 2, "the complex transform");
  constexpr int G = 1;
  constexpr int LT = LOGN - 1;
  constexpr int E = NttShape<LT, G>::kE;
  constexpr int EC = 2 * E;
  constexpr int T = 64;
  constexpr int M = 1 << LT;     // transform elements per polynomial
  constexpr int NC = 1 << LOGN;  // ring coefficients per polynomial
  const int lane = c.tid();
  const int me = c.group() >> 1;   // polynomial / output column
  const int mine = c.group() & 1;  // the sample I transform = the key part I accumulate
  u32* acc = c.acc(mine);

  {  // acc = X^{-b~} * (0, ..., 0, tv << tv_shift) of my sample: only the body polynomial is non-zero
    const u32 b_tilde = switch_modulus_2n(lwe[mine][P.n], LOGN);
    const u32 m = (2u * NC - b_tilde) & (2u * NC - 1u);
    const int deg = (int)(m & (NC - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < EC; ++r) {
      const int j = r * T + lane;
      u32 val = 0;
      if (me == K) {
        const u32 t = tv[mine][(j - deg) & (NC - 1)] << P.tv_shift;
        val = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
      }
      acc[j] = val;
    }
  }
  c.team_sync();

  const size_t ggsw_words = (size_t)(K + 1) * P.levels * (K + 1) * F::kParts * M;  // elements per GGSW
  const RoundConsts rc = round_consts(P.ignored_bits);
  constexpr int CH = 4;  // elements per key chunk (64 bytes per lane and chunk)
  constexpr int PIECES = E / CH;
  constexpr int CHUNKS = (K + 1) * PIECES;  // per level: source polynomial, piece of its spectrum (my part only)
#pragma unroll 1
  for (u32 i = 0; i < P.n; ++i) {
    const elem* ggsw = bsk + (size_t)i * ggsw_words;
    const u32 a_tilde = c.uniform(switch_modulus_2n(lwe[mine][i], LOGN));
    TopConsts<F, LT, G, true> ftop;
    ftop.issue(c.twiddles_uniform());
    u32 v[EC];
#pragma unroll
    for (int r = 0; r < EC; ++r) {
      const int j = r * T + lane;
      v[r] = round_value_fast(monomial_coeff<LOGN>(acc, j, a_tilde) - acc[j], rc);
    }
    ftop.ready();
    elem accum[2][E];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int r = 0; r < E; ++r) accum[s][r] = F::zero();

    auto tile_ptr = [&](u32 level, int sp) -> const elem* {
      return ggsw + (((size_t)(sp * P.levels + level) * (K + 1) + me) * F::kParts + mine) * M;
    };
    elem kbuf[2][CH];
    auto load_chunk = [&](u32 level, auto ci_c, int buf) {
      constexpr int ci = decltype(ci_c)::value;
      constexpr int r0 = (ci % PIECES) * CH, sp = ci / PIECES;
      const elem* tile = tile_ptr(level, sp);
#pragma unroll
      for (int r = 0; r < CH; ++r) kbuf[buf][r] = tile[spectrum_slot<LT, G, (int)sizeof(elem)>(lane, r0 + r)];
    };
#pragma unroll 1
    for (u32 t = 0; t < P.levels; ++t) {  // limbs LSB -> MSB; level index counts from the MSB
      const u32 level = P.levels - 1 - t;
      const u32 shift = P.first_shift + P.log_base * t;
      load_chunk(level, IntC<0>{}, 0);
      c.compiler_fence();
      {
        elem work[E];
        const u32 carry_width = (t == 0) ? 0u : 1u;
#pragma unroll
        for (int r = 0; r < E; ++r) {
          u32 dg[2];
          dg[0] = decompose_limb_fast<true>(v[r], shift, P.log_base, carry_width);
          dg[1] = decompose_limb_fast<true>(v[r + E], shift, P.log_base, carry_width);
          work[r] = F::from_digits(dg);
        }
        ntt_forward<F, LT, G, true, true>(c, work, ftop);
        elem* pub = c.scratch();
#pragma unroll
        for (int r = 0; r < E; ++r) pub[exchange_slot<LT, G>(lane, r)] = work[r];
      }
      c.team_sync();
      elem d[2][CH];
      static_for<0, CHUNKS>([&](auto ci_c) {
        constexpr int ci = decltype(ci_c)::value;
        constexpr int r0 = (ci % PIECES) * CH, sp = ci / PIECES;
        constexpr int cur = ci % 2;
        if constexpr (ci + 1 < CHUNKS) load_chunk(level, IntC<ci + 1>{}, (ci + 1) % 2);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const elem* spec = c.scratch_of(2 * sp + s);  // the spectrum wave (sp, s) published
#pragma unroll
          for (int r = 0; r < CH; ++r) d[s][r] = spec[exchange_slot<LT, G>(lane, r0 + r)];
        }
        c.compiler_fence();
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int r = 0; r < CH; ++r) accum[s][r0 + r] = F::mul_add(d[s][r], kbuf[cur][r], accum[s][r0 + r]);
      });
      c.team_sync();  // everyone is done reading before the next transform reuses the buffers
    }

    TopConsts<F, LT, G, false> itop;
    itop.issue(c.twiddles_uniform());
    itop.ready();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      ntt_inverse<F, LT, G>(c, accum[s], itop);
      u32* dst = c.acc(s);
#pragma unroll
      for (int r = 0; r < E; ++r) {
        // my part of coefficients r and r + E of column `me` of sample s (FftField::finish: lo + (hi << 16))
        c.lds_add(dst + r * T + lane, F::to_u32(accum[s][r].re) << (16 * mine));
        c.lds_add(dst + (r + E) * T + lane, F::to_u32(accum[s][r].im) << (16 * mine));
      }
    }
    c.team_sync();  // both parts of every word are in before anybody rotates the accumulators again
  }
}

}  // namespace tfhe

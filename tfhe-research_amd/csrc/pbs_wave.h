// pbs_wave.h -- per-wavefront bodies of the bootstrapping hot path (host/device source).
//
// A team of K+1 wave groups (one workgroup) owns one LWE sample for the whole blind rotation: group c
// keeps polynomial c of the GLWE accumulator in LDS for all n CMUX iterations, the groups exchange
// only digit spectra through LDS, and the only global traffic inside the loop is the batch-shared,
// L2/Infinity-Cache resident NTT-domain bootstrapping key.  The same bodies are compiled by g++ for
// the SIMT emulator that the CPU tests use (tests/emu), so what is parity-tested on the CPU is the
// code that runs on the GPU.
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

// Shape the prepared bootstrapping key of a parameter set is laid out for (wave_ntt.h::spectrum_slot, LAYOUT_E): 0 = the
// team kernel's own shape.  N = 512 with k = 1 in the complex transform: the PAIR kernel's (both polynomials of a sample in
// one wavefront, 32 lanes x 8 elements each) -- it is that shape's throughput kernel; the team / wide / standalone
// external-product kernels of the shape read the same key through the same slot formula.
template <class F, int LOGN, int K>
constexpr int key_layout_e() {
  return (F::kLogShrink == 1 && LOGN == 9 && K == 1) ? 8 : 0;
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

// The same rounding with its two constants hoisted (both wave-uniform): half = 2^(ignored_bits - 1) and
// keep = ~(2^ignored_bits - 1), or 0 and ~0 when nothing is ignored.  (v + half) & keep is the literal formula: with
// v = q 2^ig + r the bit it adds is [r >= half], and both forms wrap to 0 mod 2^32 in the same case.  Two
// instructions per coefficient instead of four and a select.
struct RoundConsts {
  u32 half, keep;
};
TFHE_HD RoundConsts round_consts(u32 ignored_bits) {
  return ignored_bits == 0 ? RoundConsts{0u, ~0u} : RoundConsts{1u << (ignored_bits - 1), ~((1u << ignored_bits) - 1u)};
}
TFHE_HD u32 round_value_fast(u32 v, RoundConsts rc) { return (v + rc.half) & rc.keep; }

// One limb of decomposer.rs:53-65.  `v` is already rounded.  Returns the digit as a wrapped u32
// and updates the carry (0/1).  Limbs below first_shift are zero after rounding, so the chain can
// start at first_shift with carry 0.
TFHE_HD u32 decompose_limb(u32 v, u32 shift, u32 log_base, u32& carry) {
  const u32 res = ((v >> shift) & ((1u << log_base) - 1u)) + carry;
  const u32 carry_mask = res & (1u << (log_base - 1));
  carry = carry_mask >> (log_base - 1);
  return res - (carry_mask << 1);
}

// The same limb for the hot loop, 7 instructions instead of 11, no state besides `v` itself: the
// carry travels between limbs as bit log_base-1 of v -- a bit of the limb just consumed (or, before the
// first limb, a bit that `carry_width` = 0 hides), dead from then on.  The limb is one bit-field
// extract, the carry-in another (width `carry_width`: 0 for the lowest limb, 1 after), the carry-out
// is written back by one bit-field insert of res's bit log_base-1, digit = res - 2 * (res & B/2) is
// one multiply-add.
// SMALL_BASE: log_base <= 23, so that res & B/2 fits the signed 24-bit multiply-add (wave-uniform; the
// caller branches once per level).
template <bool SMALL_BASE>
TFHE_HD u32 decompose_limb_fast(u32& v, u32 shift, u32 log_base, u32 carry_width) {
  const u32 half = 1u << (log_base - 1);
#if defined(__HIP_DEVICE_COMPILE__)
  const u32 res = __builtin_amdgcn_ubfe(v, shift, log_base) + __builtin_amdgcn_ubfe(v, log_base - 1, carry_width);
  const u32 hb = res & half;
  u32 digit;
  if (SMALL_BASE) {
    // hb < 2^23: the signed 24-bit multiply-add is exact (its addend is a full 32-bit word)
    asm("v_mad_i32_i24 %0, %1, -2, %2" : "=v"(digit) : "v"(hb), "v"(res));
  } else {
    digit = res - (hb << 1);
  }
  asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(v) : "s"(half), "v"(res), "v"(v));  // v = (res & half) | (v & ~half)
  return digit;
#else
  const u32 cin = carry_width ? ((v >> (log_base - 1)) & 1u) : 0u;
  const u32 res = ((v >> shift) & ((1u << log_base) - 1u)) + cin;
  const u32 hb = res & half;
  v = hb | (v & ~half);
  return res - (hb << 1);
#endif
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
// GGSW (NTT domain) x GLWE external product by a TEAM of K+1 wave groups (one workgroup).  A group
// is G wavefronts that share one polynomial (G = 1 except for N = 2048); "wave c" below means group
// c (c = ctx.group(), 0..K), which owns GLWE polynomial c on the input side and output column c:
//   - it decomposes polynomial c and forward-transforms its `levels` digit rows,
//   - publishes each spectrum in its LDS transpose buffer (free between two transforms),
//   - after a workgroup barrier every wave multiplies ALL K+1 published spectra of that level with
//     the key spectra of ITS column and accumulates in registers (F::kParts accumulators of E
//     elements -- independent of K),
//   - finally inverse-transforms its own column.
// So a wave does levels forward + kParts inverse transforms and (K+1)*levels*kParts MAC tiles, the
// accumulators are (K+1)x smaller than with one wave per sample, and nothing but spectra crosses
// waves.  Two workgroup barriers per level (publish -> consume -> reuse of the buffer).
//
//   src(j)  -> coefficient j of polynomial c of the GLWE operand (functor, lane-local)
//   ggsw    -> prepared GGSW: [R][K+1][F::kParts] spectra of N field elements (N/2 complex ones) in spectrum_slot order,
//              pre-scaled by N^-1 (so the unscaled inverse NTT lands on the true product)
//   out(j, value mod 2^32) is called once per coefficient of output polynomial c.
// Every wave of the team must call this the same number of times (it contains barriers).
//
// external_product_team_keys multiplies ONE GLWE operand with KEYS prepared GGSWs (ggsw, ggsw +
// ggsw_words, ...): the decomposition and the forward transforms are shared, only the
// multiply-accumulate and the inverse transforms are per key (the unrolled blind rotation of
// notes/BMMP Bootstrapping.md needs three products of the same accumulator).  out(m, j, value) is
// called for key m = 0 .. KEYS-1 in turn, end_of_key(m) once after the last coefficient of key m.
// ---------------------------------------------------------------------------------------------
#ifndef TFHE_TOP_PREFETCH
#define TFHE_TOP_PREFETCH 1
#endif
// NS > 1: the team multiplies NS GLWE operands (NS independent samples of a batch) with the SAME prepared GGSW in
// one pass.  Every key chunk is fetched from L2 once and used NS times, the NS transforms of a step run in lockstep
// through one set of transposes and barriers (ntt_forward_multi), and a team barrier covers NS products instead
// of one -- the kernels whose waves wait on the key stream and on each other (twelve-wave teams at N = 2048; many
// digit rows at N = 512) spend half as much of both per product.  Sample s works in exchange buffer s
// (Ctx::with_exchange_buffer) and has its own accumulators; src(s, j) / out(m, s, j, value) name the sample.
// KEYS > 1 (the unrolled blind rotation) keeps NS = 1.
template <class F, int LOGN, int K, int G, int KEYS, int NS, class Ctx, class Src, class Out, class EndKey>
TFHE_HD void external_product_team_multi(const Ctx& c, const PbsParams& P, const typename F::elem* ggsw,
                                         size_t ggsw_words, Src src, Out out, EndKey end_of_key) {
  typedef typename F::elem elem;
  static_assert(KEYS == 1 || NS == 1, "several keys or several samples, not both");
  // LT: log2 of the transform size -- the ring degree in the prime fields, half of it for the complex
  // transform, whose elements hold the coefficient pairs (j, j + N/2) (field_fft.h).  A lane holds E
  // transform elements and EC = E * F::kCoeffs ring coefficients per array; element r of a lane pairs the
  // coefficients r and r + E of its coefficient array (indices (r + q E) T + lane).
  constexpr int LT = LOGN - F::kLogShrink;
  constexpr int E = NttShape<LT, G>::kE;
  constexpr int EC = NttShape<LOGN, G>::kE;
  constexpr int CO = F::kCoeffs;
  static_assert(EC == E * CO, "coefficients per lane");
  constexpr int T = NttShape<LOGN, G>::kThreads;  // threads per polynomial
  constexpr int N = 1 << LT;                       // transform elements per polynomial
  constexpr int PARTS = F::kParts;
  const int lane = c.tid();   // thread index inside my polynomial's group of G waves
  const int me = c.group();   // polynomial / output column owned by my group
  // (h, l) accumulator pairs only for a single key and sample: a second set per accumulator does not fit otherwise
  constexpr bool SPLIT = F::template split_accum<E>() && KEYS == 1 && NS == 1;
  constexpr int ACCS = KEYS * PARTS;  // accumulator a = m * PARTS + q of a sample: key m, key part q

  elem accum[NS][ACCS][E];
  // second accumulator set of fields that add up unreduced (h, l) product pairs (F::split_accum)
  elem accum_lo[SPLIT ? ACCS : 1][SPLIT ? E : 1];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int q = 0; q < ACCS; ++q)
#pragma unroll
      for (int r = 0; r < E; ++r) {
        accum[s][q][r] = SPLIT ? F::accum_init() : F::zero();
        if (SPLIT) accum_lo[SPLIT ? q : 0][SPLIT ? r : 0] = F::zero();
      }

  // v[s][r]: rounded coefficient; once a limb has been consumed its bit log_base-1 carries the digit
  // chain's carry to the next limb (decompose_limb_fast)
  // The lane-uniform constants of the forward transforms' top window (wave_ntt.h::TopConsts) are fetched
  // while the operand is read and rounded, stay in scalar registers for all levels, and make room for
  // the inverse transforms' block after the last level.
  TopConsts<F, LT, G, true> ftop;
  if constexpr (TFHE_TOP_PREFETCH) ftop.issue(c.twiddles_uniform());
  u32 v[NS][EC];
  const RoundConsts rc = round_consts(P.ignored_bits);
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int r = 0; r < EC; ++r) v[s][r] = round_value_fast(src(s, r * T + lane), rc);
  if constexpr (TFHE_TOP_PREFETCH) ftop.ready();

  // Key tiles of one level for my column: idx = s * PARTS + q, s = source polynomial 0..K, q = part.
  // A tile is consumed in chunks of CH registers (order: s, piece of the spectrum, q); chunks are staged
  // through two register buffers: the loads of chunk i+1 are issued before the arithmetic of chunk
  // i, and chunk 0 of a level is loaded before that level's forward transform, so no key load sits
  // on the critical path.  CH = 8 keeps the staging at 3 x 16 VGPRs whatever E is (16 measured the
  // same speed with more spill: profiles/r01_chunk_ab.txt).
#ifndef TFHE_CHUNK
#define TFHE_CHUNK 8
#endif
  constexpr int TILES = (K + 1) * ACCS;
  // (NS samples share a chunk: half the registers per chunk give the same arithmetic per fetch, and the staging
  // buffers and the digit-spectrum pieces of NS samples fit beside NS accumulator sets)
#ifndef TFHE_NS_CHUNK_DIV
#define TFHE_NS_CHUNK_DIV 2
#endif
  constexpr int CH_MAX = TFHE_CHUNK * 8 / (int)sizeof(elem) / (NS > 1 ? TFHE_NS_CHUNK_DIV : 1);  // TFHE_CHUNK counts 8-byte registers
  constexpr int CH = E < CH_MAX ? E : CH_MAX;
  constexpr int CHUNKS = TILES * (E / CH);
  // tile of source polynomial sp and accumulator a = (key m, part q)
  // (TFHE_PROBE_HOT_KEY: dev_switches.h -- a WRONG-BITS timing probe, TFHE_DEV_BUILD only)
  auto tile_ptr = [&](u32 level, int sp, int a) -> const elem* {
    const int m = a / PARTS, q = a % PARTS;
    if (TFHE_PROBE_HOT_KEY) return ggsw;
    return ggsw + (size_t)m * ggsw_words + (((size_t)(sp * P.levels + level) * (K + 1) + me) * PARTS + q) * N;
  };
  // Spectrum exchange through LDS.  With ONE buffer per group a level needs two team barriers
  // (publish -> consume -> the next transform reuses the buffer).  With TWO buffers (and one sample) level t works
  // in buffer t & 1: whoever still reads buffer (t-1) & 1 is not disturbed, and buffer t & 1 was last
  // read in the MAC of level t-2, which every wave left before the barrier of level t-1 -- one
  // barrier per level, plus one at the end of the product (the first level of the next product
  // writes buffer 0 again).  The inverse transforms run in buffer levels & 1 under the same rule.
  // With NS samples buffer s belongs to sample s and every level has its two barriers -- for NS products.
  const bool two = NS == 1 && c.exchange_buffers() == 2;
#ifndef TFHE_KEY_BUFFERS
#define TFHE_KEY_BUFFERS 2
#endif
  // staging buffers: chunk i + NB - 1 is fetched while chunk i is consumed.  Two everywhere (three and four measured
  // nothing at one sample per team, profiles/r02_kernel_ab.txt).  The half-size chunks of two samples per team with one
  // wave per polynomial had a third while the key came from beyond the L2s (cfg3 in one launch over the whole key:
  // 68.6 -> 63.8 ms; a fourth 65.9; at N = 2048 the third costs 31 more spilled registers: 57.3 -> 63.2 ms); with the key
  // blocked for the L2s (kernels.hip::blind_rotate_plan) two are enough and spill nothing: cfg3 53.6 -> 52.8 ms
  // (three: 10 spilled registers, four: 54.1 ms; profiles/r03_kernel_ab.txt)
#ifndef TFHE_KEY_BUFFERS_NS
#define TFHE_KEY_BUFFERS_NS 2
#endif
  constexpr int NB = (NS > 1 && G == 1) ? TFHE_KEY_BUFFERS_NS : TFHE_KEY_BUFFERS;
  elem kbuf[NB][CH];
  // OWN (TFHE_OWN_FIRST): a wave starts every level's multiply-accumulate with the digit spectrum it has just computed
  // itself -- straight from its registers, BEFORE the team barrier that makes the other groups' spectra visible -- and takes
  // the source polynomials in the order me, me + 1, ... (mod k + 1): one spectrum fewer to read back from LDS per level, and
  // a (k+1)-th of the products under the barrier's wait.  (The order of the exact integer sum changes, its value does not.)
  // The complex transform with one sample per team and one wave per polynomial: cfg2 31.9 -> 30.3 ms (reading the own
  // spectrum back from LDS, i.e. the barrier overlap alone: 30.8), cfg1 10.6 -> 10.5.  Not elsewhere: with two samples per
  // team the kept spectra spill (cfg3 52.7 -> 55.6 ms; through LDS: level; cfg5 51 -> 55-66 ms), the prime fields'
  // 16-element arrays spill or gain nothing (+-1 %) -- profiles/r03_kernel_ab.txt.
#ifndef TFHE_OWN_FIRST
#define TFHE_OWN_FIRST 1
#endif
  constexpr bool OWN = TFHE_OWN_FIRST && F::kLogShrink == 1 && NS == 1 && G == 1 && !SPLIT;
  constexpr bool OWN_REGS = OWN;
  // LATE (TFHE_LATE_BARRIER): the barrier that ends a level -- everyone is done reading the spectra before a buffer is
  // written again -- moves from behind the multiply-accumulate to just in front of the NEXT transform's first store into
  // the buffer (its first transpose; the inverse transform's after the last level): decomposition and the first register
  // pass touch no buffer and run under the barrier's wait.  Same shapes as OWN: cfg2 30.15 -> 28.9 ms, cfg1 10.5 -> 10.45;
  // with two samples per team level (cfg3 51.4 -> 51.3 ms) or worse (cfg5 51.0 -> 54.2 ms, 9 more spilled registers).
#ifndef TFHE_LATE_BARRIER
#define TFHE_LATE_BARRIER 1
#endif
  constexpr bool LATE = TFHE_LATE_BARRIER && TFHE_TOP_PREFETCH && OWN;
  auto source_of = [&](int sp) -> int {  // the source polynomial behind chunk position sp
    if (!OWN) return sp;
    const int at = me + sp;
    return at > K ? at - (K + 1) : at;
  };
  // piece of the key a chunk index names: tile (source polynomial, accumulator) and offset inside it
  auto load_chunk = [&](u32 level, auto ci_c, int buf) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr int PIECES = E / CH;
    constexpr int q = ci % ACCS, r0 = ((ci / ACCS) % PIECES) * CH, src_poly = ci / (ACCS * PIECES);
    const elem* tile = tile_ptr(level, source_of(src_poly), q);
#pragma unroll
    for (int r = 0; r < CH; ++r)
      kbuf[buf][r] = tile[TFHE_PROBE_HOT_KEY ? (lane & 63) + r : spectrum_slot<LT, G, (int)sizeof(elem), key_layout_e<F, LOGN, K>()>(lane, r0 + r)];
  };
  // the contexts of the samples' buffers (NS == 1: the level's parity buffer, see above)
  auto buffers_of_level = [&](u32 t, Ctx (&cl)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) cl[s] = c.with_exchange_buffer(NS > 1 ? s : (two ? (int)(t & 1u) : 0));
  };
  // (not unrolled: with the level count fixed at 3 and the loop fully unrolled the kernel issues 2.8 %
  // fewer instructions from 3x the code and runs no faster, profiles/r02_kernel_ab.txt)
#pragma unroll 1
  for (u32 t = 0; t < P.levels; ++t) {  // limbs LSB -> MSB; level index counts from the MSB
    const u32 level = P.levels - 1 - t;
    const u32 shift = P.first_shift + P.log_base * t;
    Ctx cl[NS];
    buffers_of_level(t, cl);
    static_for<0, (NB - 1 < CHUNKS ? NB - 1 : CHUNKS)>([&](auto pre_c) { load_chunk(level, pre_c, decltype(pre_c)::value); });
    c.compiler_fence();
    // The samples' digit rows are transformed one after the other (TFHE_NS_LOCKSTEP_FORWARD = 0): in lockstep
    // (ntt_forward_multi over all of them, shared transposes) two working arrays are live next to every sample's
    // accumulators and the kernels of the 168-register shapes spill (89 / 33 registers at N = 2048 / N = 512, k = 2);
    // the inverse transforms do run in lockstep -- their arrays ARE the accumulators.
#ifndef TFHE_NS_LOCKSTEP_FORWARD
#define TFHE_NS_LOCKSTEP_FORWARD 0
#endif
    constexpr int FS = (NS > 1 && !TFHE_NS_LOCKSTEP_FORWARD) ? 1 : NS;  // samples per forward transform call
    const u32 carry_width = (t == 0) ? 0u : 1u;  // wave-uniform: the lowest kept limb has no carry-in
    elem own[OWN_REGS ? NS : 1][OWN_REGS ? E : 1];  // OWN_REGS: my digit spectrum of this level, as published
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += FS) {
      elem work[FS][E];
      // F::kMaxLogBase: the largest gadget base the field's exactness bound admits at all; only the
      // Goldilocks fields reach bases above 2^23 (one-level decompositions) and branch at run time
      if (F::kMaxLogBase <= 23 || P.log_base <= 23) {
#pragma unroll
        for (int s = 0; s < FS; ++s)
#pragma unroll
          for (int r = 0; r < E; ++r) {
            u32 dg[CO];
#pragma unroll
            for (int q = 0; q < CO; ++q) dg[q] = decompose_limb_fast<true>(v[s0 + s][r + q * E], shift, P.log_base, carry_width);
            work[s][r] = F::from_digits(dg);
          }
      } else {
#pragma unroll
        for (int s = 0; s < FS; ++s)
#pragma unroll
          for (int r = 0; r < E; ++r) {
            u32 dg[CO];
#pragma unroll
            for (int q = 0; q < CO; ++q) dg[q] = decompose_limb_fast<false>(v[s0 + s][r + q * E], shift, P.log_base, carry_width);
            work[s][r] = F::from_digits(dg);
          }
      }
      // digits are tiny (|d| <= B <= 2^F::kSmallBits, enforced when the context picks the field):
      // the first butterfly stage uses F::mul_small.  A team barrier precedes every level (the
      // caller's for level 0, the previous level's below) and nobody reads these buffers after it.
      Ctx cf[FS];
#pragma unroll
      for (int s = 0; s < FS; ++s) cf[s] = cl[s0 + s];
      if constexpr (LATE) {
        // (level 0 follows the previous product's inverse transforms, which use a wave's own buffer only: no barrier owed)
        ntt_forward_multi<F, LT, G, true, true>(cf, work, ftop, [&]() { if (t > 0 && !two && s0 == 0) c.team_sync(); });
      } else if constexpr (TFHE_TOP_PREFETCH) {
        ntt_forward_multi<F, LT, G, true, true>(cf, work, ftop);
      } else {
        const TopFromTable<elem> top{c.twiddles_uniform(), 1 << LT};
        ntt_forward_multi<F, LT, G, true, true>(cf, work, top);
      }
#pragma unroll
      for (int s = 0; s < FS; ++s) {
        if (F::kReduceSpectrum) {  // little lazy headroom: MAC terms must start from |d| <= p/2
#pragma unroll
          for (int r = 0; r < E; ++r) work[s][r] = F::reduce(work[s][r]);
        }
        // publish: element r of thread tid at exchange_slot(tid, r) -- inside my wave's own part of
        // the buffer (wave_ntt.h), conflict-free 8-byte accesses
        elem* mine = cf[s].scratch();
#pragma unroll
        for (int r = 0; r < E; ++r) mine[exchange_slot<LT, G>(lane, r)] = work[s][r];
        if constexpr (OWN_REGS) {
#pragma unroll
          for (int r = 0; r < E; ++r) own[s0 + s][r] = work[s][r];
        }
      }
    }
    if constexpr (!OWN) c.team_sync();
    // chunk order: source polynomial sp, then the CH-register piece of its spectrum, then the
    // accumulator (key, key part) -- so that a piece of a digit spectrum is read from LDS once and
    // used for every key and part, and a key chunk is fetched once and used for every sample
    elem d[NS][CH];
    constexpr bool MAC_LOW = mac_lowers_priority<elem, E>() && KEYS == 1;  // wave_ntt.h: issue priority by phase (not the unrolled rotation: measured worse)
    if constexpr (MAC_LOW) wave_priority<0>();
    static_for<0, CHUNKS>([&](auto ci_c) {
      constexpr int ci = decltype(ci_c)::value;
      constexpr int PIECES = E / CH;
      constexpr int q = ci % ACCS, r0 = ((ci / ACCS) % PIECES) * CH, sp = ci / (ACCS * PIECES);
      constexpr int cur = ci % NB;
      if constexpr (OWN && ci == ACCS * PIECES) c.team_sync();  // my own spectrum is done with: now the others'
      if constexpr (ci + NB - 1 < CHUNKS) load_chunk(level, IntC<ci + NB - 1>{}, (ci + NB - 1) % NB);
      // (TFHE_PROBE_NO_EXCHANGE_READS: dev_switches.h -- a WRONG-BITS timing probe, TFHE_DEV_BUILD only)
      if constexpr (OWN_REGS && sp == 0) {
        if constexpr (q == 0) {
#pragma unroll
          for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < CH; ++r) d[s][r] = own[s][r0 + r];
        }
      } else if constexpr (q == 0 && !(TFHE_PROBE_NO_EXCHANGE_READS)) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const elem* spec = cl[s].scratch_of(source_of(sp));
#pragma unroll
          for (int r = 0; r < CH; ++r) d[s][r] = spec[exchange_slot<LT, G>(lane, r0 + r)];
        }
      } else if constexpr (q == 0 && ci == 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
          for (int r = 0; r < CH; ++r) d[s][r] = kbuf[0][r];
      }
      c.compiler_fence();  // keep the next chunk's loads above this chunk's arithmetic
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < CH; ++r) {
          if (SPLIT)
            F::mac(d[s][r], kbuf[cur][r], accum[s][q][r0 + r], accum_lo[SPLIT ? q : 0][SPLIT ? r0 + r : 0]);
          else
            accum[s][q][r0 + r] = F::mul_add(d[s][r], kbuf[cur][r], accum[s][q][r0 + r]);
        }
    });
    if constexpr (MAC_LOW) wave_priority<2>();
    if (!two && !LATE) c.team_sync();  // everyone is done reading before the next transform reuses the buffer
  }

  Ctx ci[NS];
  buffers_of_level(P.levels, ci);
  TopConsts<F, LT, G, false> itop;  // arrives during the first two passes of the first inverse transform
  if constexpr (TFHE_TOP_PREFETCH) itop.issue(c.twiddles_uniform());
  // (TFHE_INVERSE_PAIR: one sample's two key parts go through the inverse transform half a step apart, each one's
  // transposes covered by the other's register passes -- wave_ntt.h::ntt_inverse_pair)
#ifndef TFHE_INVERSE_PAIR
#define TFHE_INVERSE_PAIR 1
#endif
  // Measured where it is on (profiles/r03_kernel_ab.txt): the complex transform at 8 elements per lane (N = 1024), +0.9 %;
  // at 4 elements per lane (N = 512) it is level, the prime fields' 16-element arrays spill with it.
  constexpr bool PAIR = TFHE_INVERSE_PAIR && NS == 1 && PARTS == 2 && G == 1 && !SPLIT && NttShape<LT, G>::kPasses == 3 &&
                        TFHE_TOP_PREFETCH && F::kLogShrink == 1 && E == 8;
  static_for<0, KEYS>([&](auto key_c) {
    constexpr int m = decltype(key_c)::value;
    if constexpr (PAIR) {
#pragma unroll
      for (int r = 0; r < E; ++r) {
        accum[0][m * PARTS][r] = F::before_inverse(accum[0][m * PARTS][r]);
        accum[0][m * PARTS + 1][r] = F::before_inverse(accum[0][m * PARTS + 1][r]);
      }
      if constexpr (m == 0) itop.ready();
      if constexpr (LATE && m == 0)
        ntt_inverse_pair<F, LT, G>(ci[0], accum[0][m * PARTS], accum[0][m * PARTS + 1], itop, [&]() { if (!two) c.team_sync(); });
      else
        ntt_inverse_pair<F, LT, G>(ci[0], accum[0][m * PARTS], accum[0][m * PARTS + 1], itop);
    }
    static_for<0, PAIR ? 0 : PARTS>([&](auto part_c) {
      constexpr int q = m * PARTS + decltype(part_c)::value;
      // part q of every sample goes through the inverse transform together
      elem x[NS][E];
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < E; ++r)
          x[s][r] = SPLIT ? F::mac_finish(accum[s][q][r], accum_lo[SPLIT ? q : 0][SPLIT ? r : 0])
                          : F::before_inverse(accum[s][q][r]);
      if constexpr (LATE) {
        if constexpr (q == 0) {
          itop.ready();
          ntt_inverse_multi<F, LT, G>(ci, x, itop, [&]() { if (!two) c.team_sync(); });  // the last level's barrier
        } else {
          ntt_inverse_multi<F, LT, G>(ci, x, itop);
        }
      } else if constexpr (TFHE_TOP_PREFETCH) {
        if constexpr (q == 0) itop.ready();  // nothing else is in flight here: the wait is for the block alone
        ntt_inverse_multi<F, LT, G>(ci, x, itop);
      } else {
        const TopFromTable<elem> top{c.twiddles_uniform(), 1 << LT};
        ntt_inverse_multi<F, LT, G>(ci, x, top);
      }
#pragma unroll
      for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int r = 0; r < E; ++r) accum[s][q][r] = x[s][r];
    });
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int r = 0; r < E; ++r) {
        elem parts[PARTS];
#pragma unroll
        for (int q = 0; q < PARTS; ++q) parts[q] = accum[s][m * PARTS + q][r];
        u32 vals[CO];
        F::finish(parts, vals);
#pragma unroll
        for (int q = 0; q < CO; ++q) out(m, s, (r + q * E) * T + lane, vals[q]);
      }
    end_of_key(m);
  });
  // two buffers: the MAC reads of the last level must be over before a following product (or any
  // other user of the buffers) writes buffer 0; this barrier also orders the out() stores of the
  // whole team.  One buffer: the last level already ended with a barrier.
  if (two) c.team_sync();
}

// one operand, KEYS keys (the unrolled blind rotation): src(j), out(m, j, value)
template <class F, int LOGN, int K, int G, int KEYS, class Ctx, class Src, class Out, class EndKey>
TFHE_HD void external_product_team_keys(const Ctx& c, const PbsParams& P, const typename F::elem* ggsw,
                                        size_t ggsw_words, Src src, Out out, EndKey end_of_key) {
  external_product_team_multi<F, LOGN, K, G, KEYS, 1>(
      c, P, ggsw, ggsw_words, [&](int, int j) -> u32 { return src(j); },
      [&](int m, int, int j, u32 value) { out(m, j, value); }, end_of_key);
}

// one GGSW (ggsw.rs:132-161)
template <class F, int LOGN, int K, int G, class Ctx, class Src, class Out>
TFHE_HD void external_product_team(const Ctx& c, const PbsParams& P, const typename F::elem* ggsw,
                                   Src src, Out out) {
  external_product_team_keys<F, LOGN, K, G, 1>(
      c, P, ggsw, 0, src, [&](int, int j, u32 value) { out(j, value); }, [](int) {});
}

// ---------------------------------------------------------------------------------------------
// Blind rotation of NS LWE samples (bootstrapping.rs:67-105 for each) by a team of K+1 groups of G waves.
// Group c keeps polynomial c of sample s's accumulator in its LDS array c.acc(s) (N u32, natural order)
// for all n iterations; on return it holds polynomial c of the final GLWE accumulator of sample s.
// NS = 1 is the plain one-sample-per-team form; with NS = 2 every CMUX iteration multiplies both samples'
// rotated differences with GGSW_i in one pass (external_product_team_multi).
// ---------------------------------------------------------------------------------------------
// A blind rotation may be cut into SEGMENTS of CMUX iterations [i_begin, i_end), one launch each: a segment that does not
// start at 0 resumes from the GLWE accumulators the previous one left in global memory (resume[s]: [K+1][N] of sample
// s), and the caller stores c.acc(s) again if i_end < n.  (Teams of one launch start a segment together, so a long
// rotation's drift -- and with it the span of the key the L2s have to hold -- is bounded by the segment; see
// kernels.hip::blind_rotate_plan.)
template <class F, int LOGN, int K, int G, int NS, class Ctx>
TFHE_HD void blind_rotate_team_multi(const Ctx& c, const PbsParams& P, const u32* const* lwe /* NS x (n+1) */,
                                     const u32* const* tv /* NS x N, un-encoded */,
                                     const typename F::elem* bsk /* prepared */, u32 i_begin, u32 i_end,
                                     const u32* const* resume /* NS x [K+1][N], read if i_begin > 0 */) {
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  constexpr int N = 1 << LOGN;
  const int lane = c.tid();
  const int me = c.group();

  if (i_begin > 0) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      u32* acc = c.acc(s);
      const u32* from = resume[s] + (size_t)me * N;
#pragma unroll
      for (int r = 0; r < E; ++r) acc[r * T + lane] = from[r * T + lane];
    }
  } else {
  // acc = X^{-b~} * (0, ..., 0, tv << tv_shift): only the body polynomial (wave K) is non-zero
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    u32* acc = c.acc(s);
    const u32 b_tilde = switch_modulus_2n(lwe[s][P.n], LOGN);
    const u32 m = (2u * N - b_tilde) & (2u * N - 1u);
    const int deg = (int)(m & (N - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int j = r * T + lane;
      u32 val = 0;
      if (me == K) {
        const u32 t = tv[s][(j - deg) & (N - 1)] << P.tv_shift;
        val = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
      }
      acc[j] = val;
    }
  }
  }
  c.poly_sync();

  const size_t ggsw_words = (size_t)(K + 1) * P.levels * (K + 1) * F::kParts * (N >> F::kLogShrink);  // elements
#pragma unroll 1
  for (u32 i = i_begin; i < i_end; ++i) {
    u32 a_tilde[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) a_tilde[s] = c.uniform(switch_modulus_2n(lwe[s][i], LOGN));
    // cmux(ggsw_i, acc, X^{a~} * acc) = external_product(ggsw_i, X^{a~} acc - acc) + acc.
    // (a~ = 0 gives all-zero digits and leaves acc unchanged; it is not skipped because every wave
    // of the team has to take part in the barriers.)
    auto src = [&](int s, int j) -> u32 {
      const u32* acc = c.acc(s);
      return monomial_coeff<LOGN>(acc, j, a_tilde[s]) - acc[j];
    };
    // all rotated reads of acc happen before the first inverse transform: in-place update is safe
    auto out = [&](int, int s, int j, u32 value) { c.lds_add(c.acc(s) + j, value); };
    external_product_team_multi<F, LOGN, K, G, 1, NS>(c, P, bsk + (size_t)i * ggsw_words, 0, src, out, [](int) {});
    // G > 1: the other waves of my group read what I just wrote (with two exchange buffers and one sample the
    // product already ended with a team barrier)
    if (NS > 1 || c.exchange_buffers() != 2) c.poly_sync();
  }
}

template <class F, int LOGN, int K, int G, int NS, class Ctx>
TFHE_HD void blind_rotate_team_multi(const Ctx& c, const PbsParams& P, const u32* const* lwe, const u32* const* tv,
                                     const typename F::elem* bsk) {
  blind_rotate_team_multi<F, LOGN, K, G, NS>(c, P, lwe, tv, bsk, 0u, P.n, nullptr);
}

template <class F, int LOGN, int K, int G, class Ctx>
TFHE_HD void blind_rotate_team(const Ctx& c, const PbsParams& P, const u32* lwe /* n+1 */,
                               const u32* tv /* N, un-encoded */,
                               const typename F::elem* bsk /* prepared */) {
  const u32* const lwes[1] = {lwe};
  const u32* const tvs[1] = {tv};
  blind_rotate_team_multi<F, LOGN, K, G, 1>(c, P, lwes, tvs, bsk);
}

// ---------------------------------------------------------------------------------------------
// The WIDE team: the latency shape for batches that leave most of the chip idle (the reference's own call
// shape is ONE ciphertext per bootstrap(), bootstrapping.rs:58-65, one pair per gate, boolean.rs:9-37).
//
// A team of K+1 waves serialises, per CMUX, `levels` forward transforms, (K+1) levels kParts multiply-accumulate
// tiles and kParts inverse transforms in every wave.  The wide team spends kParts = 2 waves per polynomial and
// splits that chain by LEVEL and by KEY PART instead of by coefficient (no extra register pass, no cross-wave
// transpose): wave (c, q), c = polynomial / output column, q = half,
//   - reads and rounds X^a acc_c - acc_c like its twin (16 words per lane: cheap to do twice) and runs the digit
//     chain from the lowest limb, but forward-transforms only ITS levels -- the lower ceil(l/2) limbs for q = 0,
//     the upper ones for q = 1 -- each in the LDS buffer of its digit row (row = c l + level), where the
//     spectrum then stays published;
//   - barrier A: all (K+1) l digit spectra are visible (and every wave has read the accumulator polynomials);
//   - accumulates key part q of column c over ALL rows (half the tiles of a one-wave-per-polynomial team);
//   - inverse-transforms its one accumulator (in its own transpose buffer) and adds lift(part q) << 16 q to the
//     accumulator polynomial with ds_add_u32 -- wrapping addition commutes, so the two halves of a column need
//     no order: acc_c += t_lo + 2^16 t_hi mod 2^32 exactly as FftField::finish forms it;
//   - barrier B (the caller's, after the product): the accumulators are complete and the row buffers free.
// Per wave and CMUX at N = 1024, k = 1, l = 3: 2 (or 1) forward transforms, 6 tiles, 1 inverse transform instead
// of 3, 12 and 2.  The same exact integer sums in another order: same bits (field_fft.h: the rounding bound does
// not depend on the order of the rows).  Needs (K+1) l row buffers and 2 (K+1) transpose buffers of N x 8 bytes in
// LDS: offered up to N = 1024.  TWO team barriers per CMUX, whatever the level count.
//
// Ctx additionally provides: half() (q), row_buffer(r): the buffer of digit row r; scratch() is the wave's own buffer.
// ---------------------------------------------------------------------------------------------
// Key delivery.  LEVELS = 0: the level count is a run-time value; the tiles of my (column, part) come in chunks of CH
// elements, double-buffered in registers, one chunk ahead of the arithmetic (any parameter set).  LEVELS > 0 (the level
// count the kernel was instantiated for; P.levels must equal it): the KEY RING.  One wave alone on its SIMD has nobody to
// hide a load behind, and a chunk's arithmetic (16 fused multiply-adds) is a seventh of an L2 round trip: with chunks one
// ahead the multiply-accumulate of an idle chip waits for the key ~24 times per CMUX (profiles/r04_a_batch_sweep*: 6.8 us
// per CMUX at N = 1024, k = 1 against 2.7 us of instructions).  The register file is the one store big enough to take a
// GGSW ahead of time (4 waves x 192 registers x 256 B = the 197 KB of cfg2's GGSW): the ring holds RING whole rows
// (E elements each) of my tiles, RING | (K+1) LEVELS; row m of a product sits in slot m % RING; as soon as a row has
// been multiplied its slot is refilled with row m + RING -- of this product, or of the NEXT CMUX's GGSW (`ggsw_next`),
// so the loads of a CMUX travel under the previous CMUX's arithmetic and the forward transforms in between.
// The ring lives in the caller (it survives from product to product): WideKeyRing.
template <class F, int LOGN, int K, int LEVELS, int RING>
struct WideKeyRing {
  static constexpr int E = NttShape<LOGN - 1, 1>::kE;
  static constexpr int R = (K + 1) * LEVELS;
  static constexpr int kRing = RING;
  static_assert(LEVELS > 0 && RING > 0 && R % RING == 0, "the ring holds a divisor of the product's rows");
  typename F::elem slot[RING][E];
  // row m (multiply-accumulate order: level = m / (K+1), source polynomial = m % (K+1)) of the GGSW at `ggsw`, for
  // output column `me` and key part q
  template <int M>
  TFHE_HD void load(const typename F::elem* ggsw, int me, int q, int lane) {
    constexpr int level = M / (K + 1), sp = M % (K + 1);
    const typename F::elem* tile = ggsw + ((((size_t)sp * LEVELS + level) * (K + 1) + me) * 2 + q) * ((size_t)1 << (LOGN - 1));
#pragma unroll
    for (int r = 0; r < E; ++r)
      slot[M % RING][r] = tile[spectrum_slot<LOGN - 1, 1, (int)sizeof(typename F::elem), key_layout_e<F, LOGN, K>()>(lane, r)];
  }
  // before the first product: rows 0 .. RING-1
  TFHE_HD void prime(const typename F::elem* ggsw, int me, int q, int lane) {
    static_for<0, RING>([&](auto m_c) { load<decltype(m_c)::value>(ggsw, me, q, lane); });
  }
};
struct NoKeyRing {
  static constexpr int kRing = 1;
};
// rows the ring of a (LOGN, K, LEVELS) kernel holds: the largest divisor of the product's (K+1) LEVELS rows that the register
// budget takes (kernels.hip explains the budget: 224 registers for a k = 1 team's waves, 112 for a k = 2 team's)
constexpr int largest_divisor_upto(int n, int cap) {
  int best = 1;
  for (int d = 1; d <= n && d <= cap; ++d)
    if (n % d == 0) best = d;
  return best;
}
template <int LOGN, int K, int LEVELS>
constexpr int wide_ring_rows() {
  constexpr int E = NttShape<LOGN - 1, 1>::kE;
  return LEVELS == 0 ? 0 : largest_divisor_upto((K + 1) * LEVELS, (K == 1 ? 224 : 112) / (4 * E));
}


template <class F, int LOGN, int K, int LEVELS = 0, class Ring = NoKeyRing, class Ctx, class Src, class Out>
TFHE_HD void external_product_team_wide(const Ctx& c, const PbsParams& P, const typename F::elem* ggsw,
                                        const typename F::elem* ggsw_next, Ring& ring, Src src, Out out) {
  typedef typename F::elem elem;
  static_assert(F::kParts == 2 && F::kLogShrink == 1 && F::kCoeffs == 2, "the complex transform: two key parts, two coefficients per element");
  constexpr int LT = LOGN - 1;
  constexpr int E = NttShape<LT, 1>::kE;
  constexpr int EC = 2 * E;
  constexpr int T = 64;
  constexpr int N = 1 << LT;  // transform elements per polynomial
  constexpr bool RINGED = LEVELS > 0;
  const int lane = c.tid();
  const int me = c.group();
  const int q = c.half();
  const u32 levels = RINGED ? (u32)LEVELS : P.levels;
  // levels counted from the least significant limb (t), as the digit chain runs; level index = levels - 1 - t
  const u32 split = (levels + 1u) / 2u;
  const u32 t_begin = q == 0 ? 0u : split, t_end = q == 0 ? split : levels;

  TopConsts<F, LT, 1, true> ftop;
  ftop.issue(c.twiddles_uniform());
  u32 v[EC];
  const RoundConsts rc = round_consts(P.ignored_bits);
#pragma unroll
  for (int r = 0; r < EC; ++r) v[r] = round_value_fast(src(r * T + lane), rc);
  ftop.ready();

  // (no ring) key chunks of my (column, part): row by row, CH elements at a time, double-buffered in registers
  // (an even number of chunks per level keeps the staging buffer of a chunk a compile-time constant across levels)
  constexpr int CH0 = E < 4 ? E : 4;
  constexpr int CH = (((K + 1) * (E / CH0)) % 2 == 0) ? CH0 : CH0 / 2;
  constexpr int PIECES = E / CH;
  constexpr int CHK = (K + 1) * PIECES;  // chunks per level: source polynomial sp, then the piece of its spectrum
  elem kbuf[RINGED ? 1 : 2][RINGED ? 1 : CH];
  auto load_chunk = [&](u32 level, auto ci_c, int buf) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr int sp = ci / PIECES, r0 = (ci % PIECES) * CH;
    const elem* tile = ggsw + ((((size_t)sp * levels + level) * (K + 1) + me) * 2 + q) * N;
#pragma unroll
    for (int r = 0; r < CH; ++r)
      kbuf[RINGED ? 0 : buf][RINGED ? 0 : r] = tile[spectrum_slot<LT, 1, (int)sizeof(elem), key_layout_e<F, LOGN, K>()>(lane, r0 + r)];
  };
  if constexpr (!RINGED) {
    load_chunk(0u, IntC<0>{}, 0);  // in flight under the forward transforms
    c.compiler_fence();
  }

#pragma unroll 1
  for (u32 t = 0; t < t_end; ++t) {
    const u32 shift = P.first_shift + P.log_base * t;
    const u32 carry_width = (t == 0) ? 0u : 1u;
    elem work[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
      u32 dg[2];
      dg[0] = decompose_limb_fast<true>(v[r], shift, P.log_base, carry_width);
      dg[1] = decompose_limb_fast<true>(v[r + E], shift, P.log_base, carry_width);
      work[r] = F::from_digits(dg);
    }
    if (t < t_begin) continue;  // (wave-uniform) my twin's limb: only the carries were needed
    const u32 row = (u32)me * levels + (levels - 1u - t);
    // transposes in my own buffer (fixed addresses: the swizzled offsets stay loop invariants), the spectrum then goes
    // to its row's buffer, which nobody has read since barrier B of the previous product
    ntt_forward<F, LT, 1, true, true>(c, work, ftop);
    elem* published = c.row_buffer((int)row);
#pragma unroll
    for (int r = 0; r < E; ++r) published[exchange_slot<LT, 1>(lane, r)] = work[r];
  }
  TopConsts<F, LT, 1, false> itop;  // arrives under the multiply-accumulate
  itop.issue(c.twiddles_uniform());
  c.team_sync();  // A: every digit spectrum is published

  elem accum[E];
#pragma unroll
  for (int r = 0; r < E; ++r) accum[r] = F::zero();
  if constexpr (RINGED) {
    constexpr int R = (K + 1) * LEVELS;
    constexpr int RING_ROWS = Ring::kRing;
    static_for<0, R>([&](auto m_c) {
      constexpr int m = decltype(m_c)::value;
      constexpr int level = m / (K + 1), sp = m % (K + 1);
      const elem* spec = c.row_buffer(sp * LEVELS + level);
      elem d[E];
#pragma unroll
      for (int r = 0; r < E; ++r) d[r] = spec[exchange_slot<LT, 1>(lane, r)];
#pragma unroll
      for (int r = 0; r < E; ++r) accum[r] = F::mul_add(d[r], ring.slot[m % RING_ROWS][r], accum[r]);
      c.compiler_fence();  // the refill below stays behind this row's arithmetic: its registers are the row's
      // refill the slot: row m + RING of this product, or row m + RING - R of the next CMUX's GGSW
      if constexpr (m + RING_ROWS < R) ring.template load<m + RING_ROWS>(ggsw, me, q, lane);
      else ring.template load<m + RING_ROWS - R>(ggsw_next, me, q, lane);
    });
  } else {
#pragma unroll 1
    for (u32 level = 0; level < levels; ++level) {
      static_for<0, CHK>([&](auto ci_c) {
        constexpr int ci = decltype(ci_c)::value;
        constexpr int sp = ci / PIECES, r0 = (ci % PIECES) * CH;
        constexpr int cur = ci % 2;
        static_assert(CHK % 2 == 0, "an even number of chunks per level keeps the buffer parity static");
        if constexpr (ci + 1 < CHK) {
          load_chunk(level, IntC<ci + 1>{}, (ci + 1) % 2);
        } else {
          if (level + 1u < levels) load_chunk(level + 1u, IntC<0>{}, 0);
        }
        const elem* spec = c.row_buffer((int)((u32)sp * levels + level));
        elem d[CH];
#pragma unroll
        for (int r = 0; r < CH; ++r) d[r] = spec[exchange_slot<LT, 1>(lane, r0 + r)];
        c.compiler_fence();  // keep the next chunk's loads above this chunk's arithmetic
#pragma unroll
        for (int r = 0; r < CH; ++r) accum[r0 + r] = F::mul_add(d[r], kbuf[RINGED ? 0 : cur][RINGED ? 0 : r], accum[r0 + r]);
      });
    }
  }
  // (no barrier here: the inverse transform works in my own buffer and the accumulator polynomial is only read before
  // barrier A; the row buffers are written again after barrier C)

#pragma unroll
  for (int r = 0; r < E; ++r) accum[r] = F::before_inverse(accum[r]);
  itop.ready();
  ntt_inverse<F, LT, 1>(c, accum, itop);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    out(q, r * T + lane, F::to_u32(accum[r].re));
    out(q, (r + E) * T + lane, F::to_u32(accum[r].im));
  }
}

// Blind rotation of ONE sample by a wide team (bootstrapping.rs:67-105); segments as blind_rotate_team_multi.
// LEVELS / RING: the key ring (above); LEVELS = 0: run-time level count, chunked key loads.
template <class F, int LOGN, int K, int LEVELS = 0, int RING = 0, class Ctx>
TFHE_HD void blind_rotate_team_wide(const Ctx& c, const PbsParams& P, const u32* lwe /* n+1 */, const u32* tv /* N, un-encoded */,
                                    const typename F::elem* bsk /* prepared */, u32 i_begin, u32 i_end,
                                    const u32* resume /* [K+1][N], read if i_begin > 0 */) {
  constexpr int N = 1 << LOGN;
  constexpr int EC = N / 64;  // accumulator words per lane
  const int lane = c.tid();
  const int me = c.group();
  const int q = c.half();
  u32* acc = c.acc();
  const u32 levels = LEVELS > 0 ? (u32)LEVELS : P.levels;
  const size_t ggsw_words = (size_t)(K + 1) * levels * (K + 1) * 2 * (N >> 1);  // elements
  typename std::conditional<(LEVELS > 0), WideKeyRing<F, LOGN, K, (LEVELS > 0 ? LEVELS : 1), (RING > 0 ? RING : 1)>, NoKeyRing>::type ring;
  if constexpr (LEVELS > 0) {
    if (i_begin < i_end) ring.prime(bsk + (size_t)i_begin * ggsw_words, me, q, lane);  // under the accumulator's set-up
  }
  // each half fills (and, in the kernel's epilogue, stores) half of the polynomial's words: registers [q EC/2, (q+1) EC/2)
  if (i_begin > 0) {
    const u32* from = resume + (size_t)me * N;
#pragma unroll
    for (int r = 0; r < EC / 2; ++r) acc[(r + q * (EC / 2)) * 64 + lane] = from[(r + q * (EC / 2)) * 64 + lane];
  } else {
    // acc = X^{-b~} * (0, ..., 0, tv << tv_shift)
    const u32 b_tilde = switch_modulus_2n(lwe[P.n], LOGN);
    const u32 m = (2u * N - b_tilde) & (2u * N - 1u);
    const int deg = (int)(m & (N - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < EC / 2; ++r) {
      const int j = (r + q * (EC / 2)) * 64 + lane;
      u32 val = 0;
      if (me == K) {
        const u32 t = tv[(j - deg) & (N - 1)] << P.tv_shift;
        val = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
      }
      acc[j] = val;
    }
  }
  c.team_sync();
#pragma unroll 1
  for (u32 i = i_begin; i < i_end; ++i) {
    const u32 a_tilde = c.uniform(switch_modulus_2n(lwe[i], LOGN));
    auto src = [&](int j) -> u32 { return monomial_coeff<LOGN>(acc, j, a_tilde) - acc[j]; };
    // every rotated read of acc happens before barrier A, every update after it: in place is safe
    auto out = [&](int half, int j, u32 value) { c.lds_add(acc + j, value << (16 * half)); };
    const typename F::elem* ggsw = bsk + (size_t)i * ggsw_words;
    // the ring's refills of the last CMUX have no successor: they re-read this GGSW's first rows (valid memory, unused)
    const typename F::elem* ggsw_next = i + 1 < i_end ? ggsw + ggsw_words : ggsw;
    external_product_team_wide<F, LOGN, K, LEVELS>(c, P, ggsw, ggsw_next, ring, src, out);
    c.team_sync();  // B: both halves of every column have added their part; all reads of the spectra are over
  }
}

// ---------------------------------------------------------------------------------------------
// The PAIR kernel: ONE wavefront per LWE sample at k = 1, N = 512.
//
// A team of two waves at N = 512 holds 4 transform elements per lane: four register passes and three transposes per
// transform, two team barriers per level, and a CU holds 6 samples (3 waves per SIMD).  Here both polynomials of a sample sit
// side by side in ONE wave -- polynomial c in lanes 32 c .. 32 c + 31, 8 elements per lane (NttShape<LT, 0>) -- and the same
// instruction stream transforms both: three register passes, TWO transposes per transform that serve both polynomials, the
// two digit spectra change halves through the wave's own LDS buffers under a wave-level fence, and there is NO workgroup
// barrier anywhere (one wave's LDS operations execute in order).  A third fewer LDS cycles per sample, and 8 samples per CU
// (2 waves per SIMD at the N = 1024 kernel's register footprint) instead of 6.
//   half c: decomposes polynomial c, forward-transforms its `levels` digit rows, accumulates BOTH key parts of output
//   column c over both polynomials' spectra (its own from registers, the other half's read back from LDS), inverse-transforms
//   its two accumulators (one after the other's transposes: ntt_inverse_pair) and updates accumulator polynomial c.
// Ctx: tid() = lane within the half (0..31), group() = half = polynomial, scratch() / scratch_of(c) = a half's buffer.
// ---------------------------------------------------------------------------------------------
template <class F, int LOGN, class Ctx, class Src, class Out>
TFHE_HD void external_product_pair(const Ctx& c, const PbsParams& P, const typename F::elem* ggsw, Src src, Out out) {
  typedef typename F::elem elem;
  static_assert(F::kParts == 2 && F::kLogShrink == 1 && F::kCoeffs == 2, "the complex transform: two key parts, two coefficients per element");
  constexpr int K = 1;
  constexpr int LT = LOGN - 1;
  constexpr int G = 0;  // half a wave per polynomial
  constexpr int E = NttShape<LT, G>::kE;
  constexpr int EC = 2 * E;
  constexpr int T = NttShape<LT, G>::kThreads;  // 32
  constexpr int N = 1 << LT;
  const int tid = c.tid();
  const int me = c.group();

  elem accum[2][E];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int r = 0; r < E; ++r) accum[q][r] = F::zero();

  TopConsts<F, LT, G, true> ftop;
  ftop.issue(c.twiddles_uniform());
  u32 v[EC];
  const RoundConsts rc = round_consts(P.ignored_bits);
#pragma unroll
  for (int r = 0; r < EC; ++r) v[r] = round_value_fast(src(r * T + tid), rc);
  ftop.ready();

  // key chunks of a level for my column: source polynomial (mine first, then the other half's), piece of the spectrum, key
  // part; CH elements at a time, double-buffered in registers
  constexpr int CH = 4;
  constexpr int PIECES = E / CH;
  constexpr int CHUNKS = 2 * PIECES * 2;  // per level
  elem kbuf[2][CH];
  auto load_chunk = [&](u32 level, auto ci_c, int buf) {
    constexpr int ci = decltype(ci_c)::value;
    constexpr int q = ci % 2, r0 = ((ci / 2) % PIECES) * CH, other = ci / (2 * PIECES);
    const int sp = other ? 1 - me : me;
    const elem* tile = ggsw + ((((size_t)sp * P.levels + level) * (K + 1) + me) * 2 + q) * N;
#pragma unroll
    for (int r = 0; r < CH; ++r) kbuf[buf][r] = tile[spectrum_slot<LT, G, (int)sizeof(elem)>(tid, r0 + r)];
  };
#pragma unroll 1
  for (u32 t = 0; t < P.levels; ++t) {
    const u32 level = P.levels - 1 - t;
    const u32 shift = P.first_shift + P.log_base * t;
    const u32 carry_width = (t == 0) ? 0u : 1u;
    load_chunk(level, IntC<0>{}, 0);  // in flight under the forward transform
    c.compiler_fence();
    elem work[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
      u32 dg[2];
      dg[0] = decompose_limb_fast<true>(v[r], shift, P.log_base, carry_width);
      dg[1] = decompose_limb_fast<true>(v[r + E], shift, P.log_base, carry_width);
      work[r] = F::from_digits(dg);
    }
    ntt_forward<F, LT, G, true, true>(c, work, ftop);
    // hand my spectrum to the other half: through my buffer (free between two transforms)
    elem* mine = c.scratch();
#pragma unroll
    for (int r = 0; r < E; ++r) mine[exchange_slot<LT, G>(tid, r)] = work[r];
    c.wave_sync();
    elem d[CH];
    static_for<0, CHUNKS>([&](auto ci_c) {
      constexpr int ci = decltype(ci_c)::value;
      constexpr int q = ci % 2, r0 = ((ci / 2) % PIECES) * CH, other = ci / (2 * PIECES);
      if constexpr (ci + 1 < CHUNKS) load_chunk(level, IntC<ci + 1>{}, (ci + 1) % 2);
      if constexpr (q == 0) {
        if constexpr (other == 0) {
#pragma unroll
          for (int r = 0; r < CH; ++r) d[r] = work[r0 + r];
        } else {
          const elem* spec = c.scratch_of(1 - me);
#pragma unroll
          for (int r = 0; r < CH; ++r) d[r] = spec[exchange_slot<LT, G>(tid, r0 + r)];
        }
      }
      c.compiler_fence();  // keep the next chunk's loads above this chunk's arithmetic
#pragma unroll
      for (int r = 0; r < CH; ++r) accum[q][r0 + r] = F::mul_add(d[r], kbuf[ci % 2][r], accum[q][r0 + r]);
    });
    c.wave_sync();  // the other half has read my spectrum before the next transform's transposes overwrite it
  }

  TopConsts<F, LT, G, false> itop;
  itop.issue(c.twiddles_uniform());
#pragma unroll
  for (int r = 0; r < E; ++r) {
    accum[0][r] = F::before_inverse(accum[0][r]);
    accum[1][r] = F::before_inverse(accum[1][r]);
  }
  itop.ready();
  ntt_inverse_pair<F, LT, G>(c, accum[0], accum[1], itop);
#pragma unroll
  for (int r = 0; r < E; ++r) {
    elem parts[2] = {accum[0][r], accum[1][r]};
    u32 vals[2];
    F::finish(parts, vals);
    out(r * T + tid, vals[0]);
    out((r + E) * T + tid, vals[1]);
  }
}

// Blind rotation of ONE sample by one wave (bootstrapping.rs:67-105); segments as blind_rotate_team_multi.
template <class F, int LOGN, class Ctx>
TFHE_HD void blind_rotate_pair(const Ctx& c, const PbsParams& P, const u32* lwe /* n+1 */, const u32* tv /* N, un-encoded */,
                               const typename F::elem* bsk /* prepared, pair layout */, u32 i_begin, u32 i_end,
                               const u32* resume /* [2][N], read if i_begin > 0 */) {
  constexpr int N = 1 << LOGN;
  constexpr int T = 32;
  constexpr int EC = N / T;  // accumulator words per lane
  constexpr int K = 1;
  const int tid = c.tid();
  const int me = c.group();
  u32* acc = c.acc();
  if (i_begin > 0) {
    const u32* from = resume + (size_t)me * N;
#pragma unroll
    for (int r = 0; r < EC; ++r) acc[r * T + tid] = from[r * T + tid];
  } else {
    // acc = X^{-b~} * (0, tv << tv_shift)
    const u32 b_tilde = switch_modulus_2n(lwe[P.n], LOGN);
    const u32 m = (2u * N - b_tilde) & (2u * N - 1u);
    const int deg = (int)(m & (N - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < EC; ++r) {
      const int j = r * T + tid;
      u32 val = 0;
      if (me == K) {
        const u32 t = tv[(j - deg) & (N - 1)] << P.tv_shift;
        val = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
      }
      acc[j] = val;
    }
  }
  c.wave_sync();
  const size_t ggsw_words = (size_t)(K + 1) * P.levels * (K + 1) * 2 * (N >> 1);  // elements
#pragma unroll 1
  for (u32 i = i_begin; i < i_end; ++i) {
    const u32 a_tilde = c.uniform(switch_modulus_2n(lwe[i], LOGN));
    auto src = [&](int j) -> u32 { return monomial_coeff<LOGN>(acc, j, a_tilde) - acc[j]; };
    // each half reads and updates only its own polynomial, the rotated reads all happen before the first update
    auto out = [&](int j, u32 value) { c.lds_add(acc + j, value); };
    external_product_pair<F, LOGN>(c, P, bsk + (size_t)i * ggsw_words, src, out);
    c.wave_sync();
  }
}

// ---------------------------------------------------------------------------------------------
// Unrolled blind rotation (notes/BMMP Bootstrapping.md:13-25): two key bits per step.  With
//   X^{a s + a' s'} = s s' (X^{a+a'} - 1) + s (1 - s') (X^a - 1) + (1 - s) s' (X^{a'} - 1) + 1
// one step is   acc += sum_{m<3} (X^{e_m} - 1) * (BK_{3j+m} [x] acc)   with e = (a+a', a, a') and the
// three GGSW encryptions BK_{3j} = s_{2j} s_{2j+1}, BK_{3j+1} = s_{2j} (1 - s_{2j+1}),
// BK_{3j+2} = s_{2j+1} (1 - s_{2j}): n/2 decompositions and n/2 * levels forward transforms instead of
// n and n * levels, against 1.5x the multiply-accumulates and inverse transforms and a 1.5x key.
// The three products share the digits of acc (external_product_team_keys); each one is lifted on
// its own and the monomial factor is applied in the coefficient domain -- an index rotation of the
// staged product (my own transpose buffer: free once the inverse transforms are done and private
// until I publish again) -- so every value is an ordinary external product within the field's bound
// and the wrapping u32 sums are exact.  n must be even.  Not the reference's bootstrap(): the key
// material differs, so the output bits differ from it (same plaintext); the oracle twin is
// oracle.bootstrap_bmmp.
// ---------------------------------------------------------------------------------------------
template <class F, int LOGN, int K, int G, class Ctx>
TFHE_HD void blind_rotate_bmmp_team(const Ctx& c, const PbsParams& P, const u32* lwe /* n+1 */,
                                    const u32* tv /* N, un-encoded */,
                                    const typename F::elem* bsk /* prepared [n/2][3] GGSWs */) {
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  constexpr int N = 1 << LOGN;
  const int lane = c.tid();
  const int me = c.group();
  u32* acc = c.acc(0);
  {
    const u32 b_tilde = switch_modulus_2n(lwe[P.n], LOGN);
    const u32 m = (2u * N - b_tilde) & (2u * N - 1u);
    const int deg = (int)(m & (N - 1));
    const u32 flip = (m >> LOGN) & 1u;
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int j = r * T + lane;
      u32 val = 0;
      if (me == K) {
        const u32 t = tv[(j - deg) & (N - 1)] << P.tv_shift;
        val = (flip ^ (u32)(j < deg)) ? (0u - t) : t;
      }
      acc[j] = val;
    }
    c.poly_sync();
  }
  const size_t ggsw_words = (size_t)(K + 1) * P.levels * (K + 1) * F::kParts * (N >> F::kLogShrink);  // elements
  u32* stage = reinterpret_cast<u32*>(c.scratch());  // N u32 of my group's transpose buffer
#pragma unroll 1
  for (u32 pair = 0; pair < P.n / 2; ++pair) {
    const u32 a0 = c.uniform(switch_modulus_2n(lwe[2 * pair], LOGN));
    const u32 a1 = c.uniform(switch_modulus_2n(lwe[2 * pair + 1], LOGN));
    const u32 e[3] = {(a0 + a1) & (2u * N - 1u), a0, a1};
    auto src = [&](int j) -> u32 { return acc[j]; };
    auto out = [&](int, int j, u32 value) { stage[j] = value; };
    auto end_of_key = [&](int m) {
      c.poly_sync();  // the staged product is complete (G > 1: across the group's waves)
#pragma unroll
      for (int r = 0; r < E; ++r) {
        const int j = r * T + lane;
        acc[j] += monomial_coeff<LOGN>(stage, j, e[m]) - stage[j];
      }
      c.poly_sync();  // before the next inverse transform reuses the buffer
    };
    external_product_team_keys<F, LOGN, K, G, 3>(c, P, bsk + (size_t)pair * 3 * ggsw_words, ggsw_words, src, out,
                                                 end_of_key);
    if (c.exchange_buffers() != 2) c.poly_sync();
  }
}

// sample_extract at index 0 (bootstrapping.rs:122-156): wave c < K writes the N mask words of its
// polynomial, wave K writes the body word
template <int LOGN, int K, int G, class Ctx>
TFHE_HD void sample_extract_team(const Ctx& c, u32* out /* K*N + 1 */, int s = 0) {
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  constexpr int N = 1 << LOGN;
  const int lane = c.tid();
  const int me = c.group();
  const u32* acc = c.acc(s);
  if (me < K) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
      const int x = r * T + lane;
      out[me * N + x] = (x == 0) ? acc[0] : (0u - acc[N - x]);
    }
  } else if (lane == 0) {
    out[K * N] = acc[0];
  }
}

// Forward NTT of one u32 polynomial of the bootstrapping key into the prepared layout,
// pre-scaled by N^-1.
// LAYOUT_E: the shape the key is laid out for (key_layout_e; 0: this transform's own)
template <class F, int LOGN, int G, int LAYOUT_E = 0, class Ctx>
TFHE_HD void bsk_prepare_wave(const Ctx& c, const u32* poly, typename F::elem* spec /* [kParts][N >> kLogShrink] */,
                              typename F::elem n_inv) {
  typedef typename F::elem elem;
  constexpr int LT = LOGN - F::kLogShrink;
  constexpr int E = NttShape<LT, G>::kE;
  constexpr int CO = F::kCoeffs;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  constexpr int NS = 1 << LT;
  const int lane = c.tid();
#pragma unroll 1
  for (int part = 0; part < F::kParts; ++part) {
    elem x[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
      u32 w[CO];
#pragma unroll
      for (int q = 0; q < CO; ++q) w[q] = poly[(r + q * E) * T + lane];
      x[r] = F::from_key_words(w, part);
    }
    ntt_forward<F, LT, G>(c, x);
#pragma unroll
    for (int r = 0; r < E; ++r) spec[(size_t)part * NS + spectrum_slot<LT, G, (int)sizeof(elem), LAYOUT_E>(lane, r)] = F::scale_key(x[r], n_inv);
  }
}

// ---------------------------------------------------------------------------------------------
// sum_{i<k} masks[i] (*) sk[i], negacyclic mod 2^32, by one group of G waves: the a*s term of
// encrypt_glwe_zero (glwe.rs:197, poly_dot_product utils.rs:163-173) and of
// decrypt_glwe_ciphertext (glwe.rs:252).  The secret polynomials are BINARY (sample_binary): they
// play the digit role (small operand, F::from_digit), the uniform mask words the key role
// (F::from_key_word splits them into F::kParts limbs), so the field bound is the external
// product's with k rows of digits of magnitude 1.  out(j, value mod 2^32) once per coefficient.
// ---------------------------------------------------------------------------------------------
template <class F, int LOGN, int G, class Ctx, class Out>
TFHE_HD void glwe_mask_dot_key(const Ctx& c, u32 k, const u32* masks /* [k][N] */,
                               const u32* sk /* [k][N] */, typename F::elem n_inv, Out out) {
  typedef typename F::elem elem;
  constexpr int LT = LOGN - F::kLogShrink;
  constexpr int E = NttShape<LT, G>::kE;
  constexpr int CO = F::kCoeffs;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  constexpr int N = 1 << LOGN;
  constexpr int PARTS = F::kParts;
  const int lane = c.tid();

  elem accum[PARTS][E];
#pragma unroll
  for (int q = 0; q < PARTS; ++q)
#pragma unroll
    for (int r = 0; r < E; ++r) accum[q][r] = F::zero();

#pragma unroll 1
  for (u32 i = 0; i < k; ++i) {
    elem s[E];
#pragma unroll
    for (int r = 0; r < E; ++r) {
      u32 w[CO];
#pragma unroll
      for (int q = 0; q < CO; ++q) w[q] = sk[(size_t)i * N + (r + q * E) * T + lane];
      s[r] = F::from_digits(w);
    }
    ntt_forward<F, LT, G, true>(c, s);
#pragma unroll
    for (int r = 0; r < E; ++r) s[r] = F::scale_key(s[r], n_inv);
    static_for<0, PARTS>([&](auto part_c) {
      constexpr int q = decltype(part_c)::value;
      elem x[E];
#pragma unroll
      for (int r = 0; r < E; ++r) {
        u32 w[CO];
#pragma unroll
        for (int qq = 0; qq < CO; ++qq) w[qq] = masks[(size_t)i * N + (r + qq * E) * T + lane];
        x[r] = F::from_key_words(w, q);
      }
      ntt_forward<F, LT, G>(c, x);
#pragma unroll
      for (int r = 0; r < E; ++r)
        accum[q][r] = F::mul_add(F::kReduceSpectrum ? F::reduce(x[r]) : x[r], s[r], accum[q][r]);
    });
  }

  static_for<0, PARTS>([&](auto part_c) {
    constexpr int q = decltype(part_c)::value;
#pragma unroll
    for (int r = 0; r < E; ++r) accum[q][r] = F::before_inverse(accum[q][r]);
    ntt_inverse<F, LT, G>(c, accum[q]);
  });
#pragma unroll
  for (int r = 0; r < E; ++r) {
    elem parts[PARTS];
#pragma unroll
    for (int q = 0; q < PARTS; ++q) parts[q] = accum[q][r];
    u32 vals[CO];
    F::finish(parts, vals);
#pragma unroll
    for (int q = 0; q < CO; ++q) out((r + q * E) * T + lane, vals[q]);
  }
}

}  // namespace tfhe

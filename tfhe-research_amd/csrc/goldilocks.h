// goldilocks.h -- arithmetic in F_p, p = 2^64 - 2^32 + 1, for the exact negacyclic NTT.
//
// Why this field: the reference multiplies polynomials in Z_{2^32}[X]/(X^N+1) with an O(N^2)
// Toeplitz product (reference src/utils.rs:113-173).  Z_{2^32} has no 2N-th roots of unity, so the
// fast path computes the *integer* negacyclic convolution exactly in F_p and reduces mod 2^32 at
// the end.  Worst case |sum| = R*N*B*2^32 < 2^55 for every supported parameter set (checked at
// key-load time, see tfhe_hip.cpp), far below p/2, so the centred lift is exact.
//
// gfx950 note (profiles/r01_valu_issue_rates_gfx950.txt): v_mad_u64_u32, v_add_co/addc and
// v_mul_lo/hi all issue at ~4.4 cycles per wave, so cost = instruction count, not "multiplies".
//
// Representation: canonical, every stored value is in [0, p).  add/sub/mul take canonical inputs
// and return canonical outputs, so there are no rare double-carry corner cases to reason about.
#pragma once
#include "platform.h"

namespace tfhe {
namespace gl {

constexpr u64 P = 0xFFFFFFFF00000001ull;
constexpr u64 EPS = 0xFFFFFFFFull;  // 2^64 mod p = 2^32 - 1 = 2^64 - p

// a + b mod p, a, b < p
TFHE_HD u64 add(u64 a, u64 b) {
  u64 s = a + b;
  u64 d = s + EPS;  // s - p (mod 2^64)
  // a + b >= p  <=>  the 64-bit add carried, or s + (2^64 - p) carries
  return ((s < a) | (d < s)) ? d : s;
}

// a - b mod p, a, b < p
TFHE_HD u64 sub(u64 a, u64 b) {
  u64 d = a - b;
  return d - ((a < b) ? EPS : 0ull);  // borrow: add p == subtract EPS (mod 2^64)
}

// x = (hi, lo) 128-bit -> canonical residue.  2^64 = EPS, 2^96 = -1 (mod p).
TFHE_HD u64 reduce128(u64 lo, u64 hi) {
  u32 x2 = (u32)hi, x3 = (u32)(hi >> 32);
  // t = lo - x3 (mod p); a borrow means we are 2^64 too high in Z, i.e. subtract EPS
  u64 t = lo - (u64)x3;
  t -= (lo < (u64)x3) ? EPS : 0ull;
  // r = t + x2 * EPS; x2*EPS < p so at most one wrap, and the wrapped sum + EPS cannot wrap
  u64 u = ((u64)x2 << 32) - (u64)x2;
  u64 r = t + u;
  r += (r < u) ? EPS : 0ull;
  // canonicalise: r >= p  <=>  r + EPS wraps
  u64 c = r + EPS;
  return (c < r) ? c : r;
}

TFHE_HD u64 mul(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  // four 32x32+64 multiply-adds (v_mad_u64_u32); none of the sums can exceed 64 bits
  u64 t0 = (u64)a0 * b0;
  u64 t1 = (u64)a0 * b1 + (t0 >> 32);
  u64 t2 = (u64)a1 * b0 + (u32)t1;
  u64 t3 = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
  u64 lo = (t2 << 32) | (u32)t0;
  return reduce128(lo, t3);
}

// small signed integer (|d| < 2^31, given as a wrapped u32) -> canonical field element
TFHE_HD u64 from_i32(u32 d) {
  return ((i32)d < 0) ? P - (u64)(u32)(0u - d) : (u64)d;
}

// centred lift of a residue v whose true integer x satisfies |x| < 2^62, reduced mod 2^32.
// x >= 0: v = x (top bit clear) -> lo32(v).  x < 0: v = x + p has its top bit set and
// p = 1 (mod 2^32) -> lo32(v) - 1.
TFHE_HD u32 lift_mod_2_32(u64 v) { return (u32)v - (u32)(v >> 63); }

TFHE_HD u64 pow(u64 base, u64 e) {
  u64 r = 1;
  while (e) {
    if (e & 1) r = mul(r, base);
    base = mul(base, base);
    e >>= 1;
  }
  return r;
}

TFHE_HD u64 inv(u64 a) { return pow(a, P - 2); }

constexpr u64 GENERATOR = 7;  // multiplicative generator of F_p^*

// primitive 2^k-th root of unity (k <= 32): 7^((p-1)/2^k)
TFHE_HD u64 root_of_unity(int log2_order) { return pow(GENERATOR, (P - 1) >> log2_order); }

}  // namespace gl
}  // namespace tfhe

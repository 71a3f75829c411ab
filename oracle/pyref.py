"""Independent numpy restatement of the reference's bootstrapping path (TEST INFRASTRUCTURE).

Written from the reference text separately from oracle/tfhe_oracle.c and with different
mechanics (vectorised numpy, negacyclic products through np.convolve on uint64, closed-form digit
extraction), so that agreement between the two restatements is evidence rather than tautology.
Only tests/ may import this.  Small sizes only (it is slow).

Reference lines followed: decomposer.rs:27-80, utils.rs:13-33,183-207,221-236, glwe.rs:20-108,
141-151,232-243, ggsw.rs:132-178, bootstrapping.rs:58-156, key_switching.rs:63-103,
test_vector.rs:38-67.
"""
from __future__ import annotations

import numpy as np

M32 = np.uint64(0xFFFFFFFF)


def round_value(v, log_base, levels):
    v = np.asarray(v, dtype=np.uint64)
    ig = 32 - log_base * levels
    if ig == 0:
        return v.astype(np.uint32)
    msb = (v >> np.uint64(ig - 1)) & np.uint64(1)
    return ((((v >> np.uint64(ig)) + msb) << np.uint64(ig)) & M32).astype(np.uint32)


def decompose(v, log_base, levels):
    """-> array (..., levels) of wrapped-u32 digits, MSB first (decomposer.rs:42-80)."""
    v = round_value(v, log_base, levels).astype(np.int64)
    L = 32 // log_base
    B = 1 << log_base
    carry = np.zeros_like(v)
    limbs = []
    for l in range(L):
        res = ((v >> (log_base * l)) & (B - 1)) + carry
        cm = res & (B >> 1)
        res = res - 2 * cm
        carry = cm >> (log_base - 1)
        limbs.append(res)
    limbs = limbs[::-1][:levels]
    return (np.stack(limbs, axis=-1) & 0xFFFFFFFF).astype(np.uint32)


def switch_modulus(v, log_from, log_to):
    v = np.asarray(v, dtype=np.uint64)
    d = np.uint64(1 << (log_from - log_to))
    r = v // d + ((v % d + (d >> np.uint64(1))) // d)
    return (r % np.uint64(1 << log_to)).astype(np.uint32)


def negacyclic_mul(a, b):
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    n = a.size
    with np.errstate(over="ignore"):
        full = np.convolve(a, b)  # wraps mod 2^64, which preserves the value mod 2^32
        lo = full[:n].copy()
        lo[: n - 1] -= full[n:]
    return (lo & M32).astype(np.uint32)


def mul_monomial(p, idx):
    p = np.asarray(p, dtype=np.uint32)
    n = p.shape[-1]
    m = idx % (2 * n)  # python % is non-negative
    flip, deg = divmod(m, n)
    out = np.roll(p, deg, axis=-1).astype(np.int64)
    out[..., :deg] *= -1
    if flip:
        out = -out
    return (out & 0xFFFFFFFF).astype(np.uint32)


def decompose_glwe(glwe, log_base, levels):
    glwe = np.asarray(glwe, dtype=np.uint32)
    d = decompose(glwe, log_base, levels)  # (rows, N, levels)
    return np.concatenate([d[r].T for r in range(glwe.shape[0])], axis=0)  # (rows*levels, N)


def external_product(ggsw, glwe, log_base, levels):
    ggsw = np.asarray(ggsw, dtype=np.uint32)
    digits = decompose_glwe(glwe, log_base, levels)
    k1 = ggsw.shape[1]
    out = np.zeros((k1, ggsw.shape[2]), dtype=np.uint64)
    for c in range(k1):
        for r in range(ggsw.shape[0]):
            out[c] += negacyclic_mul(digits[r], ggsw[r, c])
    return (out & M32).astype(np.uint32)


def cmux(ggsw, ct0, ct1, log_base, levels):
    ct0 = np.asarray(ct0, dtype=np.uint32)
    diff = (np.asarray(ct1, dtype=np.uint32) - ct0).astype(np.uint32)
    return (external_product(ggsw, diff, log_base, levels) + ct0).astype(np.uint32)


def sample_extract0(glwe):
    glwe = np.asarray(glwe, dtype=np.uint32)
    parts = []
    for row in glwe[:-1]:
        neg = (np.uint32(0) - row[:0:-1]).astype(np.uint32)
        parts.append(np.concatenate([row[:1], neg]))
    parts.append(glwe[-1, :1])
    return np.concatenate(parts)


def key_switch(lwe, ksk, log_base, levels):
    lwe = np.asarray(lwe, dtype=np.uint32)
    ksk = np.asarray(ksk, dtype=np.uint64)
    digits = decompose(lwe[:-1], log_base, levels).reshape(-1).astype(np.uint64)
    with np.errstate(over="ignore"):
        s = (digits[:, None] * ksk).sum(axis=0)
    out = ((np.uint64(0) - s) & M32)
    out[-1] = (out[-1] + np.uint64(lwe[-1])) & M32
    return out.astype(np.uint32)


def test_from_lut(lut, log_n, log_p):
    rep = (1 << log_n) >> log_p
    tv = np.repeat(np.asarray(lut, dtype=np.int64), rep)
    head = tv[: rep // 2]
    head[head != 0] = (1 << log_p) - head[head != 0]
    return np.roll(tv, -(rep // 2)).astype(np.uint32)


def bootstrap(lwe, bsk, ksk, tv, *, log_n, log_p, padding, pbs, ks, return_acc=False):
    """pbs = (log_base, levels), ks likewise; bsk [n][R][k+1][N]."""
    lwe = np.asarray(lwe, dtype=np.uint32)
    bsk = np.asarray(bsk, dtype=np.uint32)
    n = lwe.size - 1
    k1, N = bsk.shape[2], bsk.shape[3]
    a = switch_modulus(lwe, 32, log_n + 1)
    acc = np.zeros((k1, N), dtype=np.uint32)
    acc[-1] = (np.asarray(tv, dtype=np.uint64) << np.uint64(32 - log_p - padding)).astype(np.uint32)
    acc = mul_monomial(acc, -int(a[n]))
    for i in range(n):
        acc = cmux(bsk[i], acc, mul_monomial(acc, int(a[i])), *pbs)
    if return_acc:
        return acc
    return key_switch(sample_extract0(acc), ksk, *ks)

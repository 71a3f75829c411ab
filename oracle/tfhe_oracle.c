/*
 * tfhe_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See tfhe_oracle.h.
 *
 * Literal restatement of the reference's bootstrapping path.  All uint32_t arithmetic wraps,
 * which is what the Rust code does in a release build (the reference mixes wrapping_* calls with
 * plain operators, e.g. utils.rs:195, key_switching.rs:88, so only release mode is total).
 */
#include "tfhe_oracle.h"

#include <stdlib.h>
#include <string.h>

static int g_poly_mul_mode = 0;

void orc_set_poly_mul_mode(int mode) { g_poly_mul_mode = mode; }
int orc_get_poly_mul_mode(void) { return g_poly_mul_mode; }

static size_t degree_of(const orc_params *p) { return (size_t)1 << p->glwe_poly_degree; }
static size_t ggsw_rows(const orc_params *p) {
    return (size_t)(p->glwe_dimension + 1) * p->pbs_decomposer.levels;
}
size_t orc_ggsw_words(const orc_params *p) {
    return ggsw_rows(p) * (p->glwe_dimension + 1) * degree_of(p);
}
size_t orc_bsk_words(const orc_params *p) { return (size_t)p->lwe_dimension * orc_ggsw_words(p); }
size_t orc_ksk_words(const orc_params *p) {
    /* key_switching.rs:29-32 with from = lwe_params_post_pbs (lib.rs:58-66), to = lwe_params */
    return degree_of(p) * p->glwe_dimension * p->ks_decomposer.levels *
           ((size_t)p->lwe_dimension + 1);
}

static int decomposer_validate(const orc_decomposer *d) {
    if (d->log_q != 32) return 1;                       /* code is hard-typed to u32 */
    if (d->log_base == 0 || d->log_base >= 32) return 2; /* 1 << (log_base-1), 1 << log_base */
    if (d->levels == 0) return 3;
    if (d->log_base * d->levels > d->log_q) return 4;   /* usize underflow, decomposer.rs:28 */
    if (d->levels > d->log_q / d->log_base) return 5;   /* truncation loop never ends, :74-77 */
    return 0;
}

int orc_params_validate(const orc_params *p) {
    int e;
    if (p->log_q != 32) return 10;
    if ((e = decomposer_validate(&p->pbs_decomposer))) return 20 + e;
    if ((e = decomposer_validate(&p->ks_decomposer))) return 30 + e;
    if (p->glwe_poly_degree + 1 >= 32 || p->glwe_poly_degree == 0) return 11;
    if (p->log_p + p->padding_bits > 32) return 12;     /* glwe.rs:145 shift */
    if (p->log_p > p->glwe_poly_degree) return 13;      /* test_vector.rs:46 repetition >= 1 */
    if (p->lwe_dimension == 0 || p->glwe_dimension == 0) return 14;
    return 0;
}

void orc_params_default(orc_params *p, int cfg_test) {
    /* lib.rs:77-99 (cfg(test)) and lib.rs:101-123 */
    p->glwe_dimension = 2;
    p->glwe_poly_degree = 9;
    p->lwe_dimension = cfg_test ? 4 : 722;
    p->log_p = 2;
    p->log_q = 32;
    p->ks_decomposer.log_base = 4;
    p->ks_decomposer.levels = 5;
    p->ks_decomposer.log_q = 32;
    p->pbs_decomposer.log_base = 4;
    p->pbs_decomposer.levels = 6;
    p->pbs_decomposer.log_q = 32;
    p->padding_bits = 1;
    p->lwe_std_dev = 0.000013071021089943935;
    p->glwe_std_dev = 0.00000004990272175010415;
}

/* ------------------------------------------------------------------ decomposer.rs */

/* decomposer.rs:27-40 */
/* 0 (default) = the reference's literal decomposer.  1 = "aligned" extension for bases with
 * beta^l != q (notes/TFHE.md:116,407 leave it as a TODO): gadget factors q/beta^(level+1) and limbs
 * taken from the top of the word, so that a base whose log does not divide 32 still recomposes.
 * Identical to the literal decomposer whenever log_base divides log_q. */
static int g_decomposer_aligned = 0;
void orc_set_decomposer_aligned(int aligned) { g_decomposer_aligned = aligned != 0; }
int orc_get_decomposer_aligned(void) { return g_decomposer_aligned; }
/* bit position of the gadget factor of `level` (MSB-first): factor = 1 << orc_gadget_shift() */
uint32_t orc_gadget_shift(const orc_decomposer *d, uint32_t level) {
    uint32_t top = g_decomposer_aligned ? d->log_q : d->log_base * (d->log_q / d->log_base);
    return top - d->log_base * (level + 1);
}

uint32_t orc_round_value(const orc_decomposer *d, uint32_t value) {
    uint32_t ignored_bits = d->log_q - d->log_base * d->levels;
    if (ignored_bits == 0) return value;
    uint32_t ignored_mask = ((uint32_t)1 << ignored_bits) - 1;
    uint32_t ignored_value = value & ignored_mask;
    uint32_t ignored_msb = ignored_value >> (ignored_bits - 1);
    return ((value >> ignored_bits) + ignored_msb) << ignored_bits;
}

/* decomposer.rs:42-80.  Digits are pushed LSB first over ALL log_q/log_base limbs at bit offsets
 * log_base*l counted from bit 0 (not from the first kept bit), reversed, then truncated to
 * `levels` from the tail.  A limb equal to B (B-1 plus carry) keeps the value B and emits no
 * carry, because its B/2 bit is clear. */
int orc_decompose(const orc_decomposer *d, uint32_t value, uint32_t *out) {
    uint32_t limbs[32];
    uint32_t count = 0;
    if (decomposer_validate(d)) return 1;
    value = orc_round_value(d, value);
    uint32_t log_base = d->log_base;
    uint32_t base_mask = ((uint32_t)1 << log_base) - 1;
    uint32_t base_by_2_mask = (uint32_t)1 << (log_base - 1);
    uint32_t carry = 0;
    /* literal: limbs at bit log_base*l for all floor(log_q/log_base) limbs.  Aligned extension
     * (NOT the reference): the `levels` kept limbs sit directly below bit log_q. */
    uint32_t n_limbs = g_decomposer_aligned ? d->levels : d->log_q / d->log_base;
    uint32_t bit0 = g_decomposer_aligned ? d->log_q - d->log_base * d->levels : 0;
    for (uint32_t l = 0; l < n_limbs; ++l) {
        uint32_t res = ((value >> (bit0 + log_base * l)) & base_mask) + carry;
        uint32_t carry_mask = res & base_by_2_mask;
        res = res - (carry_mask << 1);
        carry = carry_mask >> (log_base - 1);
        limbs[count++] = res;
    }
    /* reverse (:69) then drop from the tail until `levels` remain (:72-77) */
    for (uint32_t i = 0; i < d->levels; ++i) out[i] = limbs[count - 1 - i];
    return 0;
}

/* decomposer.rs:83-95 */
uint32_t orc_recompose(const orc_decomposer *d, const uint32_t *legs) {
    uint32_t value = 0;
    for (uint32_t index = 0; index < d->levels; ++index) {
        uint32_t leg_shifted = legs[index] << (d->log_base * (d->levels - 1 - index));
        value += leg_shifted;
    }
    uint32_t ignored_bits = d->log_q - d->log_base * d->levels;
    return ignored_bits >= 32 ? 0 : value << ignored_bits;
}

/* ------------------------------------------------------------------ utils.rs */

/* utils.rs:13-18 */
uint32_t orc_integer_division(uint32_t a, uint32_t divisor) {
    uint32_t rational = a / divisor;
    uint32_t fractional = a % divisor;
    return rational + ((fractional + (divisor >> 1)) / divisor);
}

/* utils.rs:23-33 */
void orc_switch_modulus(const uint32_t *values, size_t len, uint32_t log_from, uint32_t log_to,
                        uint32_t *out) {
    for (size_t i = 0; i < len; ++i) {
        uint32_t v = orc_integer_division(values[i], (uint32_t)1 << (log_from - log_to));
        out[i] = v % ((uint32_t)1 << log_to);
    }
}

/* utils.rs:113-153: row i = p[i], p[i-1], ..., p[0], -p[n-1], ..., -p[i+1] */
void orc_teoplitz(const uint32_t *p, size_t n, uint32_t *matrix) {
    size_t w = 0;
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = i + 1; j-- > 0;) matrix[w++] = p[j];
        for (size_t j = n; j-- > i + 1;) matrix[w++] = (uint32_t)0 - p[j];
    }
}

/* utils.rs:221-236 */
void orc_school_book_negacylic_mul(const uint32_t *p0, const uint32_t *p1, size_t n,
                                   uint32_t *res) {
    for (size_t i = 0; i < n; ++i) {
        uint32_t acc = 0;
        for (size_t j = 0; j < i + 1; ++j) acc += p0[j] * p1[i - j];
        for (size_t j = i + 1; j < n; ++j) acc -= p0[j] * p1[n - (j - i)];
        res[i] = acc;
    }
}

/* utils.rs:155-160: Toeplitz(p0) . p1 (ndarray 0.15.6 Array2<u32>::dot(Array1<u32>) = plain
 * row-by-vector inner products with wrapping u32 arithmetic in release mode) */
void orc_poly_mul(const uint32_t *p0, const uint32_t *p1, size_t n, uint32_t *res) {
    if (g_poly_mul_mode == 1) {
        orc_school_book_negacylic_mul(p0, p1, n, res);
        return;
    }
    uint32_t *matrix = (uint32_t *)malloc(n * n * sizeof(uint32_t));
    orc_teoplitz(p0, n, matrix);
    for (size_t i = 0; i < n; ++i) {
        const uint32_t *row = matrix + i * n;
        uint32_t acc = 0;
        for (size_t c = 0; c < n; ++c) acc += row[c] * p1[c];
        res[i] = acc;
    }
    free(matrix);
}

/* utils.rs:163-173 */
void orc_poly_dot_product(const uint32_t *p0, const uint32_t *p1, size_t p1_row_stride,
                          size_t rows, size_t n, uint32_t *res) {
    uint32_t *r = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_poly_mul(p0, p1, n, res);
    for (size_t row = 1; row < rows; ++row) {
        orc_poly_mul(p0 + row * n, p1 + row * p1_row_stride, n, r);
        for (size_t i = 0; i < n; ++i) res[i] += r[i];
    }
    free(r);
}

/* utils.rs:183-207.  `monomial_index as usize % (2*n)`: a negative isize becomes 2^64 - |x|,
 * and 2n divides 2^64, so the residue is (2n - |x| mod 2n) mod 2n. */
void orc_poly_mul_monomial(const uint32_t *p0, size_t n, int64_t monomial_index, uint32_t *res) {
    size_t idx = (size_t)((uint64_t)monomial_index % (uint64_t)(2 * n));
    size_t flip_sign = idx / n;
    size_t degree = idx % n;
    uint32_t factor = flip_sign ? 0xFFFFFFFFu : 1u; /* u32::MAX.pow(flip_sign) */
    /* rotate_right(degree): new[(i + degree) % n] = old[i] */
    for (size_t i = 0; i < n; ++i) res[(i + degree) % n] = p0[i] * factor;
    for (size_t i = 0; i < degree; ++i) res[i] = (uint32_t)0 - res[i];
}

/* ------------------------------------------------------------------ glwe.rs */

/* glwe.rs:20-34 */
void orc_glwe_mul_monomial(const uint32_t *glwe, size_t rows, size_t n, int64_t monomial_index,
                           uint32_t *out) {
    for (size_t r = 0; r < rows; ++r)
        orc_poly_mul_monomial(glwe + r * n, n, monomial_index, out + r * n);
}

void orc_glwe_add_assign(uint32_t *lhs, const uint32_t *rhs, size_t len) {
    for (size_t i = 0; i < len; ++i) lhs[i] += rhs[i];
}

void orc_glwe_sub_assign(uint32_t *lhs, const uint32_t *rhs, size_t len) {
    for (size_t i = 0; i < len; ++i) lhs[i] -= rhs[i];
}

/* glwe.rs:69-85: column `term` of the (levels x n) matrix = decompose(poly[term]) */
int orc_decompose_poly(const uint32_t *poly, size_t n, const orc_decomposer *d, uint32_t *out) {
    uint32_t legs[32];
    for (size_t term = 0; term < n; ++term) {
        if (orc_decompose(d, poly[term], legs)) return 1;
        for (uint32_t l = 0; l < d->levels; ++l) out[(size_t)l * n + term] = legs[l];
    }
    return 0;
}

/* glwe.rs:90-108: blocks of `levels` rows, one block per GLWE polynomial, in row order */
int orc_decompose_glwe_ciphertext(const uint32_t *glwe, size_t rows, size_t n,
                                  const orc_decomposer *d, uint32_t *out) {
    for (size_t r = 0; r < rows; ++r)
        if (orc_decompose_poly(glwe + r * n, n, d, out + r * d->levels * n)) return 1;
    return 0;
}

/* glwe.rs:141-151: asserts m < 2^log_p; shorter messages leave trailing zeros */
int orc_glwe_encode_message(const orc_params *p, const uint32_t *message, size_t len,
                            uint32_t *out) {
    size_t n = degree_of(p);
    memset(out, 0, n * sizeof(uint32_t));
    for (size_t i = 0; i < len && i < n; ++i) {
        if (p->log_p < 32 && message[i] >= ((uint32_t)1 << p->log_p)) return 1; /* assert! */
        out[i] = message[i] << (p->log_q - (p->log_p + p->padding_bits));
    }
    return 0;
}

/* glwe.rs:232-243 */
void orc_trivial_encrypt_glwe_plaintext(const orc_params *p, const uint32_t *plaintext,
                                        uint32_t *out) {
    size_t n = degree_of(p);
    memset(out, 0, (size_t)(p->glwe_dimension + 1) * n * sizeof(uint32_t));
    memcpy(out + (size_t)p->glwe_dimension * n, plaintext, n * sizeof(uint32_t));
}

/* ------------------------------------------------------------------ ggsw.rs */

/* ggsw.rs:132-161 */
int orc_external_product(const orc_params *p, const uint32_t *ggsw, const uint32_t *glwe,
                         uint32_t *out) {
    size_t n = degree_of(p);
    size_t k1 = p->glwe_dimension + 1;
    size_t rows = ggsw_rows(p);
    uint32_t *decomposed = (uint32_t *)malloc(rows * n * sizeof(uint32_t));
    if (orc_decompose_glwe_ciphertext(glwe, k1, n, &p->pbs_decomposer, decomposed)) {
        free(decomposed);
        return 1;
    }
    for (size_t ggsw_col = 0; ggsw_col < k1; ++ggsw_col) {
        /* col = ggsw[.., ggsw_col, ..]: row r lives at ggsw + (r*(k+1) + col)*N */
        orc_poly_dot_product(decomposed, ggsw + ggsw_col * n, k1 * n, rows, n,
                             out + ggsw_col * n);
    }
    free(decomposed);
    return 0;
}

/* ggsw.rs:164-178 */
int orc_cmux(const orc_params *p, const uint32_t *ggsw, const uint32_t *ct0, uint32_t *ct1,
             uint32_t *out) {
    size_t len = (size_t)(p->glwe_dimension + 1) * degree_of(p);
    orc_glwe_sub_assign(ct1, ct0, len);
    if (orc_external_product(p, ggsw, ct1, out)) return 1;
    orc_glwe_add_assign(out, ct0, len);
    return 0;
}

/* ------------------------------------------------------------------ bootstrapping.rs */

/* bootstrapping.rs:122-156 */
int orc_sample_extract(const orc_params *p, const uint32_t *glwe, size_t sample_index,
                       uint32_t *out) {
    size_t n = degree_of(p);
    size_t k = p->glwe_dimension;
    if (!(sample_index < n)) return 1; /* assert! :127 */
    uint32_t lwe_b = glwe[k * n + sample_index];
    size_t w = 0;
    for (size_t row = 0; row < k; ++row) {
        const uint32_t *poly = glwe + row * n;
        for (size_t i = sample_index + 1; i-- > 0;) out[w++] = poly[i];
        for (size_t i = n; i-- > sample_index + 1;) out[w++] = (uint32_t)0 - poly[i];
    }
    out[w] = lwe_b;
    return 0;
}

/* key_switching.rs:63-103 */
int orc_key_switch_lwe(const uint32_t *lwe, size_t from_n, size_t to_n, const orc_decomposer *d,
                       const uint32_t *ksk, uint32_t *out) {
    uint32_t legs[32];
    size_t width = to_n + 1;
    memset(out, 0, width * sizeof(uint32_t));
    for (size_t i = 0; i < from_n; ++i) {
        if (orc_decompose(d, lwe[i], legs)) return 1;
        for (uint32_t l = 0; l < d->levels; ++l) {
            const uint32_t *row = ksk + (i * d->levels + l) * width;
            uint32_t a_ij = legs[l];
            for (size_t c = 0; c < width; ++c) out[c] += a_ij * row[c]; /* scaled_add :88 */
        }
    }
    for (size_t c = 0; c < width; ++c) out[c] = (uint32_t)0 - out[c];
    out[to_n] += lwe[from_n];
    return 0;
}

/* bootstrapping.rs:67-105 */
int orc_blind_rotate(const orc_params *p, const uint32_t *lwe_ct, const uint32_t *bsk,
                     const uint32_t *test_vector_poly, uint32_t *acc,
                     orc_bootstrap_trace *trace) {
    size_t n = degree_of(p);
    size_t k1 = p->glwe_dimension + 1;
    size_t glwe_len = k1 * n;
    size_t lwe_n = p->lwe_dimension;
    size_t ggsw_len = orc_ggsw_words(p);
    int rc = 0;

    uint32_t *approximate_lwe = (uint32_t *)malloc((lwe_n + 1) * sizeof(uint32_t));
    uint32_t *encoded = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *v_x = (uint32_t *)malloc(glwe_len * sizeof(uint32_t));
    uint32_t *c1 = (uint32_t *)malloc(glwe_len * sizeof(uint32_t));
    uint32_t *next = (uint32_t *)malloc(glwe_len * sizeof(uint32_t));

    /* :67-71 */
    orc_switch_modulus(lwe_ct, lwe_n + 1, p->log_q, p->glwe_poly_degree + 1, approximate_lwe);
    if (trace && trace->approximate_lwe)
        memcpy(trace->approximate_lwe, approximate_lwe, (lwe_n + 1) * sizeof(uint32_t));

    /* :79-86 */
    int64_t b_approx = -(int64_t)approximate_lwe[lwe_n];
    if (orc_glwe_encode_message(p, test_vector_poly, n, encoded)) { rc = 1; goto done; }
    orc_trivial_encrypt_glwe_plaintext(p, encoded, v_x);
    orc_glwe_mul_monomial(v_x, k1, n, b_approx, acc);
    if (trace && trace->acc_init) memcpy(trace->acc_init, acc, glwe_len * sizeof(uint32_t));

    /* :90-105 */
    for (size_t i = 0; i < lwe_n; ++i) {
        orc_glwe_mul_monomial(acc, k1, n, (int64_t)approximate_lwe[i], c1);
        if (orc_cmux(p, bsk + i * ggsw_len, acc, c1, next)) { rc = 1; goto done; }
        memcpy(acc, next, glwe_len * sizeof(uint32_t));
        if (trace && trace->acc_after_each)
            memcpy(trace->acc_after_each + i * glwe_len, acc, glwe_len * sizeof(uint32_t));
    }
    if (trace && trace->acc_final) memcpy(trace->acc_final, acc, glwe_len * sizeof(uint32_t));

done:
    free(approximate_lwe);
    free(encoded);
    free(v_x);
    free(c1);
    free(next);
    return rc;
}

/* bootstrapping.rs:58-120 */
int orc_bootstrap(const orc_params *p, const uint32_t *lwe_ct, const uint32_t *bsk,
                  const uint32_t *ksk, const uint32_t *test_vector_poly, uint32_t *out,
                  orc_bootstrap_trace *trace) {
    if (orc_params_validate(p)) return 2;
    size_t n = degree_of(p);
    size_t k = p->glwe_dimension;
    uint32_t *acc = (uint32_t *)malloc((k + 1) * n * sizeof(uint32_t));
    uint32_t *bootstrapped_lwe = (uint32_t *)malloc((k * n + 1) * sizeof(uint32_t));
    int rc = orc_blind_rotate(p, lwe_ct, bsk, test_vector_poly, acc, trace);
    if (!rc) rc = orc_sample_extract(p, acc, 0, bootstrapped_lwe); /* :108 */
    if (!rc && trace && trace->extracted_lwe)
        memcpy(trace->extracted_lwe, bootstrapped_lwe, (k * n + 1) * sizeof(uint32_t));
    /* :111-117: from = lwe_params_post_pbs (dimension N*k, lib.rs:60), to = lwe_params */
    if (!rc)
        rc = orc_key_switch_lwe(bootstrapped_lwe, n * k, p->lwe_dimension, &p->ks_decomposer,
                                ksk, out);
    free(acc);
    free(bootstrapped_lwe);
    return rc;
}

/* ------------------------------------------------------------------ test_vector.rs */

/* test_vector.rs:38-67 */
int orc_construct_test_from_lut(const orc_params *p, const uint32_t *lut, size_t lut_len,
                                uint32_t *out) {
    uint32_t plaintext_modulus = (uint32_t)1 << p->log_p;
    if (lut_len != plaintext_modulus) return 1; /* assert! :41 */
    size_t n = degree_of(p);
    size_t repetition = n / ((size_t)1 << p->log_p);
    size_t len = repetition * lut_len;
    uint32_t *tv = (uint32_t *)malloc((len ? len : 1) * sizeof(uint32_t));
    size_t w = 0;
    for (size_t v = 0; v < lut_len; ++v)
        for (size_t r = 0; r < repetition; ++r) tv[w++] = lut[v];
    for (size_t i = 0; i < repetition / 2; ++i)
        if (tv[i] != 0) tv[i] = plaintext_modulus - tv[i];
    /* rotate_left(mid): new[i] = old[(i + mid) % len] */
    size_t mid = repetition / 2;
    for (size_t i = 0; i < len; ++i) out[i] = tv[(i + mid) % len];
    free(tv);
    return 0;
}

/* test_vector.rs:23-35 */
int orc_construct_identity_test_vector(const orc_params *p, uint32_t *out) {
    uint32_t pm = (uint32_t)1 << p->log_p;
    uint32_t *lut = (uint32_t *)malloc(pm * sizeof(uint32_t));
    for (uint32_t i = 0; i < pm; ++i) lut[i] = i;
    int rc = orc_construct_test_from_lut(p, lut, pm, out);
    free(lut);
    return rc;
}

/* test_vector.rs:5-20: lookup_table[i] = f((i >> 1) & 1, i & 1) */
int orc_construct_test_vector_boolean(const orc_params *p, const uint32_t truth[4],
                                      uint32_t *out) {
    uint32_t pm = (uint32_t)1 << p->log_p;
    uint32_t *lut = (uint32_t *)malloc(pm * sizeof(uint32_t));
    for (uint32_t i = 0; i < pm; ++i) lut[i] = truth[(((i >> 1) & 1) << 1) | (i & 1)];
    int rc = orc_construct_test_from_lut(p, lut, pm, out);
    free(lut);
    return rc;
}

/* ------------------------------------------------------------------ lwe.rs / boolean.rs */

void orc_lwe_add(const uint32_t *a, const uint32_t *b, size_t len, uint32_t *out) {
    for (size_t i = 0; i < len; ++i) out[i] = a[i] + b[i];
}

void orc_lwe_mul_scalar(const uint32_t *a, uint32_t s, size_t len, uint32_t *out) {
    for (size_t i = 0; i < len; ++i) out[i] = a[i] * s;
}

/* boolean.rs:9-30: ct_in = &(ct1 * 2u32) + ct0; bootstrap with the closure's test vector */
int orc_boolean_gate(const orc_params *p, const uint32_t truth[4], const uint32_t *ct0,
                     const uint32_t *ct1, const uint32_t *bsk, const uint32_t *ksk,
                     uint32_t *out) {
    size_t n = degree_of(p);
    size_t len = (size_t)p->lwe_dimension + 1;
    uint32_t *tv = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint32_t *ct_in = (uint32_t *)malloc(len * sizeof(uint32_t));
    int rc = orc_construct_test_vector_boolean(p, truth, tv);
    if (!rc) {
        orc_lwe_mul_scalar(ct1, 2u, len, ct_in);
        orc_lwe_add(ct_in, ct0, len, ct_in);
        rc = orc_bootstrap(p, ct_in, bsk, ksk, tv, out, NULL);
    }
    free(tv);
    free(ct_in);
    return rc;
}

/*
 * tfhe_oracle_crypto.c -- CPU ORACLE, host-side key generation / encryption / decryption.
 * Test infrastructure only (see tfhe_oracle.h).  Off the hot path: it exists so the parity tests
 * can assert decrypt-correctness the way the reference's own tests do.  Structure follows the
 * reference line by line; the RNG is ours (seeded SplitMix64; the reference uses thread_rng()).
 */
#include "tfhe_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static size_t degree_of(const orc_params *p) { return (size_t)1 << p->glwe_poly_degree; }

void orc_rng_seed(orc_rng *r, uint64_t seed) {
    r->state = seed;
    r->literal_noise = 0;
    r->have_spare = 0;
    r->spare = 0.0;
}

uint64_t orc_rng_next_u64(orc_rng *r) {
    uint64_t z = (r->state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint32_t orc_rng_next_u32(orc_rng *r) { return (uint32_t)(orc_rng_next_u64(r) >> 32); }

void orc_fill_uniform_u32(orc_rng *r, uint32_t *out, size_t len) {
    for (size_t i = 0; i < len; ++i) out[i] = orc_rng_next_u32(r);
}

/* utils.rs:36-41.  `frac as u32` saturates: negative -> 0, >= 2^32 -> u32::MAX.  With
 * literal == 0 a negative value wraps mod 2^32 instead (a proper two-sided error). */
uint32_t orc_f64_to_torus(double v, int literal) {
    double frac = v - round(v);
    frac *= 4294967296.0;
    frac = round(frac);
    if (literal) {
        if (!(frac > 0.0)) return 0;
        if (frac >= 4294967295.0) return 0xFFFFFFFFu;
        return (uint32_t)frac;
    }
    int64_t t = (int64_t)frac;
    return (uint32_t)(uint64_t)t;
}

static double uniform01(orc_rng *r) {
    return ((double)(orc_rng_next_u64(r) >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

/* utils.rs:43-54 (Normal(0, std_dev) through Box-Muller) */
uint32_t orc_sample_gaussian(orc_rng *r, double std_dev) {
    double g;
    if (r->have_spare) {
        r->have_spare = 0;
        g = r->spare;
    } else {
        double u1 = uniform01(r), u2 = uniform01(r);
        double mag = sqrt(-2.0 * log(u1));
        g = mag * cos(6.283185307179586 * u2);
        r->spare = mag * sin(6.283185307179586 * u2);
        r->have_spare = 1;
    }
    return orc_f64_to_torus(g * std_dev, r->literal_noise);
}

/* utils.rs:68-93: bits of successive random bytes, LSB first */
void orc_sample_binary(orc_rng *r, uint32_t *out, size_t len) {
    uint8_t curr_byte = (uint8_t)orc_rng_next_u32(r);
    int bit_index = 0;
    for (size_t i = 0; i < len; ++i) {
        out[i] = (curr_byte >> bit_index) & 1u;
        if (++bit_index == 8) {
            curr_byte = (uint8_t)orc_rng_next_u32(r);
            bit_index = 0;
        }
    }
}

void orc_lwe_secret_key_random(const orc_params *p, orc_rng *r, uint32_t *sk) {
    orc_sample_binary(r, sk, p->lwe_dimension);
}

void orc_glwe_secret_key_random(const orc_params *p, orc_rng *r, uint32_t *sk) {
    orc_sample_binary(r, sk, (size_t)p->glwe_dimension * degree_of(p));
}

/* lwe.rs:81-92 */
int orc_lwe_encode(const orc_params *p, uint32_t m, uint32_t *pt) {
    if (p->log_p < 32 && m >= ((uint32_t)1 << p->log_p)) return 1; /* assert! :84 */
    *pt = m << (p->log_q - (p->log_p + p->padding_bits));
    return 0;
}

/* lwe.rs:100-107: plain shift, no rounding, no mask */
uint32_t orc_lwe_decode(const orc_params *p, uint32_t pt) {
    return pt >> (p->log_q - (p->log_p + p->padding_bits));
}

static uint32_t dot_u32(const uint32_t *a, const uint32_t *b, size_t n) {
    uint32_t acc = 0;
    for (size_t i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}

/* lwe.rs:138-160 with the draws hoisted out: ct[0..n) already holds the uniform mask, ct[n] the
 * error sample; b = <s, a> + error + pt */
void orc_encrypt_lwe_from_samples(size_t n, const uint32_t *sk, uint32_t pt, uint32_t *ct) {
    uint32_t error = ct[n];
    uint32_t a_s = dot_u32(sk, ct, n);
    a_s += error;
    a_s += pt;
    ct[n] = a_s;
}

/* lwe.rs:138-160 (encrypt_lwe_zero :117-136 is the pt = 0 case); draw order: error, then mask */
void orc_encrypt_lwe_plaintext(size_t n, double std_dev, const uint32_t *sk, uint32_t pt,
                               orc_rng *r, uint32_t *ct) {
    uint32_t error = orc_sample_gaussian(r, std_dev);
    orc_fill_uniform_u32(r, ct, n);
    ct[n] = error;
    orc_encrypt_lwe_from_samples(n, sk, pt, ct);
}

/* lwe.rs:162-173 */
uint32_t orc_decrypt_lwe(size_t n, const uint32_t *sk, const uint32_t *ct) {
    return ct[n] - dot_u32(sk, ct, n);
}

/* glwe.rs:190-209 with the draws hoisted out: the k mask polynomials of ct already hold the
 * uniform samples, the body polynomial the error samples; body = sum a_i * s_i + error */
void orc_encrypt_glwe_zero_from_samples(const orc_params *p, const uint32_t *sk, uint32_t *ct) {
    size_t n = degree_of(p);
    size_t k = p->glwe_dimension;
    uint32_t *body = ct + k * n;
    uint32_t *a_s = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_poly_dot_product(ct, sk, n, k, n, a_s);
    for (size_t i = 0; i < n; ++i) body[i] += a_s[i];
    free(a_s);
}

/* glwe.rs:190-209; draw order: masks, then errors */
void orc_encrypt_glwe_zero(const orc_params *p, const uint32_t *sk, orc_rng *r, uint32_t *ct) {
    size_t n = degree_of(p);
    size_t k = p->glwe_dimension;
    orc_fill_uniform_u32(r, ct, k * n);
    uint32_t *body = ct + k * n;
    for (size_t i = 0; i < n; ++i) body[i] = orc_sample_gaussian(r, p->glwe_std_dev);
    orc_encrypt_glwe_zero_from_samples(p, sk, ct);
}

/* glwe.rs:211-230 */
void orc_encrypt_glwe_plaintext(const orc_params *p, const uint32_t *pt, const uint32_t *sk,
                                orc_rng *r, uint32_t *ct) {
    size_t n = degree_of(p);
    orc_encrypt_glwe_zero(p, sk, r, ct);
    uint32_t *body = ct + (size_t)p->glwe_dimension * n;
    for (size_t i = 0; i < n; ++i) body[i] += pt[i];
}

/* glwe.rs:245-265 */
void orc_decrypt_glwe_ciphertext(const orc_params *p, const uint32_t *sk, const uint32_t *ct,
                                 uint32_t *pt) {
    size_t n = degree_of(p);
    size_t k = p->glwe_dimension;
    uint32_t *a_s = (uint32_t *)malloc(n * sizeof(uint32_t));
    orc_poly_dot_product(ct, sk, n, k, n, a_s);
    for (size_t i = 0; i < n; ++i) pt[i] = ct[k * n + i] - a_s[i];
    free(a_s);
}

/* ggsw.rs:76-130: row = poly_index*levels + level is a GLWE encryption of zero with
 * message * 2^{log_base*(floor(log_q/log_base) - (level+1))} added to coefficient 0 of
 * polynomial `poly_index` (:96-103); nothing is added when message == 0 */
static void ggsw_rows(const orc_params *p, uint32_t message, const uint32_t *glwe_sk, orc_rng *r,
                      uint32_t *ggsw) {
    size_t n = degree_of(p);
    size_t k1 = p->glwe_dimension + 1;
    const orc_decomposer *d = &p->pbs_decomposer;
    for (size_t i = 0; i < k1; ++i) {
        for (uint32_t level_index = 0; level_index < d->levels; ++level_index) {
            uint32_t *row = ggsw + (i * d->levels + level_index) * k1 * n;
            if (r) orc_encrypt_glwe_zero(p, glwe_sk, r, row);
            else orc_encrypt_glwe_zero_from_samples(p, glwe_sk, row);
            if (message != 0) {
                /* m * beta^{l-(level_index+1)} (ggsw.rs:96-99); orc_gadget_shift is that exponent
                 * in literal mode */
                uint32_t decomposition_factor = message * ((uint32_t)1 << orc_gadget_shift(d, level_index));
                row[i * n + 0] += decomposition_factor;
            }
        }
    }
}

void orc_encrypt_ggsw_plaintext(const orc_params *p, uint32_t message, const uint32_t *glwe_sk,
                                orc_rng *r, uint32_t *ggsw) {
    ggsw_rows(p, message, glwe_sk, r, ggsw);
}

/* the same with every row pre-filled (masks + errors) by the caller */
void orc_encrypt_ggsw_from_samples(const orc_params *p, uint32_t message, const uint32_t *glwe_sk,
                                   uint32_t *ggsw) {
    ggsw_rows(p, message, glwe_sk, NULL, ggsw);
}

/* key_switching.rs:20-60: row s_index*levels + level = LWE_to(0) with
 * s_bit * 2^{log_base*(l - (level+1))} added to the b slot */
static void ksk_rows(const uint32_t *from_sk, size_t from_n, const uint32_t *to_sk, size_t to_n,
                     double to_std_dev, const orc_decomposer *d, orc_rng *r, uint32_t *ksk) {
    for (size_t s_index = 0; s_index < from_n; ++s_index) {
        for (uint32_t level_index = 0; level_index < d->levels; ++level_index) {
            uint32_t factor = (uint32_t)1 << orc_gadget_shift(d, level_index); /* beta^{l-(level+1)} */
            factor *= from_sk[s_index];
            uint32_t *row = ksk + (s_index * d->levels + level_index) * (to_n + 1);
            if (r) orc_encrypt_lwe_plaintext(to_n, to_std_dev, to_sk, 0u, r, row);
            else orc_encrypt_lwe_from_samples(to_n, to_sk, 0u, row);
            row[to_n] += factor;
        }
    }
}

void orc_generate_ksk(const uint32_t *from_sk, size_t from_n, const uint32_t *to_sk, size_t to_n,
                      double to_std_dev, const orc_decomposer *d, orc_rng *r, uint32_t *ksk) {
    ksk_rows(from_sk, from_n, to_sk, to_n, to_std_dev, d, r, ksk);
}

/* the same with every row pre-filled (mask + error in the b slot) by the caller */
void orc_generate_ksk_from_samples(const uint32_t *from_sk, size_t from_n, const uint32_t *to_sk,
                                   size_t to_n, const orc_decomposer *d, uint32_t *ksk) {
    ksk_rows(from_sk, from_n, to_sk, to_n, 0.0, d, NULL, ksk);
}

/* bootstrapping.rs:23-56 */
void orc_bootstrapping_key_gen(const orc_params *p, const uint32_t *lwe_sk,
                               const uint32_t *glwe_sk, orc_rng *r, uint32_t *bsk,
                               uint32_t *ksk) {
    size_t n = degree_of(p);
    size_t ggsw_len = orc_ggsw_words(p);
    for (size_t i = 0; i < p->lwe_dimension; ++i)
        orc_encrypt_ggsw_plaintext(p, lwe_sk[i], glwe_sk, r, bsk + i * ggsw_len);
    /* LweSecretKey::from(&GlweSecretKey) (lwe.rs:62-73) is the row-major flattening, i.e. the
     * glwe_sk buffer itself viewed as k*N words */
    orc_generate_ksk(glwe_sk, n * p->glwe_dimension, lwe_sk, p->lwe_dimension, p->lwe_std_dev,
                     &p->ks_decomposer, r, ksk);
}

/* bootstrapping.rs:23-56 with bsk and ksk pre-filled (masks + errors) by the caller */
void orc_bootstrapping_key_gen_from_samples(const orc_params *p, const uint32_t *lwe_sk,
                                            const uint32_t *glwe_sk, uint32_t *bsk, uint32_t *ksk) {
    size_t n = degree_of(p);
    size_t ggsw_len = orc_ggsw_words(p);
    for (size_t i = 0; i < p->lwe_dimension; ++i)
        orc_encrypt_ggsw_from_samples(p, lwe_sk[i], glwe_sk, bsk + i * ggsw_len);
    orc_generate_ksk_from_samples(glwe_sk, n * p->glwe_dimension, lwe_sk, p->lwe_dimension,
                                  &p->ks_decomposer, ksk);
}

/*
 * selftest.c -- pins the CPU oracle (test infrastructure) against the reference's own asserting
 * tests, restated with deterministic seeds, plus known answers derived from the reference text.
 * Usage: oracle_selftest [--full]   (--full runs the exhaustive 10^8 decomposition loop)
 */
#include "tfhe_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int failures = 0;
#define EXPECT(cond, ...)                                   \
    do {                                                    \
        if (!(cond)) {                                      \
            ++failures;                                     \
            fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
            fprintf(stderr, __VA_ARGS__);                   \
            fprintf(stderr, "\n");                          \
        }                                                   \
    } while (0)

/* decomposer.rs:103-115 `decomposition` */
static void test_decomposition(int full) {
    orc_decomposer d = {4, 7, 32};
    uint32_t legs[32];
    uint32_t limit = full ? 100000000u : 2000000u;
    for (uint32_t i = 0; i < limit; ++i) {
        orc_decompose(&d, i, legs);
        if (orc_recompose(&d, legs) != orc_round_value(&d, i)) {
            EXPECT(0, "decomposition mismatch at %u", i);
            return;
        }
    }
    /* the same identity must hold over the whole u32 range for every dividing base */
    orc_rng r;
    orc_rng_seed(&r, 1);
    orc_decomposer ds[] = {{4, 5, 32}, {4, 6, 32}, {4, 7, 32}, {4, 8, 32}, {8, 2, 32},
                           {8, 3, 32}, {8, 4, 32}, {16, 2, 32}, {2, 16, 32}, {1, 32, 32}};
    for (size_t k = 0; k < sizeof(ds) / sizeof(ds[0]); ++k)
        for (int t = 0; t < 200000; ++t) {
            uint32_t v = orc_rng_next_u32(&r);
            orc_decompose(&ds[k], v, legs);
            if (orc_recompose(&ds[k], legs) != orc_round_value(&ds[k], v)) {
                EXPECT(0, "recompose != round at base %u levels %u v=%u", ds[k].log_base,
                       ds[k].levels, v);
                break;
            }
        }
}

/* known answers derived by hand from decomposer.rs:27-80 */
static void test_decompose_known() {
    uint32_t legs[32];
    orc_decomposer d46 = {4, 6, 32};
    EXPECT(orc_round_value(&d46, 0xABCDEF12u) == 0xABCDEF00u, "round (4,6)");
    orc_decompose(&d46, 0xABCDEF12u, legs);
    int32_t want46[6] = {-5, -4, -3, -2, -1, -1};
    for (int i = 0; i < 6; ++i) EXPECT((int32_t)legs[i] == want46[i], "digit (4,6)[%d]=%d", i, (int32_t)legs[i]);
    /* misaligned base: digits sit at bits 21,14,7; bits 28..31 are dropped */
    orc_decomposer d73 = {7, 3, 32};
    EXPECT(orc_round_value(&d73, 0xABCDEF12u) == 0xABCDF000u, "round (7,3)");
    orc_decompose(&d73, 0xABCDEF12u, legs);
    EXPECT(legs[0] == 0xFFFFFFDEu && legs[1] == 0x38u && legs[2] == 0xFFFFFFE0u,
           "digits (7,3) = %08x %08x %08x", legs[0], legs[1], legs[2]);
    /* a limb equal to B stays B with no carry out */
    orc_decomposer d44 = {4, 8, 32};
    orc_decompose(&d44, 0x000000F8u, legs); /* limbs LSB first: 8 -> -8 carry 1; F+1 = 16 -> stays 16 */
    EXPECT(legs[7] == (uint32_t)-8 && legs[6] == 16u && legs[5] == 0u, "B digit: %d %d %d",
           (int32_t)legs[7], (int32_t)legs[6], (int32_t)legs[5]);
}

/* utils.rs:265-272 `poly_mul_works` + the value of the product */
static void test_poly_mul_works() {
    uint32_t v0[7] = {12, 4, 123, 43, 3, 2, 3};
    uint32_t v1[7] = {12, 232, 5, 3, 2, 4, 2};
    uint32_t a[7], b[7];
    orc_set_poly_mul_mode(0);
    orc_poly_mul(v0, v1, 7, a);
    orc_school_book_negacylic_mul(v0, v1, 7, b);
    EXPECT(memcmp(a, b, sizeof(a)) == 0, "teoplitz != schoolbook");
    uint32_t want[7] = {4294966139u, 2387u, 2353u, 29088u, 10647u, 1354u, 930u};
    EXPECT(memcmp(a, want, sizeof(a)) == 0, "poly_mul known answer");
    /* random sizes, both paths */
    orc_rng r;
    orc_rng_seed(&r, 2);
    for (size_t n = 1; n <= 64; n += 7) {
        uint32_t p0[64], p1[64], x[64], y[64];
        orc_fill_uniform_u32(&r, p0, n);
        orc_fill_uniform_u32(&r, p1, n);
        orc_poly_mul(p0, p1, n, x);
        orc_school_book_negacylic_mul(p0, p1, n, y);
        EXPECT(memcmp(x, y, n * 4) == 0, "teoplitz != schoolbook at n=%zu", n);
    }
    /* teoplitz_works (utils.rs:258-262) layout */
    uint32_t v[5] = {0, 1, 2, 3, 4}, m[25];
    orc_teoplitz(v, 5, m);
    int32_t wantm[25] = {0, -4, -3, -2, -1, 1, 0, -4, -3, -2, 2, 1, 0, -4, -3,
                         3, 2,  1,  0,  -4, 4, 3, 2,  1,  0};
    for (int i = 0; i < 25; ++i) EXPECT((int32_t)m[i] == wantm[i], "teoplitz[%d]", i);
}

/* utils.rs:275-305 `poly_mul_monomial_works`, over all indices including negative ones */
static void test_poly_mul_monomial() {
    orc_rng r;
    orc_rng_seed(&r, 3);
    for (size_t n = 1; n <= 16; n *= 2) {
        uint32_t v0[16], got[16], want[16], mono[16];
        orc_fill_uniform_u32(&r, v0, n);
        for (int64_t idx = -(int64_t)(4 * n); idx <= (int64_t)(4 * n); ++idx) {
            orc_poly_mul_monomial(v0, n, idx, got);
            int64_t m = ((idx % (int64_t)(2 * n)) + (int64_t)(2 * n)) % (int64_t)(2 * n);
            memset(mono, 0, sizeof(mono));
            mono[m % n] = (m / n) ? 0xFFFFFFFFu : 1u;
            orc_school_book_negacylic_mul(v0, mono, n, want);
            EXPECT(memcmp(got, want, n * 4) == 0, "monomial n=%zu idx=%lld", n, (long long)idx);
        }
    }
}

static void test_switch_modulus_known() {
    uint32_t in[7] = {0u, 1u << 21, (1u << 21) + 1, 1u << 22, 0xFFFFFFFFu, 0xFFE00000u, 0xFFDFFFFFu};
    uint32_t want[7] = {0, 1, 1, 1, 0, 0, 1023};
    uint32_t out[7];
    orc_switch_modulus(in, 7, 32, 10, out);
    for (int i = 0; i < 7; ++i) EXPECT(out[i] == want[i], "switch_modulus[%d]=%u", i, out[i]);
}

static void test_test_vectors() {
    orc_params p;
    orc_params_default(&p, 1);
    uint32_t tv[512];
    uint32_t and_truth[4] = {0, 0, 0, 1};
    orc_construct_test_vector_boolean(&p, and_truth, tv);
    /* run lengths 0x320, 1x128, 0x64 */
    int ok = 1;
    for (int i = 0; i < 512; ++i) ok &= tv[i] == (uint32_t)((i >= 320 && i < 448) ? 1 : 0);
    EXPECT(ok, "AND test vector");
    orc_construct_identity_test_vector(&p, tv);
    ok = 1;
    for (int i = 0; i < 512; ++i) {
        uint32_t w = i < 64 ? 0 : i < 192 ? 1 : i < 320 ? 2 : i < 448 ? 3 : 0;
        ok &= tv[i] == w;
    }
    EXPECT(ok, "identity test vector");
}

/* lwe.rs:183-194 */
static void test_encrypt_and_decrypt_lwe() {
    orc_params p;
    orc_params_default(&p, 0);
    orc_rng r;
    orc_rng_seed(&r, 4);
    uint32_t *sk = malloc(p.lwe_dimension * 4), *ct = malloc((p.lwe_dimension + 1) * 4);
    orc_lwe_secret_key_random(&p, &r, sk);
    for (uint32_t m = 0; m < 4; ++m) {
        uint32_t pt;
        orc_lwe_encode(&p, m, &pt);
        orc_encrypt_lwe_plaintext(p.lwe_dimension, p.lwe_std_dev, sk, pt, &r, ct);
        /* decode is a plain shift (no rounding): add half a slot before decoding like a user would */
        uint32_t dec = orc_decrypt_lwe(p.lwe_dimension, sk, ct);
        uint32_t msg = orc_lwe_decode(&p, dec + (1u << (32 - p.log_p - p.padding_bits - 1)));
        EXPECT(msg == m, "lwe roundtrip m=%u got %u", m, msg);
    }
    free(sk);
    free(ct);
}

/* glwe.rs:275-294 (the decoded-message assert) */
static void test_encrypt_and_decrypt_glwe() {
    orc_params p;
    orc_params_default(&p, 1);
    orc_set_poly_mul_mode(1);
    orc_rng r;
    orc_rng_seed(&r, 5);
    size_t n = 512, k = 2;
    uint32_t *sk = malloc(k * n * 4), *ct = malloc((k + 1) * n * 4), *msg = malloc(n * 4),
             *pt = malloc(n * 4), *back = malloc(n * 4);
    orc_glwe_secret_key_random(&p, &r, sk);
    for (size_t i = 0; i < n; ++i) msg[i] = orc_rng_next_u32(&r) & 3;
    orc_glwe_encode_message(&p, msg, n, pt);
    orc_encrypt_glwe_plaintext(&p, pt, sk, &r, ct);
    orc_decrypt_glwe_ciphertext(&p, sk, ct, back);
    int ok = 1;
    uint32_t half = 1u << (32 - p.log_p - p.padding_bits - 1);
    for (size_t i = 0; i < n; ++i) ok &= (orc_lwe_decode(&p, back[i] + half) & 3) == msg[i];
    EXPECT(ok, "glwe roundtrip");
    free(sk); free(ct); free(msg); free(pt); free(back);
}

/* key_switching.rs:118-159 */
static void test_key_switching_works() {
    orc_params p;
    orc_params_default(&p, 1);
    orc_rng r;
    orc_rng_seed(&r, 6);
    size_t from_n = 1024, to_n = p.lwe_dimension;
    uint32_t *from_sk = malloc(from_n * 4), *to_sk = malloc(to_n * 4);
    uint32_t *ct = malloc((from_n + 1) * 4), *out = malloc((to_n + 1) * 4);
    uint32_t *ksk = malloc(from_n * p.ks_decomposer.levels * (to_n + 1) * 4);
    orc_sample_binary(&r, from_sk, from_n);
    orc_sample_binary(&r, to_sk, to_n);
    uint32_t pt;
    orc_lwe_encode(&p, 1, &pt);
    orc_encrypt_lwe_plaintext(from_n, p.lwe_std_dev, from_sk, pt, &r, ct);
    orc_generate_ksk(from_sk, from_n, to_sk, to_n, p.lwe_std_dev, &p.ks_decomposer, &r, ksk);
    orc_key_switch_lwe(ct, from_n, to_n, &p.ks_decomposer, ksk, out);
    uint32_t dec = orc_decrypt_lwe(to_n, to_sk, out);
    uint32_t msg = orc_lwe_decode(&p, dec + (1u << (32 - p.log_p - p.padding_bits - 1))) & 3;
    EXPECT(msg == 1, "key switch preserves message, got %u", msg);
    free(from_sk); free(to_sk); free(ct); free(out); free(ksk);
}

/* bootstrapping.rs:194-230 and boolean.rs:67-101 (cfg(test) params: n = 4) */
static void test_bootstrapping_and_gates() {
    orc_params p;
    orc_params_default(&p, 1);
    orc_set_poly_mul_mode(1);
    orc_rng r;
    orc_rng_seed(&r, 7);
    size_t n = p.lwe_dimension, N = 512, k = 2;
    uint32_t *lwe_sk = malloc(n * 4), *glwe_sk = malloc(k * N * 4);
    uint32_t *bsk = malloc(orc_bsk_words(&p) * 4), *ksk = malloc(orc_ksk_words(&p) * 4);
    orc_lwe_secret_key_random(&p, &r, lwe_sk);
    orc_glwe_secret_key_random(&p, &r, glwe_sk);
    orc_bootstrapping_key_gen(&p, lwe_sk, glwe_sk, &r, bsk, ksk);
    uint32_t tv[512], ct[5], out[5], pt;
    uint32_t half = 1u << (32 - p.log_p - p.padding_bits - 1);

    orc_construct_identity_test_vector(&p, tv);
    for (uint32_t m = 0; m < 4; ++m) {
        orc_lwe_encode(&p, m, &pt);
        orc_encrypt_lwe_plaintext(n, p.lwe_std_dev, lwe_sk, pt, &r, ct);
        EXPECT(orc_bootstrap(&p, ct, bsk, ksk, tv, out, NULL) == 0, "bootstrap rc");
        uint32_t msg = orc_lwe_decode(&p, orc_decrypt_lwe(n, lwe_sk, out) + half) & 3;
        EXPECT(msg == m, "bootstrapping_works m=%u got %u", m, msg);
    }
    uint32_t and_truth[4] = {0, 0, 0, 1}, or_truth[4] = {0, 1, 1, 1}, nand_truth[4] = {1, 1, 1, 0};
    const uint32_t *truths[3] = {and_truth, or_truth, nand_truth};
    for (int g = 0; g < 3; ++g)
        for (uint32_t i = 0; i < 4; ++i) {
            uint32_t lhs = (i >> 1) & 1, rhs = i & 1, ct0[5], ct1[5];
            orc_lwe_encode(&p, lhs, &pt);
            orc_encrypt_lwe_plaintext(n, p.lwe_std_dev, lwe_sk, pt, &r, ct1);
            orc_lwe_encode(&p, rhs, &pt);
            orc_encrypt_lwe_plaintext(n, p.lwe_std_dev, lwe_sk, pt, &r, ct0);
            EXPECT(orc_boolean_gate(&p, truths[g], ct0, ct1, bsk, ksk, out) == 0, "gate rc");
            uint32_t msg = orc_lwe_decode(&p, orc_decrypt_lwe(n, lwe_sk, out) + half) & 3;
            EXPECT(msg == truths[g][i], "gate %d input %u got %u", g, i, msg);
        }
    /* literal (Toeplitz) and schoolbook poly_mul give identical bootstraps */
    uint32_t out2[5];
    orc_lwe_encode(&p, 2, &pt);
    orc_encrypt_lwe_plaintext(n, p.lwe_std_dev, lwe_sk, pt, &r, ct);
    orc_set_poly_mul_mode(1);
    orc_bootstrap(&p, ct, bsk, ksk, tv, out, NULL);
    orc_set_poly_mul_mode(0);
    orc_bootstrap(&p, ct, bsk, ksk, tv, out2, NULL);
    EXPECT(memcmp(out, out2, sizeof(out)) == 0, "literal vs schoolbook bootstrap");
    free(lwe_sk); free(glwe_sk); free(bsk); free(ksk);
}

int main(int argc, char **argv) {
    int full = argc > 1 && strcmp(argv[1], "--full") == 0;
    test_decomposition(full);
    test_decompose_known();
    test_poly_mul_works();
    test_poly_mul_monomial();
    test_switch_modulus_known();
    test_test_vectors();
    test_encrypt_and_decrypt_lwe();
    test_encrypt_and_decrypt_glwe();
    test_key_switching_works();
    test_bootstrapping_and_gates();
    if (failures) {
        fprintf(stderr, "%d FAILURE(S)\n", failures);
        return 1;
    }
    printf("oracle selftest OK%s\n", full ? " (full)" : "");
    return 0;
}

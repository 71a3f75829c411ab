/*
 * tfhe_hip.h -- C ABI of the MI355X-native TFHE programmable-bootstrapping engine.
 *
 * Drop-in boundary for the bootstrapping path of Janmajayamall/tfhe-research (Rust crate `tfhe`
 * v0.1.0).  The reference has no FFI of its own (all modules are private, no extern "C"); each
 * entry point below replaces one crate-internal function and takes that function's data in the
 * reference's own memory layout: contiguous row-major u32 ndarrays viewed as `*const u32 + len`
 * (the reference relies on `as_slice().unwrap()`, e.g. bootstrapping.rs:68, key_switching.rs:73).
 * Reference citations are file:line under /root/reference/src.
 *
 * Conventions
 *   - every word is uint32_t, arithmetic wraps mod 2^32 (reference release-mode semantics);
 *   - every call returns an int status (0 = ok); nothing throws or unwinds across this boundary
 *     (the reference signals failure by panicking: assert!/unwrap);
 *   - plain entry points take HOST pointers, block until the result is in host memory, and may
 *     be called from any thread (one context = one stream; do not share a context between
 *     threads without external locking);
 *   - `_device` entry points take DEVICE pointers, enqueue on the context's HIP stream and
 *     return without synchronising; they are graph-capture safe (no allocation, no sync);
 *   - the two unused secret-key arguments of the reference's bootstrap()/and()/or()
 *     (bootstrapping.rs:61-62, boolean.rs:16,39) carry no information and do not cross the ABI.
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TFHE_OK 0
#define TFHE_ERR_INVALID_PARAMS 1   /* parameter set the reference itself could not run */
#define TFHE_ERR_UNSUPPORTED 2      /* valid, but no kernel is instantiated for this shape */
#define TFHE_ERR_NO_KEY 3           /* bootstrapping key not loaded */
#define TFHE_ERR_HIP 4              /* HIP runtime error (see tfhe_last_error) */
#define TFHE_ERR_INVALID_ARGUMENT 5 /* null pointer / bad count / value the reference assert!s on */
#define TFHE_ERR_NO_DEVICE 6        /* no usable GPU: this library has no CPU fallback */
#define TFHE_ERR_EXACTNESS 7        /* parameter set exceeds the exact-NTT bound */
#define TFHE_ERR_IO 8               /* file missing, truncated, not in this format, or corrupt */

/* decomposer.rs:2-6 DecomposerParams */
typedef struct tfhe_decomposer_params {
    uint32_t log_base;
    uint32_t levels;
    uint32_t log_q;
} tfhe_decomposer_params;

/* lib.rs:23-34 TfheParams (noise parameters omitted: they only matter for key generation).
 * glwe_poly_degree is log2(N), exactly as in the reference (lib.rs:40,60). */
typedef struct tfhe_params {
    uint32_t glwe_dimension;   /* k */
    uint32_t glwe_poly_degree; /* log2 N, supported: 9, 10, 11 */
    uint32_t lwe_dimension;    /* n */
    uint32_t padding_bits;
    uint32_t log_p;
    uint32_t log_q;            /* must be 32 */
    tfhe_decomposer_params ks_decomposer;
    tfhe_decomposer_params pbs_decomposer;
} tfhe_params;

typedef struct tfhe_context tfhe_context;

/* Which decomposer of the parameter set an entry point should use. */
#define TFHE_DECOMPOSER_PBS 0
#define TFHE_DECOMPOSER_KS 1

/* ---- parameters / context --------------------------------------------------------------- */

/* lib.rs:101-123 (cfg_test = 0) or lib.rs:77-99 (cfg_test = 1: n = 4) */
void tfhe_params_default(tfhe_params *params, int cfg_test);
/* 0 if the reference could run this parameter set (no underflow / endless loop / shift >= 32) */
int tfhe_params_validate(const tfhe_params *params);

/* Exact-NTT backends.  All give identical bits; they differ in speed and in the parameter sets
 * they can lift exactly (checked at context creation, TFHE_ERR_EXACTNESS otherwise):
 *   FP64_P49   49-bit prime, fp64 arithmetic, the key word taken whole (one spectrum per key
 *              polynomial): half the multiply-accumulate work of FP64; needs
 *              (k+1)*l * N * B * 2^31 < 2^48.25 and (k+1)*l <= 20 -- small gadget bases, e.g. the
 *              reference's default parameters (N = 512, k = 2, l = 6, log_base = 4)
 *   FP64       42-bit prime, fp64 arithmetic, key split into 16-bit halves; needs
 *              (k+1)*l * N * B * 2^15 < 2^40.9 and log_base <= 9
 *   GOLDILOCKS p = 2^64 - 2^32 + 1, u64 arithmetic; needs (k+1)*l * N * B * 2^32 < 2^62
 *   GOLDILOCKS_SPLIT  the same field with the key split into 16-bit halves; needs
 *              (k+1)*l * N * B * 2^15 < 2^62, which every base the reference can express satisfies
 *   AUTO       the first of FP64_FFT (below), FP64_P49, FP64, GOLDILOCKS, GOLDILOCKS_SPLIT whose bound holds
 *              (env TFHE_HIP_BACKEND=fp64-fft|fp64-p49|fp64|goldilocks|goldilocks-split overrides AUTO). */
#define TFHE_BACKEND_AUTO 0
#define TFHE_BACKEND_GOLDILOCKS 1
#define TFHE_BACKEND_FP64 2
#define TFHE_BACKEND_GOLDILOCKS_SPLIT 3
#define TFHE_BACKEND_FP64_P49 4
/* FP64_FFT: the negacyclic product through a complex FFT in fp64 (N/2 points, two coefficients per element,
 *            key split into 16-bit halves), exact by a proven bound on the rounding error of every output
 *            coefficient (csrc/field_fft.h: (3 n eta + sqrt 2 (R + 1) u) R M^1.5 |x| |y| < 1/4, e.g. 0.013 at N = 1024,
 *            k = 1, l = 3, log_base = 7).  Same bits as the exact-NTT fields. */
#define TFHE_BACKEND_FP64_FFT 5

/* Creates a context bound to HIP device `device`.  Fails with TFHE_ERR_NO_DEVICE when no GPU is
 * present: there is deliberately no CPU path behind this ABI. */
int tfhe_context_create(const tfhe_params *params, int device, tfhe_context **out);
int tfhe_context_create_with_backend(const tfhe_params *params, int device, int backend,
                                     tfhe_context **out);
/* "fp64-fft", "fp64-p49", "fp64-p42", "goldilocks" or "goldilocks-split" */
const char *tfhe_context_backend(const tfhe_context *ctx);
void tfhe_context_destroy(tfhe_context *ctx);
/* Run on an existing hipStream_t, e.g. torch.cuda.current_stream().cuda_stream.  A NULL handle is
 * HIP's default stream (that is what torch hands out unless the caller switched streams).  A new
 * context starts on a private non-blocking stream; tfhe_context_use_own_stream goes back to one.
 * Whatever the stream, everything a call enqueues is ordered on it: the blind rotation of a large batch
 * forks half of its launches onto a second, context-owned stream and joins it again before the call
 * returns (events; tfhe_debug_blind_rotate_plan).  During a stream capture it stays on the one stream. */
int tfhe_context_set_stream(tfhe_context *ctx, void *hip_stream);
int tfhe_context_use_own_stream(tfhe_context *ctx);
int tfhe_context_synchronize(tfhe_context *ctx);
/* Pre-sizes the per-batch workspace so that later _device calls up to `max_batch` never allocate. */
int tfhe_context_reserve(tfhe_context *ctx, size_t max_batch);
const char *tfhe_last_error(const tfhe_context *ctx);
const char *tfhe_status_string(int status);

/* ---- key upload: BootstrappingKey (bootstrapping.rs:18-21) ------------------------------- */
/* bsk: the n GGSW ciphertexts of lwe_sk_ggsw_enc back to back, [n][(k+1)*l][k+1][N]
 *      (ggsw.rs:37-41: row = poly_index*l + level, each row one GLWE, body last);
 * ksk: KeySwitchingKey.data, [k*N*l_ks][n+1] (key_switching.rs:13-15).
 * The device keeps the BSK in the NTT domain (u64, pre-scaled by 1/N) and the KSK as is. */
int tfhe_load_bootstrapping_key(tfhe_context *ctx, const uint32_t *bsk, const uint32_t *ksk);
int tfhe_load_bootstrapping_key_device(tfhe_context *ctx, const uint32_t *bsk, const uint32_t *ksk);

/* Unrolled blind rotation (the crate only sketches it: notes/BMMP Bootstrapping.md:13-25).  Two key
 * bits are consumed per step with three GGSWs per pair,
 *   bsk_bmmp [n/2][3][(k+1)*l][k+1][N]:  GGSW(s_2j s_2j+1), GGSW(s_2j (1 - s_2j+1)), GGSW(s_2j+1 (1 - s_2j)),
 *   acc += sum_{m<3} (X^{e_m} - 1) * external_product(bsk_bmmp[j][m], acc),  e = (a_2j + a_2j+1, a_2j, a_2j+1):
 * half the decompositions and forward transforms for a 1.5x key.  Loading such a key puts the
 * context into this mode (tfhe_bootstrap_batch, tfhe_blind_rotate_batch, the gates ... then run it);
 * loading an ordinary key switches back.  NOT the reference's bootstrap(): same plaintext, different
 * key material, different ciphertext bits (checked against oracle.bootstrap_bmmp instead).  Needs
 * even n, N = 512 and a context in the GOLDILOCKS or FP64_P49 backend -- the fields where its three accumulator
 * sets fit the registers: +10 % / -5 % against the loop there, 1.9-3.4x slower on 50-172 spilled registers in the
 * two-spectra fields, where it is refused (TFHE_ERR_UNSUPPORTED otherwise, with the reason in tfhe_last_error). */
int tfhe_load_bootstrapping_key_bmmp(tfhe_context *ctx, const uint32_t *bsk_bmmp, const uint32_t *ksk);
int tfhe_load_bootstrapping_key_bmmp_device(tfhe_context *ctx, const uint32_t *bsk_bmmp, const uint32_t *ksk);
/* 1 if the loaded key is a BMMP key */
int tfhe_context_uses_bmmp(const tfhe_context *ctx);

/* ---- bootstrap(): bootstrapping.rs:58-120 ------------------------------------------------- */
/* lwe_in  [batch][n+1]   LweCiphertext.data (a_0..a_{n-1}, b)              (lwe.rs:110-115)
 * test_vector_poly [tv_count][N], tv_count = 1 (shared) or batch; un-encoded values < 2^log_p,
 *         as produced by construct_test_from_lut (encoded on the device: glwe.rs:141-151)
 * lwe_out [batch][n+1] */
int tfhe_bootstrap_batch(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                         const uint32_t *test_vector_poly, size_t tv_count, uint32_t *lwe_out);
int tfhe_bootstrap_batch_device(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                                const uint32_t *test_vector_poly, size_t tv_count,
                                uint32_t *lwe_out);

/* Blind rotation only (bootstrapping.rs:67-105): glwe_out [batch][k+1][N] */
int tfhe_blind_rotate_batch(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                            const uint32_t *test_vector_poly, size_t tv_count, uint32_t *glwe_out);
int tfhe_blind_rotate_batch_device(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                                   const uint32_t *test_vector_poly, size_t tv_count,
                                   uint32_t *glwe_out);

/* sample_extract(): bootstrapping.rs:122-156.  glwe [batch][k+1][N] -> lwe_out [batch][k*N+1] */
int tfhe_sample_extract_batch(tfhe_context *ctx, const uint32_t *glwe, size_t batch,
                              size_t sample_index, uint32_t *lwe_out);

/* key_switch_lwe(): key_switching.rs:63-103 with the loaded KSK (from dimension k*N to n).
 * lwe_in [batch][k*N+1] -> lwe_out [batch][n+1] */
int tfhe_key_switch_batch(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                          uint32_t *lwe_out);
int tfhe_key_switch_batch_device(tfhe_context *ctx, const uint32_t *lwe_in, size_t batch,
                                 uint32_t *lwe_out);

/* ---- ggsw.rs ------------------------------------------------------------------------------ */
/* external_product(): ggsw.rs:132-161.  ggsw [ggsw_count][(k+1)*l][k+1][N] with ggsw_count = 1
 * (one GGSW for the whole batch, the blind-rotation shape) or batch; glwe [batch][k+1][N]. */
int tfhe_external_product_batch(tfhe_context *ctx, const uint32_t *ggsw, size_t ggsw_count,
                                const uint32_t *glwe_in, size_t batch, uint32_t *glwe_out);
/* Device-pointer form used by the benchmark: `ggsw_prepared` (NTT domain, 8-byte words,
 * tfhe_prepared_ggsw_words() of them per GGSW) comes from tfhe_prepare_ggsw_device. */
int tfhe_prepared_ggsw_words(const tfhe_context *ctx, size_t *words);
int tfhe_prepare_ggsw_device(tfhe_context *ctx, const uint32_t *ggsw, size_t ggsw_count,
                             void *ggsw_prepared);
int tfhe_external_product_prepared_device(tfhe_context *ctx, const void *ggsw_prepared,
                                          size_t ggsw_count, const uint32_t *glwe_in, size_t batch,
                                          uint32_t *glwe_out);
/* cmux(): ggsw.rs:164-178.  Like the reference, ct1 is CLOBBERED with ct1 - ct0. */
int tfhe_cmux_batch(tfhe_context *ctx, const uint32_t *ggsw, size_t ggsw_count,
                    const uint32_t *ct0, uint32_t *ct1, size_t batch, uint32_t *glwe_out);

/* ---- decomposer.rs / glwe.rs / utils.rs --------------------------------------------------- */
/* SignedDecomposer::decompose: decomposer.rs:42-80.  digits_out [count][levels], MSB first. */
int tfhe_decompose(tfhe_context *ctx, int which_decomposer, const uint32_t *values, size_t count,
                   uint32_t *digits_out);
/* decompose_glwe_ciphertext: glwe.rs:90-108.  glwe [batch][k+1][N] -> [batch][(k+1)*l][N] */
int tfhe_decompose_glwe_batch(tfhe_context *ctx, const uint32_t *glwe, size_t batch,
                              uint32_t *digits_out);
/* switch_modulus: utils.rs:23-33 */
int tfhe_switch_modulus(tfhe_context *ctx, const uint32_t *values, size_t count, uint32_t log_from,
                        uint32_t log_to, uint32_t *out);
/* &GlweCiphertext * &Monomial: glwe.rs:20-34, one monomial index per ciphertext */
int tfhe_glwe_mul_monomial_batch(tfhe_context *ctx, const uint32_t *glwe_in, size_t batch,
                                 const int64_t *monomial_index, uint32_t *glwe_out);

/* ---- lwe.rs ------------------------------------------------------------------------------- */
/* out = c0*ct0 + c1*ct1 over [batch][n+1] words (wrapping): `&a + &b` is (1, 1) (lwe.rs:9-15),
 * `&a * s` is (s, 0) with ct1 = NULL (lwe.rs:17-23), the gate input 2*ct1 + ct0 is (1, 2)
 * (boolean.rs:18).  `words_per_ct` lets the same call serve any LWE dimension. */
int tfhe_lwe_linear_batch(tfhe_context *ctx, uint32_t c0, const uint32_t *ct0, uint32_t c1,
                          const uint32_t *ct1, size_t batch, size_t words_per_ct, uint32_t *out);
int tfhe_lwe_linear_batch_device(tfhe_context *ctx, uint32_t c0, const uint32_t *ct0, uint32_t c1,
                                 const uint32_t *ct1, size_t batch, size_t words_per_ct,
                                 uint32_t *out);

/* ---- test_vector.rs / boolean.rs ---------------------------------------------------------- */
/* construct_test_from_lut: test_vector.rs:38-67 (host-side, no GPU).  out [N] */
int tfhe_construct_test_from_lut(const tfhe_params *params, const uint32_t *lut, size_t lut_len,
                                 uint32_t *out);
/* construct_test_vector_boolean: test_vector.rs:5-20 with the closure given as its truth table
 * truth[(lhs << 1) | rhs] */
int tfhe_construct_test_vector_boolean(const tfhe_params *params, const uint32_t truth[4],
                                       uint32_t *out);
/* and()/or(): boolean.rs:9-53 generalised over the closure: out = bootstrap(2*ct1 + ct0) with the
 * closure's test vector.  AND = {0,0,0,1}, OR = {0,1,1,1}, NAND = {1,1,1,0}, XOR = {0,1,1,0}. */
int tfhe_gate_batch(tfhe_context *ctx, const uint32_t truth[4], const uint32_t *ct0,
                    const uint32_t *ct1, size_t batch, uint32_t *lwe_out);
int tfhe_gate_batch_device(tfhe_context *ctx, const uint32_t truth[4], const uint32_t *ct0,
                           const uint32_t *ct1, size_t batch, uint32_t *lwe_out);
/* Gates of m inputs by the same recipe (notes/Boolean Gates.md:2-11): one PBS of
 * c_in = sum_i 2^i * cts[i] (cts[0] = rightmost / least significant input) with the test vector of
 * lut[x] = truth[x mod 2^m]; truth has 2^m entries < 2^log_p; 1 <= m <= min(log_p, 8) -- three-input
 * gates need a context with log_p >= 3.  cts is an array of m pointers to [batch][n+1]. */
int tfhe_lut_gate_batch(tfhe_context *ctx, const uint32_t *truth, uint32_t inputs,
                        const uint32_t *const *cts, size_t batch, uint32_t *lwe_out);
int tfhe_lut_gate_batch_device(tfhe_context *ctx, const uint32_t *truth, uint32_t inputs,
                               const uint32_t *const *cts, size_t batch, uint32_t *lwe_out);
/* NOT without a bootstrap: (-a, enc(1) - b) with enc(1) = 1 << (32 - log_p - padding_bits) */
int tfhe_lwe_not_batch(tfhe_context *ctx, const uint32_t *ct, size_t batch, uint32_t *lwe_out);
int tfhe_lwe_not_batch_device(tfhe_context *ctx, const uint32_t *ct, size_t batch,
                              uint32_t *lwe_out);

/* ---- encryption side: keygen / encrypt / decrypt --------------------------------
 * The reference draws its randomness from the caller's `rng: &mut R` (uniform masks with
 * sample_uniform_array, errors with sample_gaussian_array); the draws stay with the caller here too:
 * every in/out buffer arrives PRE-FILLED -- mask words hold the uniform samples, the body word /
 * body polynomial holds the error sample(s) -- and the call turns it into the ciphertext by adding
 * the <mask, key> term (and the message).  That makes each call a pure function of its arguments,
 * bit-comparable with the oracle.  Secret keys are host pointers and must be binary (sample_binary,
 * the only kind the reference generates): other values are refused with TFHE_ERR_INVALID_ARGUMENT. */
/* encrypt_glwe_zero glwe.rs:190-209: glwe [count][k+1][N] in/out, glwe_sk [k][N];
 * body += sum_i a_i * s_i.  encrypt_glwe_plaintext (:211-230) = error + message in the body. */
int tfhe_glwe_encrypt_zero_batch(tfhe_context *ctx, const uint32_t *glwe_sk, uint32_t *glwe,
                                 size_t count);
int tfhe_glwe_encrypt_zero_batch_device(tfhe_context *ctx, const uint32_t *glwe_sk, uint32_t *glwe,
                                        size_t count);
/* decrypt_glwe_ciphertext glwe.rs:245-265: plaintext_out [count][N] = body - sum_i a_i * s_i */
int tfhe_glwe_decrypt_batch(tfhe_context *ctx, const uint32_t *glwe_sk, const uint32_t *glwe,
                            size_t count, uint32_t *plaintext_out);
/* encrypt_ggsw_plaintext ggsw.rs:76-130 for `count` messages: ggsw [count][(k+1)l][k+1][N] in/out
 * (every row pre-filled like a GLWE above), messages [count] */
int tfhe_ggsw_encrypt_batch(tfhe_context *ctx, const uint32_t *glwe_sk, const uint32_t *messages,
                            uint32_t *ggsw, size_t count);
int tfhe_ggsw_encrypt_batch_device(tfhe_context *ctx, const uint32_t *glwe_sk,
                                   const uint32_t *messages /* host */, uint32_t *ggsw, size_t count);
/* encrypt_lwe_plaintext lwe.rs:138-160 (encrypt_lwe_zero :117-136 with plaintexts == NULL):
 * lwe [batch][dimension+1] in/out, b += <a, s> + plaintexts[i] (already encoded, lwe.rs:81-90) */
int tfhe_lwe_encrypt_batch(tfhe_context *ctx, const uint32_t *lwe_sk, size_t dimension,
                           const uint32_t *plaintexts, uint32_t *lwe, size_t batch);
int tfhe_lwe_encrypt_batch_device(tfhe_context *ctx, const uint32_t *lwe_sk, size_t dimension,
                                  const uint32_t *plaintexts /* device or NULL */, uint32_t *lwe,
                                  size_t batch);
/* decrypt_lwe lwe.rs:162-173: plaintext_out [batch] = b - <a, s> (still encoded; decode = shift) */
int tfhe_lwe_decrypt_batch(tfhe_context *ctx, const uint32_t *lwe_sk, size_t dimension,
                           const uint32_t *lwe, size_t batch, uint32_t *plaintext_out);
int tfhe_lwe_decrypt_batch_device(tfhe_context *ctx, const uint32_t *lwe_sk, size_t dimension,
                                  const uint32_t *lwe, size_t batch, uint32_t *plaintext_out);
/* KeySwitchingKey::generate_ksk key_switching.rs:20-60: ksk [from_dimension*l_ks][to_dimension+1]
 * in/out pre-filled row by row like an LWE; from_sk [from_dimension], to_sk [to_dimension]; uses
 * the context's ks_decomposer */
int tfhe_generate_ksk(tfhe_context *ctx, const uint32_t *from_sk, size_t from_dimension,
                      const uint32_t *to_sk, size_t to_dimension, uint32_t *ksk);
/* bootstrapping_key_gen bootstrapping.rs:23-56: bsk [n][(k+1)l][k+1][N] and ksk [kN*l_ks][n+1]
 * in/out, pre-filled; lwe_sk [n], glwe_sk [k][N] (the KSK goes from the flattened GLWE key
 * lwe.rs:62-73 to lwe_sk).  `load` != 0 also installs the result as the context's key, without a
 * round trip through the host for the _device form. */
int tfhe_bootstrapping_key_gen(tfhe_context *ctx, const uint32_t *lwe_sk, const uint32_t *glwe_sk,
                               uint32_t *bsk, uint32_t *ksk, int load);
int tfhe_bootstrapping_key_gen_device(tfhe_context *ctx, const uint32_t *lwe_sk,
                                      const uint32_t *glwe_sk, uint32_t *bsk, uint32_t *ksk,
                                      int load);
/* The same for the unrolled blind rotation: bsk_bmmp [n/2][3][(k+1)*l][k+1][N] pre-filled like bsk; the three
 * GGSWs of pair j encrypt s_2j s_2j+1, s_2j (1 - s_2j+1), s_2j+1 (1 - s_2j)
 * (notes/BMMP Bootstrapping.md:22-24).  `load` installs the key (BMMP mode). */
int tfhe_bootstrapping_key_gen_bmmp(tfhe_context *ctx, const uint32_t *lwe_sk, const uint32_t *glwe_sk,
                                    uint32_t *bsk_bmmp, uint32_t *ksk, int load);
int tfhe_bootstrapping_key_gen_bmmp_device(tfhe_context *ctx, const uint32_t *lwe_sk,
                                           const uint32_t *glwe_sk, uint32_t *bsk_bmmp, uint32_t *ksk,
                                           int load);

/* ---- bootstrap order --------------------------------------------------------------
 * 0 (default): the reference's order, PBS then key switch (bootstrapping.rs:58-120): ciphertexts at
 * the boundary have n+1 words.
 * 1: key switch first (notes/TFHE.md:367-400): input [batch][k*N+1] -> key_switch_lwe -> blind
 * rotation -> sample_extract -> output [batch][k*N+1]; linear combinations between bootstraps then
 * amplify only the PBS noise, not PBS + key-switch noise.  tfhe_bootstrap_batch*, tfhe_gate_batch*,
 * tfhe_lut_gate_batch* and tfhe_lwe_not_batch* all use the selected boundary dimension; the same
 * keys serve both orders. */
int tfhe_context_set_bootstrap_order(tfhe_context *ctx, int ks_first);

/* ---- kernel shape of the blind rotation -----------------------------------------------------------
 * The reference's call shape is ONE ciphertext per bootstrap() (bootstrapping.rs:58-65) and one pair per gate
 * (boolean.rs:9-37).  Two kernels compute the same bits:
 *   TFHE_SHAPE_TEAM  k+1 wave groups per sample (two samples per team where they fit): the throughput shape, the only
 *                    one for the prime-field backends and N = 2048;
 *   TFHE_SHAPE_WIDE  2 (k+1) waves per sample, split by digit level and key part: about half the time per CMUX of one
 *                    sample on an idle chip -- the latency shape (FP64_FFT backend, N <= 1024; elsewhere the team runs);
 *   TFHE_SHAPE_AUTO  (default) the launcher picks by batch: wide while the batch leaves most CUs idle.
 * Affects every entry point that rotates (bootstrap, blind_rotate, gates). */
#define TFHE_SHAPE_AUTO 0
#define TFHE_SHAPE_WIDE 1
#define TFHE_SHAPE_TEAM 2
int tfhe_context_set_kernel_shape(tfhe_context *ctx, int shape);

/* ---- decomposer alignment ------------------------------------------------------
 * 0 (default): the reference's literal decomposer -- limbs counted from bit 0 (decomposer.rs:48-70),
 * gadget factors beta^{floor(32/log_base)-(level+1)} (ggsw.rs:98, key_switching.rs:38), bit-exact
 * with the crate; when log_base does not divide 32 the top 32 mod log_base bits are never
 * represented, so such parameter sets do not decrypt (the reference's notes leave beta^l != q as a
 * TODO, notes/TFHE.md:116,407).
 * 1: aligned -- limbs and gadget factors counted down from bit 32 (factor 2^{32-log_base*(level+1)}),
 * which makes e.g. log_base = 7, levels = 3 a working parameter set.  Applies to both decomposers,
 * to the hot path and to keygen; identical bits to mode 0 whenever log_base divides 32.  Keys made
 * in one mode must be used in that mode. */
int tfhe_context_set_decomposer_alignment(tfhe_context *ctx, int aligned);

/* ---- on-disk format for keys and ciphertexts (the reference has none) -----------
 * One array per file, host side only (no GPU needed), little endian:
 *   0   char[8]  magic "TFHEAMD\1"
 *   8   u32      kind (TFHE_FILE_*)          12  u32  flags (bit 0: aligned decomposer)
 *   16  u32[12]  tfhe_params (k, log2 N, n, padding_bits, log_p, log_q, ks{log_base, levels, log_q},
 *                pbs{log_base, levels, log_q})
 *   64  u32      ndims (1..4)                68  u32[4] dims (row-major, unused = 1)   84  u32  0
 *   88  u64      payload words (= product of dims)
 *   96  u64      FNV-1a 64 of the payload bytes
 *   104 u32[]    payload: the array exactly as the ABI takes it (reference layouts)
 * Readers refuse a wrong magic, a size that disagrees with the header, and a checksum mismatch
 * (TFHE_ERR_IO); comparing the stored parameters with the context's is the caller's job
 * (tfhe_file_read_header hands them back). */
#define TFHE_FILE_BSK 1   /* [n][(k+1)l][k+1][N] */
#define TFHE_FILE_KSK 2   /* [kN*l_ks][n+1] */
#define TFHE_FILE_LWE 3   /* [batch][dim+1] */
#define TFHE_FILE_GLWE 4  /* [batch][k+1][N] */
#define TFHE_FILE_GGSW 5  /* [count][(k+1)l][k+1][N] */
#define TFHE_FILE_WORDS 6 /* any other u32 array (test vectors [N], mod-switched masks, ...) */
#define TFHE_FILE_FLAG_ALIGNED 1u
int tfhe_file_write(const char *path, uint32_t kind, const tfhe_params *params, uint32_t flags,
                    const uint32_t *dims, uint32_t ndims, const uint32_t *data);
int tfhe_file_read_header(const char *path, uint32_t *kind, tfhe_params *params, uint32_t *flags,
                          uint32_t dims[4], uint32_t *ndims, uint64_t *words);
int tfhe_file_read(const char *path, uint32_t *data, uint64_t words);

/* ---- multi-GPU pool ----------------------------------------------------------------
 * The reference's shape is ONE BootstrappingKey (bootstrapping.rs:18-21) and many independent bootstrap() calls
 * (bootstrapping.rs:58-65; the gates of boolean.rs:9-53 are one bootstrap each).  A pool holds one context per listed
 * HIP device: the key is uploaded and transformed ONCE (member 0) and the PREPARED key is replicated device to device
 * (peer copies over xGMI, concurrently on the destination members' streams -- not N host uploads and N prepares); a
 * batch is cut into contiguous slices, slice i = [i*q + min(i, r), ...) of q = batch / n (+1 for the first r =
 * batch % n members), and there is no collective anywhere in the data path.  Same bits as a single context: every
 * bootstrap is a pure function of (ciphertext, test vector, keys).
 * A device may be listed more than once (several members on one GPU); that is how a one-GPU box tests the pool.
 * Host-pointer calls block until the results are in host memory and run one host thread per member; a pool must not
 * be used from two threads at once. */
typedef struct tfhe_pool tfhe_pool;
int tfhe_pool_create(const tfhe_params *params, const int *devices, size_t n_devices, int backend,
                     tfhe_pool **out);
void tfhe_pool_destroy(tfhe_pool *pool);
size_t tfhe_pool_size(const tfhe_pool *pool);
/* Borrowed handle of member i (owned by the pool): for the per-context introspection calls (timing, backend name). */
tfhe_context *tfhe_pool_member(tfhe_pool *pool, size_t i);
const char *tfhe_pool_last_error(const tfhe_pool *pool);
/* The slice of a batch member `member` processes: rows [*first, *first + *count). */
int tfhe_pool_shard(const tfhe_pool *pool, size_t batch, size_t member, size_t *first, size_t *count);
/* tfhe_context_set_decomposer_alignment / _set_bootstrap_order / _reserve / _synchronize for every member
 * (reserve sizes each member for its slice of `max_batch`). */
int tfhe_pool_set_decomposer_alignment(tfhe_pool *pool, int aligned);
int tfhe_pool_set_bootstrap_order(tfhe_pool *pool, int ks_first);
int tfhe_pool_set_kernel_shape(tfhe_pool *pool, int shape); /* tfhe_context_set_kernel_shape for every member */
int tfhe_pool_reserve(tfhe_pool *pool, size_t max_batch);
int tfhe_pool_synchronize(tfhe_pool *pool);
/* BootstrappingKey upload, layouts as tfhe_load_bootstrapping_key: host pointers, or (_device) pointers on member
 * 0's device. */
int tfhe_pool_load_bootstrapping_key(tfhe_pool *pool, const uint32_t *bsk, const uint32_t *ksk);
int tfhe_pool_load_bootstrapping_key_device(tfhe_pool *pool, const uint32_t *bsk, const uint32_t *ksk);
/* the BMMP key of tfhe_load_bootstrapping_key_bmmp, replicated the same way */
int tfhe_pool_load_bootstrapping_key_bmmp(tfhe_pool *pool, const uint32_t *bsk_bmmp, const uint32_t *ksk);
/* bootstrapping_key_gen (bootstrapping.rs:23-56) on the pool: arguments as tfhe_bootstrapping_key_gen[_bmmp], the key is
 * generated on member 0's device and, with `load` != 0, installed on EVERY member (prepared once, replicated device to
 * device like an uploaded key).  tfhe_pool_replicate_key re-replicates whatever key member 0 holds -- after a call that
 * installed a key on tfhe_pool_member(pool, 0) alone, e.g. tfhe_bootstrapping_key_gen_device with load. */
int tfhe_pool_bootstrapping_key_gen(tfhe_pool *pool, const uint32_t *lwe_sk, const uint32_t *glwe_sk,
                                    uint32_t *bsk, uint32_t *ksk, int load);
int tfhe_pool_bootstrapping_key_gen_bmmp(tfhe_pool *pool, const uint32_t *lwe_sk, const uint32_t *glwe_sk,
                                         uint32_t *bsk_bmmp, uint32_t *ksk, int load);
int tfhe_pool_replicate_key(tfhe_pool *pool);
/* bootstrap() over a host batch sharded across the members; arguments as tfhe_bootstrap_batch. */
int tfhe_pool_bootstrap_batch(tfhe_pool *pool, const uint32_t *lwe_in, size_t batch,
                              const uint32_t *test_vector_poly, size_t tv_count, uint32_t *lwe_out);
/* and()/or()/... over a host batch sharded across the members; arguments as tfhe_gate_batch. */
int tfhe_pool_gate_batch(tfhe_pool *pool, const uint32_t truth[4], const uint32_t *ct0,
                         const uint32_t *ct1, size_t batch, uint32_t *lwe_out);
/* Device-resident shards: member i bootstraps counts[i] ciphertexts at lwe_in[i] (a pointer on ITS device) with
 * test vector(s) tv[i] (tv_counts[i] = 1 or counts[i]) into lwe_out[i].  Enqueues on every member's stream and
 * returns; tfhe_pool_synchronize waits.  counts[i] = 0 skips a member.  Arguments of all members are checked before
 * anything is enqueued (a failing call has launched nothing).
 * ORDERING CONTRACT: member i's work runs on member i's stream -- its own non-blocking stream unless
 * tfhe_context_set_stream(tfhe_pool_member(pool, i), s) gave it one.  The call does NOT order against whatever stream
 * produced lwe_in[i] / tv[i] or will consume lwe_out[i]: the caller synchronises those producers first (or hands every
 * member the producing stream), and must keep the buffers alive until tfhe_pool_synchronize. */
int tfhe_pool_bootstrap_shards_device(tfhe_pool *pool, const uint32_t *const *lwe_in, const size_t *counts,
                                      const uint32_t *const *test_vector_poly, const size_t *tv_counts,
                                      uint32_t *const *lwe_out);

/* ---- introspection for benchmarks --------------------------------------------------------- */
/* Time of the blind-rotation kernel of the most recent bootstrap/blind_rotate call, measured with
 * HIP events on the context's stream (milliseconds); negative if none was recorded.  Enable with
 * tfhe_context_set_timing(ctx, 1): the _device calls then record events (still no sync). */
int tfhe_context_set_timing(tfhe_context *ctx, int enable);
int tfhe_last_kernel_ms(tfhe_context *ctx, float *blind_rotate_ms, float *key_switch_ms);
/* The same for the bootstrap (or gate) call made `steps_ago` timed calls before the last one (0 = the last; the
 * context keeps the events of the last 64): K steps can be enqueued back to back and read afterwards, with no host
 * synchronisation inside the timed loop. */
int tfhe_kernel_ms_ago(tfhe_context *ctx, unsigned steps_ago, float *blind_rotate_ms, float *key_switch_ms);
/* HBM roofline measured on this device now: a 16-byte-per-lane stream copy of `bytes` (two scratch
 * buffers are allocated and freed inside), `reps` timed launches on the context's stream;
 * *gb_per_s = (bytes read + bytes written) / time.  Synchronises. */
int tfhe_measure_hbm_copy(tfhe_context *ctx, size_t bytes, int reps, double *gb_per_s);

/* Rounding-margin probe of the FP64_FFT backend (test instrumentation, not a reference function): the largest
 * |value - nearest integer| any lane has lifted on the device since the last reset -- the measured counterpart of
 * the proven bound in csrc/field_fft.h.  Only the probe build (libtfhe_hip_probe.so, -DTFHE_FFT_TRACK_ERROR; kept
 * beside the product library, never loaded by it) records it; the product library returns TFHE_ERR_UNSUPPORTED.
 * Synchronises the device. */
int tfhe_debug_fft_margin(tfhe_context *ctx, double *worst, int reset);

/* How tfhe_bootstrap_batch[_device] sends out the blind rotations of a batch of this size (no reference counterpart;
 * bench lines and tests).  A batch larger than what the chip rotates at once (*resident_samples) goes out in groups of
 * *samples_per_group samples; every rotation is cut into *segments launches that walk a slice of the bootstrapping key each
 * (the accumulators wait in the context's workspace in between: tfhe_context_reserve), and with *streams == 2 the two
 * halves of a group alternate on the context's stream and a second stream of its own, which is joined before the call
 * returns -- the caller still orders everything on the one stream it gave (tfhe_context_set_stream). */
int tfhe_debug_blind_rotate_plan(tfhe_context *ctx, size_t batch, size_t *samples_per_group, unsigned *segments,
                                 unsigned *streams, size_t *resident_samples);

/* The kernel shape behind that plan: waves of one workgroup that work on a sample's rotation -- (k+1) x waves per
 * polynomial for the throughput kernel (shared by *samples_per_team samples where two fit), 2 (k+1) for the wide team
 * that batches too small to fill the chip get (one sample per workgroup; the latency shape of the reference's
 * one-ciphertext bootstrap(), bootstrapping.rs:58-65). */
int tfhe_debug_blind_rotate_shape(tfhe_context *ctx, size_t batch, unsigned *waves_per_sample,
                                  unsigned *samples_per_team);

/* Library / build identification */
const char *tfhe_version(void);

#ifdef __cplusplus
}
#endif
#endif

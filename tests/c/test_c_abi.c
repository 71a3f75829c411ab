/* A plain C99 consumer of include/tfhe_hip.h: what a cgo / JNI / Rust-FFI binding sees.  No C++ anywhere on this side.
 * Compiled and linked on the CPU (the header is C, every declared entry point resolves); run on the GPU box:
 *   - the reference's cfg(test) parameter set (lib.rs:77-99), a deterministic pseudo-random key and batch;
 *   - tfhe_bootstrap_batch through one context, tfhe_pool_bootstrap_batch through a pool of two members on device 0:
 *     same words (bootstraps are pure functions of ciphertext, test vector and keys: bootstrapping.rs:58-120);
 *   - a NAND gate stream through the pool against the single context (boolean.rs:9-53 through the closure hook);
 *   - error behaviour: statuses, never a crash (the reference panics);
 *   - the committed golden fixtures (argv[1..]: directories under tests/golden/, each one bsk / ksk / lwe_in / tv / lwe_out
 *     in the library's on-disk format): read with tfhe_file_read, bootstrapped through tfhe_bootstrap_batch and
 *     tfhe_pool_bootstrap_batch, every output word memcmp'ed with lwe_out -- results, not only linkage, with no Python
 *     on the caller's side. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "tfhe_hip.h"

static uint64_t state = 0x746668650000002aull;
static uint32_t next_u32(void) { /* SplitMix64 */
  uint64_t z = (state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}

#define CHECK(expr, want)                                                            \
  do {                                                                               \
    int st_ = (expr);                                                                \
    if (st_ != (want)) {                                                             \
      fprintf(stderr, "%s -> %d (%s), wanted %d\n", #expr, st_, tfhe_status_string(st_), (want)); \
      return 1;                                                                      \
    }                                                                                \
  } while (0)

/* one array of a fixture set: header (kind, params, dims) + words; NULL on any failure */
static uint32_t *read_fixture(const char *dir, const char *name, tfhe_params *params, uint32_t dims[4], uint32_t *ndims,
                              uint32_t *flags) {
  char path[1024];
  uint32_t kind = 0;
  uint64_t words = 0;
  snprintf(path, sizeof path, "%s/%s.tfhe", dir, name);
  if (tfhe_file_read_header(path, &kind, params, flags, dims, ndims, &words) != TFHE_OK) {
    fprintf(stderr, "cannot read the header of %s\n", path);
    return NULL;
  }
  uint32_t *data = malloc(words ? words * 4 : 4);
  if (!data || tfhe_file_read(path, data, words) != TFHE_OK) {
    fprintf(stderr, "cannot read %s\n", path);
    free(data);
    return NULL;
  }
  return data;
}

/* bootstrap the fixture's inputs through a single context and through a two-member pool; compare with its lwe_out */
static int golden_set(const char *dir) {
  tfhe_params p, q;
  uint32_t dims[4], nd = 0, flags = 0, lwe_dims[4], aligned = 0;
  uint32_t *bsk = read_fixture(dir, "bsk", &p, dims, &nd, &aligned);
  uint32_t *ksk = read_fixture(dir, "ksk", &q, dims, &nd, &flags);
  uint32_t *lwe = read_fixture(dir, "lwe_in", &q, lwe_dims, &nd, &flags);
  uint32_t *tv = read_fixture(dir, "tv", &q, dims, &nd, &flags);
  uint32_t *want = read_fixture(dir, "lwe_out", &q, dims, &nd, &flags);
  if (!bsk || !ksk || !lwe || !tv || !want) return 20;
  const size_t batch = lwe_dims[0], n = p.lwe_dimension;
  if (lwe_dims[1] != n + 1 || batch == 0) return 21;
  uint32_t *got = malloc(batch * (n + 1) * 4);
  if (!got) return 2;
  tfhe_context *ctx = NULL;
  CHECK(tfhe_context_create(&p, 0, &ctx), TFHE_OK);
  CHECK(tfhe_context_set_decomposer_alignment(ctx, (aligned & TFHE_FILE_FLAG_ALIGNED) ? 1 : 0), TFHE_OK);
  CHECK(tfhe_load_bootstrapping_key(ctx, bsk, ksk), TFHE_OK);
  memset(got, 0xA5, batch * (n + 1) * 4);
  CHECK(tfhe_bootstrap_batch(ctx, lwe, batch, tv, 1, got), TFHE_OK);
  if (memcmp(got, want, batch * (n + 1) * 4) != 0) {
    fprintf(stderr, "%s: tfhe_bootstrap_batch differs from the fixture's lwe_out\n", dir);
    return 22;
  }
  const int devices[2] = {0, 0};
  tfhe_pool *pool = NULL;
  CHECK(tfhe_pool_create(&p, devices, 2, TFHE_BACKEND_AUTO, &pool), TFHE_OK);
  CHECK(tfhe_pool_set_decomposer_alignment(pool, (aligned & TFHE_FILE_FLAG_ALIGNED) ? 1 : 0), TFHE_OK);
  CHECK(tfhe_pool_load_bootstrapping_key(pool, bsk, ksk), TFHE_OK);
  memset(got, 0x5A, batch * (n + 1) * 4);
  CHECK(tfhe_pool_bootstrap_batch(pool, lwe, batch, tv, 1, got), TFHE_OK);
  if (memcmp(got, want, batch * (n + 1) * 4) != 0) {
    fprintf(stderr, "%s: tfhe_pool_bootstrap_batch differs from the fixture's lwe_out\n", dir);
    return 23;
  }
  printf("golden OK: %s (%zu ciphertexts, backend %s; single context and pool of 2 == lwe_out)\n", dir, batch,
         tfhe_context_backend(ctx));
  tfhe_pool_destroy(pool);
  tfhe_context_destroy(ctx);
  free(bsk); free(ksk); free(lwe); free(tv); free(want); free(got);
  return 0;
}

int main(int argc, char **argv) {
  tfhe_params p;
  tfhe_params_default(&p, 1); /* N = 512, k = 2, n = 4, PBS l = 6 logB = 4, KS l = 5 logB = 4 */
  CHECK(tfhe_params_validate(&p), TFHE_OK);
  const size_t N = (size_t)1 << p.glwe_poly_degree, k = p.glwe_dimension, n = p.lwe_dimension;
  const size_t R = (k + 1) * p.pbs_decomposer.levels;
  const size_t bsk_words = n * R * (k + 1) * N, ksk_words = k * N * p.ks_decomposer.levels * (n + 1);
  const size_t batch = 37; /* odd: the last team of a two-samples-per-team kernel is one sample short */
  uint32_t *bsk = malloc(bsk_words * 4), *ksk = malloc(ksk_words * 4), *lwe = malloc(batch * (n + 1) * 4),
           *lwe2 = malloc(batch * (n + 1) * 4), *tv = malloc(N * 4), *one = malloc(batch * (n + 1) * 4),
           *two = malloc(batch * (n + 1) * 4);
  if (!bsk || !ksk || !lwe || !lwe2 || !tv || !one || !two) return 2;
  for (size_t i = 0; i < bsk_words; ++i) bsk[i] = next_u32();
  for (size_t i = 0; i < ksk_words; ++i) ksk[i] = next_u32();
  for (size_t i = 0; i < batch * (n + 1); ++i) lwe[i] = next_u32(), lwe2[i] = next_u32();
  const uint32_t lut[4] = {0, 1, 2, 3};
  CHECK(tfhe_construct_test_from_lut(&p, lut, 4, tv), TFHE_OK);
  CHECK(tfhe_construct_test_from_lut(&p, lut, 3, tv), TFHE_ERR_INVALID_ARGUMENT); /* assert!(lut.len() == 2^log_p) */

  tfhe_context *ctx = NULL;
  int st = tfhe_context_create(&p, 0, &ctx);
  if (st == TFHE_ERR_NO_DEVICE) { /* CPU-only box: the product has no CPU path, and says so */
    puts("c abi OK (no device: link check only)");
    return 0;
  }
  CHECK(st, TFHE_OK);
  CHECK(tfhe_bootstrap_batch(ctx, lwe, batch, tv, 1, one), TFHE_ERR_NO_KEY);
  CHECK(tfhe_load_bootstrapping_key(ctx, bsk, ksk), TFHE_OK);
  CHECK(tfhe_bootstrap_batch(ctx, lwe, batch, tv, 1, one), TFHE_OK);

  const int devices[2] = {0, 0};
  tfhe_pool *pool = NULL;
  CHECK(tfhe_pool_create(&p, devices, 2, TFHE_BACKEND_AUTO, &pool), TFHE_OK);
  if (tfhe_pool_size(pool) != 2) return 3;
  CHECK(tfhe_pool_bootstrap_batch(pool, lwe, batch, tv, 1, two), TFHE_ERR_NO_KEY);
  CHECK(tfhe_pool_load_bootstrapping_key(pool, bsk, ksk), TFHE_OK);
  CHECK(tfhe_pool_bootstrap_batch(pool, lwe, batch, tv, 1, two), TFHE_OK);
  if (memcmp(one, two, batch * (n + 1) * 4) != 0) {
    fputs("pool and single context disagree\n", stderr);
    return 4;
  }
  size_t first = 0, count = 0;
  CHECK(tfhe_pool_shard(pool, batch, 1, &first, &count), TFHE_OK);
  if (first != 19 || count != 18) return 5; /* 37 = 19 + 18: the first batch % n members get one more */

  const uint32_t nand[4] = {1, 1, 1, 0};
  CHECK(tfhe_gate_batch(ctx, nand, lwe, lwe2, batch, one), TFHE_OK);
  CHECK(tfhe_pool_gate_batch(pool, nand, lwe, lwe2, batch, two), TFHE_OK);
  if (memcmp(one, two, batch * (n + 1) * 4) != 0) {
    fputs("pooled gates and single-context gates disagree\n", stderr);
    return 6;
  }
  CHECK(tfhe_pool_bootstrap_batch(pool, lwe, 0, tv, 1, two), TFHE_ERR_INVALID_ARGUMENT);
  printf("c abi OK: backend %s, %zu bootstraps and %zu NAND gates, pool of 2 == single context\n",
         tfhe_context_backend(ctx), batch, batch);
  tfhe_pool_destroy(pool);
  tfhe_context_destroy(ctx);
  free(bsk); free(ksk); free(lwe); free(lwe2); free(tv); free(one); free(two);
  for (int i = 1; i < argc; ++i) {
    int rc = golden_set(argv[i]);
    if (rc) return rc;
  }
  return 0;
}

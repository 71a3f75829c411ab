// GPU test of the C++ host-side mirror (include/tfhe.hpp).  Written to read like the reference's
// own tests (bootstrapping.rs:194-230, boolean.rs:67-101, key_switching.rs:118-159,
// ggsw.rs:203-280), with the CPU oracle supplying keys/encryption (test infrastructure) and the
// bit-exact expected values.
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "tfhe.hpp"
#include "tfhe_oracle.h"

using namespace tfhe_amd;

static int failures = 0;
#define EXPECT(cond, msg)                                         \
  do {                                                            \
    if (!(cond)) {                                                \
      ++failures;                                                 \
      std::fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, msg); \
    }                                                             \
  } while (0)

static orc_params to_orc(const TfheParams& p) {
  orc_params o;
  orc_params_default(&o, 1);
  o.glwe_dimension = p.glwe_dimension;
  o.glwe_poly_degree = p.glwe_poly_degree;
  o.lwe_dimension = p.lwe_dimension;
  o.padding_bits = p.padding_bits;
  o.log_p = p.log_p;
  o.ks_decomposer = {p.ks_decomposer.log_base, p.ks_decomposer.levels, 32};
  o.pbs_decomposer = {p.pbs_decomposer.log_base, p.pbs_decomposer.levels, 32};
  return o;
}

int main() {
  orc_set_poly_mul_mode(1);
  TfheParams tfhe_params = TfheParams::default_test_params();
  orc_params op = to_orc(tfhe_params);
  const size_t n = tfhe_params.lwe_dimension, N = tfhe_params.degree(), k = tfhe_params.glwe_dimension;
  orc_rng rng;
  orc_rng_seed(&rng, 77);

  // keys (bootstrapping_key_gen, bootstrapping.rs:23-56) in the reference's in-memory shape
  std::vector<uint32_t> lwe_sk(n), glwe_sk(k * N), bsk(orc_bsk_words(&op)), ksk(orc_ksk_words(&op));
  orc_lwe_secret_key_random(&op, &rng, lwe_sk.data());
  orc_glwe_secret_key_random(&op, &rng, glwe_sk.data());
  orc_bootstrapping_key_gen(&op, lwe_sk.data(), glwe_sk.data(), &rng, bsk.data(), ksk.data());
  BootstrappingKey bootstrapping_key;
  const size_t ggsw_words = orc_ggsw_words(&op);
  for (size_t i = 0; i < n; ++i)
    bootstrapping_key.lwe_sk_ggsw_enc.push_back(GgswCiphertext{
        std::vector<uint32_t>(bsk.begin() + i * ggsw_words, bsk.begin() + (i + 1) * ggsw_words)});
  bootstrapping_key.ksk.data = ksk;

  Engine engine(tfhe_params);
  engine.load(bootstrapping_key);

  auto encrypt = [&](uint32_t m) {
    uint32_t pt;
    orc_lwe_encode(&op, m, &pt);
    LweCiphertext ct{std::vector<uint32_t>(n + 1)};
    orc_encrypt_lwe_plaintext(n, op.lwe_std_dev, lwe_sk.data(), pt, &rng, ct.data.data());
    return ct;
  };
  auto decrypt = [&](const LweCiphertext& ct) {
    uint32_t raw = orc_decrypt_lwe(n, lwe_sk.data(), ct.data.data());
    return orc_lwe_decode(&op, raw + (1u << (32 - op.log_p - op.padding_bits - 1))) & 3u;
  };

  // bootstrapping_works
  {
    auto test_vector_poly = construct_identity_test_vector(tfhe_params);
    for (uint32_t m = 0; m < 4; ++m) {
      LweCiphertext lwe_ciphertext = encrypt(m);
      LweCiphertext bootstrapped = bootstrap(engine, lwe_ciphertext, test_vector_poly);
      EXPECT(decrypt(bootstrapped) == m, "bootstrapping_works");
      std::vector<uint32_t> want(n + 1);
      orc_bootstrap(&op, lwe_ciphertext.data.data(), bsk.data(), ksk.data(), test_vector_poly.data(), want.data(), nullptr);
      EXPECT(bootstrapped.data == want, "bootstrap bit-exact vs oracle");
    }
  }
  // boolean_gates_work (and, or, plus nand/xor through the closure hook)
  for (uint32_t i = 0; i < 4; ++i) {
    uint32_t lhs = (i >> 1) & 1, rhs = i & 1;
    LweCiphertext ct1 = encrypt(lhs), ct0 = encrypt(rhs);
    EXPECT(decrypt(and_(engine, ct0, ct1)) == (lhs & rhs), "and");
    EXPECT(decrypt(or_(engine, ct0, ct1)) == (lhs | rhs), "or");
    EXPECT(decrypt(nand(engine, ct0, ct1)) == (1 - (lhs & rhs)), "nand");
    EXPECT(decrypt(xor_(engine, ct0, ct1)) == (lhs ^ rhs), "xor");
    // the gate input is 2*ct1 + ct0 (boolean.rs:18)
    LweCiphertext ct_in = ct1 * 2u + ct0;
    auto tv = construct_test_vector_boolean(tfhe_params, [](uint32_t l, uint32_t r) { return l & r; });
    EXPECT(bootstrap(engine, ct_in, tv).data == and_(engine, ct0, ct1).data, "and == bootstrap(2*ct1+ct0)");
    EXPECT(decrypt(not_(engine, ct0)) == (1 - rhs), "not");
    for (uint32_t s = 0; s < 2; ++s) {
      LweCiphertext sel = encrypt(s);
      EXPECT(decrypt(mux(engine, sel, ct1, ct0)) == (s ? lhs : rhs), "mux");
    }
    EXPECT(lut_gate(engine, {0, 1, 1, 0}, {&ct0, &ct1}).data == xor_(engine, ct0, ct1).data, "lut_gate m=2 == xor");
  }
  // external_product / cmux / key_switch_lwe / sample_extract / decompose: bit-exact vs oracle
  {
    std::vector<uint32_t> ggsw(ggsw_words), c0((k + 1) * N), c1((k + 1) * N), want((k + 1) * N);
    orc_fill_uniform_u32(&rng, ggsw.data(), ggsw.size());
    orc_fill_uniform_u32(&rng, c0.data(), c0.size());
    orc_fill_uniform_u32(&rng, c1.data(), c1.size());
    GgswCiphertext g{ggsw};
    GlweCiphertext ct0{c0}, ct1{c1};
    orc_external_product(&op, ggsw.data(), c0.data(), want.data());
    EXPECT(external_product(engine, g, ct0).data == want, "external_product");
    std::vector<uint32_t> c1_ref = c1;
    orc_cmux(&op, ggsw.data(), c0.data(), c1_ref.data(), want.data());
    GlweCiphertext res = cmux(engine, g, ct0, ct1);
    EXPECT(res.data == want, "cmux result");
    EXPECT(ct1.data == c1_ref, "cmux clobbers ct1 with ct1 - ct0 (ggsw.rs:171)");

    LweCiphertext extracted = sample_extract(engine, res, 0);
    std::vector<uint32_t> want_lwe(k * N + 1);
    orc_sample_extract(&op, res.data.data(), 0, want_lwe.data());
    EXPECT(extracted.data == want_lwe, "sample_extract");
    LweCiphertext switched = key_switch_lwe(engine, extracted);
    std::vector<uint32_t> want_ks(n + 1);
    orc_key_switch_lwe(extracted.data.data(), k * N, n, &op.ks_decomposer, ksk.data(), want_ks.data());
    EXPECT(switched.data == want_ks, "key_switch_lwe");

    GlweCiphertext rot = operator_mul(engine, ct0, Monomial{-5});
    orc_glwe_mul_monomial(c0.data(), k + 1, N, -5, want.data());
    EXPECT(rot.data == want, "glwe * monomial");

    SignedDecomposer dec(engine, TFHE_DECOMPOSER_PBS);
    uint32_t legs[32];
    orc_decompose(&op.pbs_decomposer, 0xABCDEF12u, legs);
    auto got = dec.decompose(0xABCDEF12u);
    EXPECT(got.size() == op.pbs_decomposer.levels && std::memcmp(got.data(), legs, got.size() * 4) == 0, "decompose");
  }
  // The same flows with keys and ciphertexts made by the mirror itself (GPU keygen / encryption,
  // randomness from a C++ engine in place of the reference's thread_rng): bootstrapping_works,
  // key_switching_works (key_switching.rs:118-159), ggsw external_product_works (ggsw.rs:203-240)
  {
    std::mt19937_64 gen(20261003);
    // (the 49-bit prime field: one of the two backends that offer the unrolled blind rotation used below)
    Engine e2(tfhe_params, 0, TFHE_BACKEND_FP64_P49);
    LweSecretKey lwe_secret_key = LweSecretKey::random(n, gen);
    GlweSecretKey glwe_secret_key = GlweSecretKey::random(tfhe_params, gen);
    BootstrappingKey bk = bootstrapping_key_gen(e2, lwe_secret_key, glwe_secret_key, gen);
    EXPECT(bk.lwe_sk_ggsw_enc.size() == n && bk.ksk.data.size() == ksk.size(), "generated key shapes");
    // every GGSW of the generated key decrypts to gadget rows of its key bit: row (i, level) body
    // minus <masks, s> has message*2^{32-logB*(level+1)} on coefficient 0 of column i == k
    const uint32_t logB = tfhe_params.pbs_decomposer.log_base, L = tfhe_params.pbs_decomposer.levels;
    for (size_t bit = 0; bit < n; ++bit) {
      const auto& g = bk.lwe_sk_ggsw_enc[bit].data;
      const size_t row_words = (k + 1) * N;
      GlweCiphertext last_row{std::vector<uint32_t>(g.begin() + (k * L) * row_words, g.begin() + (k * L + 1) * row_words)};
      auto pt = decrypt_glwe_ciphertext(e2, glwe_secret_key, last_row);
      const uint32_t want = lwe_secret_key.data[bit] << (logB * (32 / logB - 1));
      const int32_t err = (int32_t)(pt[0] - want);
      EXPECT(err > -(1 << 12) && err < (1 << 12), "GGSW body row decrypts to s_i * B^{l-1} + small error");
    }
    auto test_vector_poly = construct_identity_test_vector(tfhe_params);
    auto dec_msg = [&](const LweCiphertext& ct) {
      uint32_t raw = decrypt_lwe(e2, lwe_secret_key, ct);
      return decode_plaintext(raw + (1u << (32 - tfhe_params.log_p - tfhe_params.padding_bits - 1)), tfhe_params) & 3u;
    };
    for (uint32_t m = 0; m < 4; ++m) {
      LweCiphertext ct = encrypt_lwe_plaintext(e2, tfhe_params.lwe_std_dev, lwe_secret_key,
                                               encode_message(m, tfhe_params), gen);
      EXPECT(dec_msg(ct) == m, "encrypt/decrypt round trip");
      EXPECT(dec_msg(bootstrap(e2, ct, test_vector_poly)) == m, "bootstrapping_works with generated keys");
      // key_switching_works: encrypt under the flattened GLWE key, switch, decrypt under the LWE key
      LweSecretKey big = lwe_secret_key_from(glwe_secret_key);
      LweCiphertext big_ct = encrypt_lwe_plaintext(e2, tfhe_params.lwe_std_dev, big, encode_message(m, tfhe_params), gen);
      EXPECT(dec_msg(key_switch_lwe(e2, big_ct)) == m, "key_switching_works with generated keys");
    }
    // external_product_works: GGSW(m1) x GLWE(m2) decrypts to m1*m2 (m1 a bit)
    for (uint32_t m1 = 0; m1 < 2; ++m1) {
      GgswCiphertext ggsw = encrypt_ggsw_plaintext(e2, m1, glwe_secret_key, gen);
      std::vector<uint32_t> msg(N);
      for (size_t i = 0; i < N; ++i) msg[i] = encode_message((uint32_t)(i & 3u), tfhe_params);
      GlweCiphertext glwe = encrypt_glwe_plaintext(e2, msg, glwe_secret_key, gen);
      auto pt = decrypt_glwe_ciphertext(e2, glwe_secret_key, external_product(e2, ggsw, glwe));
      bool ok = true;
      for (size_t i = 0; i < N; ++i) {
        const uint32_t got = decode_plaintext(pt[i] + (1u << (32 - tfhe_params.log_p - tfhe_params.padding_bits - 1)), tfhe_params) & 3u;
        ok = ok && got == (m1 ? (uint32_t)(i & 3u) : 0u);
      }
      EXPECT(ok, "external_product_works with generated GGSW/GLWE");
    }
    bool threw = false;
    try {
      LweSecretKey bad{std::vector<uint32_t>(n, 2u)};
      encrypt_lwe_plaintext(e2, 1e-5, bad, 0u, gen);
    } catch (const TfheError&) { threw = true; }
    EXPECT(threw, "non-binary secret key refused");
    // the unrolled blind rotation of notes/BMMP Bootstrapping.md through the mirror: a BMMP key made on
    // the GPU (3 GGSWs per pair of key bits), bootstrapping_works on top of it, then back to the loop
    if (N == 512 && n % 2 == 0) {
      BootstrappingKey bmmp = bootstrapping_key_gen(e2, lwe_secret_key, glwe_secret_key, gen, /*bmmp=*/true);
      EXPECT(bmmp.lwe_sk_ggsw_enc.size() == n / 2 * 3 && e2.uses_bmmp(), "BMMP key shape / mode");
      for (uint32_t m = 0; m < 4; ++m) {
        LweCiphertext ct = encrypt_lwe_plaintext(e2, tfhe_params.lwe_std_dev, lwe_secret_key,
                                                 encode_message(m, tfhe_params), gen);
        EXPECT(dec_msg(bootstrap(e2, ct, test_vector_poly)) == m, "bootstrapping_works with a BMMP key");
      }
      e2.load(bk);
      EXPECT(!e2.uses_bmmp(), "an ordinary key switches back to the reference's loop");
      e2.load(bmmp, true);
      EXPECT(e2.uses_bmmp(), "BMMP key reloaded");
    }
  }
  // error behaviour: the reference panics, the mirror throws
  {
    bool threw = false;
    try {
      construct_test_from_lut(tfhe_params, {0, 1, 2});  // assert!(lut.len() == 2^log_p)
    } catch (const TfheError&) { threw = true; }
    EXPECT(threw, "lut length assert");
    threw = false;
    try {
      TfheParams bad = tfhe_params;
      bad.pbs_decomposer = DecomposerParams(7, 5);  // levels > floor(32/7): endless loop in the reference
      Engine e2(bad);
    } catch (const TfheError&) { threw = true; }
    EXPECT(threw, "invalid decomposer rejected");
  }
  // One BootstrappingKey, many independent bootstrap() calls, several devices: an Engine over a device list
  // (here the one GPU listed twice) shards the batch and must give the single-device engine's bits; so must a
  // stream of NAND gates (boolean.rs:9-53)
  {
    Engine pool(tfhe_params, std::vector<int>{0, 0});
    pool.load(bootstrapping_key);
    EXPECT(pool.devices() == 2, "two members");
    auto test_vector_poly = construct_identity_test_vector(tfhe_params);
    std::vector<LweCiphertext> cts, cts2;
    for (uint32_t i = 0; i < 7; ++i) {
      cts.push_back(encrypt(i & 3u));
      cts2.push_back(encrypt((i >> 1) & 1u));
    }
    auto one = bootstrap_batch(engine, cts, test_vector_poly);
    auto two = bootstrap_batch(pool, cts, test_vector_poly);
    bool same = one.size() == two.size();
    for (size_t i = 0; same && i < one.size(); ++i) same = one[i].data == two[i].data;
    EXPECT(same, "pooled bootstrap_batch == single-device bootstrap_batch");
    for (uint32_t i = 0; i < 7; ++i) EXPECT(decrypt(two[i]) == (i & 3u), "pooled bootstrap decrypts");
    const uint32_t nand_t[4] = {1, 1, 1, 0};
    std::vector<LweCiphertext> bits0, bits1;
    for (uint32_t i = 0; i < 5; ++i) {
      bits0.push_back(encrypt(i & 1u));
      bits1.push_back(encrypt((i >> 1) & 1u));
    }
    auto gates = gate_batch(pool, nand_t, bits0, bits1);
    for (uint32_t i = 0; i < 5; ++i) {
      EXPECT(gates[i].data == nand(engine, bits0[i], bits1[i]).data, "pooled NAND == single-device NAND");
      EXPECT(decrypt(gates[i]) == (1u - ((i & 1u) & ((i >> 1) & 1u))), "pooled NAND decrypts");
    }
  }
  if (failures) {
    std::fprintf(stderr, "%d failure(s)\n", failures);
    return 1;
  }
  std::printf("host mirror OK\n");
  return 0;
}

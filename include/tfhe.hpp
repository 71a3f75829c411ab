// tfhe.hpp -- C++17 host-side mirror of the reference crate's bootstrapping-path interface, over
// the C ABI of include/tfhe_hip.h.
//
// The reference (Janmajayamall/tfhe-research) is Rust; this image has no Rust toolchain, so the
// host side above the C ABI is written in C++ with the reference's names, argument meaning and
// error behaviour (its panics become C++ exceptions thrown HERE, never across the C ABI), so that
// tests read like the reference's own.  A Rust shim with the same shape is shipped as source in
// rust/ and described in INTEGRATION.md.  Citations are file:line under /root/reference/src.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <cmath>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "tfhe_hip.h"

namespace tfhe_amd {

struct TfheError : std::runtime_error {
  int status;
  TfheError(int st, const std::string& what) : std::runtime_error(what), status(st) {}
};

// decomposer.rs:2-16
struct DecomposerParams {
  uint32_t log_base, levels, log_q;
  DecomposerParams(uint32_t log_base_, uint32_t levels_, uint32_t log_q_ = 32)
      : log_base(log_base_), levels(levels_), log_q(log_q_) {}
};

// lib.rs:23-74.  Fields are public here (the reference's are private with no constructor, which
// makes every non-default parameter set unconstructible from outside the crate).
struct TfheParams {
  uint32_t glwe_dimension = 2;
  uint32_t glwe_poly_degree = 9;  // log2 N (lib.rs:40)
  uint32_t lwe_dimension = 722;
  uint32_t padding_bits = 1;
  uint32_t log_p = 2;
  uint32_t log_q = 32;
  DecomposerParams ks_decomposer{4, 5, 32};
  DecomposerParams pbs_decomposer{4, 6, 32};
  double lwe_std_dev = 0.000013071021089943935;    // lib.rs:96,120
  double glwe_std_dev = 0.00000004990272175010415;  // lib.rs:97,121

  static TfheParams default_params() { return TfheParams{}; }  // lib.rs:101-123
  static TfheParams default_test_params() {                    // lib.rs:77-99
    TfheParams p;
    p.lwe_dimension = 4;
    return p;
  }
  size_t degree() const { return size_t(1) << glwe_poly_degree; }             // glwe.rs:124-127
  size_t lwe_dimension_post_pbs() const { return degree() * glwe_dimension; }  // lib.rs:58-66
  size_t ggsw_rows() const { return size_t(glwe_dimension + 1) * pbs_decomposer.levels; }

  tfhe_params c() const {
    tfhe_params p;
    p.glwe_dimension = glwe_dimension;
    p.glwe_poly_degree = glwe_poly_degree;
    p.lwe_dimension = lwe_dimension;
    p.padding_bits = padding_bits;
    p.log_p = log_p;
    p.log_q = log_q;
    p.ks_decomposer = {ks_decomposer.log_base, ks_decomposer.levels, ks_decomposer.log_q};
    p.pbs_decomposer = {pbs_decomposer.log_base, pbs_decomposer.levels, pbs_decomposer.log_q};
    return p;
  }
};

// lwe.rs:110-115: data = (a_0, ..., a_{n-1}, b)
struct LweCiphertext {
  std::vector<uint32_t> data;
};
// lwe.rs:9-15
inline LweCiphertext operator+(const LweCiphertext& a, const LweCiphertext& b) {
  if (a.data.size() != b.data.size()) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "LWE length mismatch");
  LweCiphertext r{a.data};
  for (size_t i = 0; i < r.data.size(); ++i) r.data[i] += b.data[i];
  return r;
}
// lwe.rs:17-23
inline LweCiphertext operator*(const LweCiphertext& a, uint32_t rhs) {
  LweCiphertext r{a.data};
  for (auto& v : r.data) v *= rhs;
  return r;
}

// glwe.rs:185-188: row-major (k+1, N), body last
struct GlweCiphertext {
  std::vector<uint32_t> data;
};
// ggsw.rs:37-41: ((k+1)*l, k+1, N), row = poly_index*l + level
struct GgswCiphertext {
  std::vector<uint32_t> data;
};
// key_switching.rs:13-15: (k*N*l_ks, n+1)
struct KeySwitchingKey {
  std::vector<uint32_t> data;
};
// bootstrapping.rs:18-21
struct BootstrappingKey {
  std::vector<GgswCiphertext> lwe_sk_ggsw_enc;
  KeySwitchingKey ksk;
};
// glwe.rs:16-18
struct Monomial {
  int64_t index;
};

// The GPU engine: owns the device-side (NTT-domain) copies of ONE BootstrappingKey on one or several GPUs.
// With a device list it is a pool (tfhe_pool_*, tfhe_hip.h): the key is uploaded and transformed once and the
// prepared key replicated device to device; bootstrap_batch() and gate_batch() cut their batch into contiguous
// slices, one per device; the single-ciphertext functions run on the first device.  Same bits either way.
class Engine {
 public:
  // backend: TFHE_BACKEND_AUTO, or one of the transforms of tfhe_hip.h (every one returns the same bits;
  // TFHE_ERR_EXACTNESS / TFHE_ERR_UNSUPPORTED if the parameter set is outside the chosen one's bound or shapes)
  explicit Engine(const TfheParams& params, int device = 0, int backend = TFHE_BACKEND_AUTO)
      : Engine(params, std::vector<int>{device}, backend) {}
  // one member per entry of `devices` (HIP device ordinals; an ordinal may repeat)
  Engine(const TfheParams& params, const std::vector<int>& devices, int backend = TFHE_BACKEND_AUTO) : params_(params) {
    tfhe_params cp = params.c();
    tfhe_pool* raw = nullptr;
    int st = tfhe_pool_create(&cp, devices.data(), devices.size(), backend, &raw);
    if (st != TFHE_OK) throw TfheError(st, std::string("tfhe_pool_create: ") + tfhe_status_string(st));
    pool_.reset(raw);
    ctx_ = tfhe_pool_member(raw, 0);
  }
  // "fp64-fft", "fp64-p49", "fp64-p42", "goldilocks" or "goldilocks-split"
  std::string backend() const { return tfhe_context_backend(ctx_); }
  size_t devices() const { return tfhe_pool_size(pool_.get()); }

  // Uploads the key in the reference's own layout (n separate GGSW arrays + the KSK array).  A key of
  // the unrolled blind rotation (notes/BMMP Bootstrapping.md; bootstrapping_key_gen_bmmp below) holds
  // 3 GGSWs per pair of key bits instead: pass bmmp = true; bootstrap() and the gates then use it.
  void load(const BootstrappingKey& bk, bool bmmp = false) {
    const size_t ggsw_words = params_.ggsw_rows() * (params_.glwe_dimension + 1) * params_.degree();
    if (bk.lwe_sk_ggsw_enc.size() != (bmmp ? params_.lwe_dimension / 2 * 3 : params_.lwe_dimension))
      throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "bootstrapping key must hold n (BMMP: 3n/2) GGSW ciphertexts");
    std::vector<uint32_t> flat;
    flat.reserve(ggsw_words * bk.lwe_sk_ggsw_enc.size());
    for (const auto& g : bk.lwe_sk_ggsw_enc) {
      if (g.data.size() != ggsw_words) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "GGSW shape");
      flat.insert(flat.end(), g.data.begin(), g.data.end());
    }
    const size_t ksk_words = params_.lwe_dimension_post_pbs() * params_.ks_decomposer.levels *
                             (size_t(params_.lwe_dimension) + 1);
    if (bk.ksk.data.size() != ksk_words) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "KSK shape");
    check_pool(bmmp ? tfhe_pool_load_bootstrapping_key_bmmp(pool_.get(), flat.data(), bk.ksk.data.data())
                    : tfhe_pool_load_bootstrapping_key(pool_.get(), flat.data(), bk.ksk.data.data()));
  }
  bool uses_bmmp() const { return tfhe_context_uses_bmmp(ctx_) != 0; }

  // Extensions beyond the reference (tfhe_hip.h): the aligned decomposer for bases with
  // beta^l != q, and the key-switch-then-PBS order of notes/TFHE.md:367-400 (ciphertexts of
  // k*N+1 words at the boundary).  Defaults are the reference's behaviour.
  void set_decomposer_alignment(bool aligned) { check_pool(tfhe_pool_set_decomposer_alignment(pool_.get(), aligned)); }
  void set_bootstrap_order(bool ks_first) { check_pool(tfhe_pool_set_bootstrap_order(pool_.get(), ks_first)); }
  // TFHE_SHAPE_AUTO (default: by batch size), TFHE_SHAPE_WIDE (the latency shape: the reference's bootstrap() takes ONE
  // ciphertext, bootstrapping.rs:58-65) or TFHE_SHAPE_TEAM for every blind rotation of the engine; same bits either way
  void set_kernel_shape(int shape) { check_pool(tfhe_pool_set_kernel_shape(pool_.get(), shape)); }

  const TfheParams& params() const { return params_; }
  tfhe_context* raw() const { return ctx_; }        // the first device's context (single-ciphertext calls)
  tfhe_pool* raw_pool() const { return pool_.get(); }  // all devices (batch calls)
  void check(int st) const {
    if (st != TFHE_OK) throw TfheError(st, std::string(tfhe_status_string(st)) + ": " + tfhe_last_error(ctx_));
  }
  void check_pool(int st) const {
    if (st != TFHE_OK) throw TfheError(st, std::string(tfhe_status_string(st)) + ": " + tfhe_pool_last_error(pool_.get()));
  }

 private:
  struct Deleter {
    void operator()(tfhe_pool* p) const { tfhe_pool_destroy(p); }
  };
  TfheParams params_;
  std::unique_ptr<tfhe_pool, Deleter> pool_;
  tfhe_context* ctx_ = nullptr;  // member 0, owned by the pool
};

// test_vector.rs:38-67
inline std::vector<uint32_t> construct_test_from_lut(const TfheParams& p, const std::vector<uint32_t>& lut) {
  std::vector<uint32_t> out(p.degree());
  tfhe_params cp = p.c();
  int st = tfhe_construct_test_from_lut(&cp, lut.data(), lut.size(), out.data());
  if (st != TFHE_OK) throw TfheError(st, "construct_test_from_lut: lut must hold 2^log_p entries (test_vector.rs:41)");
  return out;
}
// test_vector.rs:23-35
inline std::vector<uint32_t> construct_identity_test_vector(const TfheParams& p) {
  std::vector<uint32_t> lut(size_t(1) << p.log_p);
  for (size_t i = 0; i < lut.size(); ++i) lut[i] = uint32_t(i);
  return construct_test_from_lut(p, lut);
}
// test_vector.rs:5-20: f(lhs, rhs)
inline std::vector<uint32_t> construct_test_vector_boolean(const TfheParams& p,
                                                           const std::function<uint32_t(uint32_t, uint32_t)>& f) {
  std::vector<uint32_t> lut(size_t(1) << p.log_p);
  for (size_t i = 0; i < lut.size(); ++i) lut[i] = f(uint32_t(i >> 1) & 1u, uint32_t(i) & 1u);
  return construct_test_from_lut(p, lut);
}

// bootstrap(): bootstrapping.rs:58-120, over a batch that shares one test vector.  The engine
// replaces the reference's `&BootstrappingKey` argument (the key lives on the GPU after load()).
inline std::vector<LweCiphertext> bootstrap_batch(Engine& e, const std::vector<LweCiphertext>& cts,
                                                  const std::vector<uint32_t>& test_vector_poly) {
  const size_t n1 = size_t(e.params().lwe_dimension) + 1;
  std::vector<uint32_t> in(cts.size() * n1), out(cts.size() * n1);
  for (size_t b = 0; b < cts.size(); ++b) {
    if (cts[b].data.size() != n1) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "LWE length");
    std::copy(cts[b].data.begin(), cts[b].data.end(), in.begin() + b * n1);
  }
  // sharded over the engine's devices (one device: the plain single-context path)
  e.check_pool(tfhe_pool_bootstrap_batch(e.raw_pool(), in.data(), cts.size(), test_vector_poly.data(), 1, out.data()));
  std::vector<LweCiphertext> res(cts.size());
  for (size_t b = 0; b < cts.size(); ++b) res[b].data.assign(out.begin() + b * n1, out.begin() + (b + 1) * n1);
  return res;
}
inline LweCiphertext bootstrap(Engine& e, const LweCiphertext& lwe_ciphertext,
                               const std::vector<uint32_t>& test_vector_poly) {
  return bootstrap_batch(e, {lwe_ciphertext}, test_vector_poly)[0];
}

// key_switch_lwe(): key_switching.rs:63-103 (from the post-PBS dimension to n, with the loaded KSK)
inline LweCiphertext key_switch_lwe(Engine& e, const LweCiphertext& ct) {
  LweCiphertext out{std::vector<uint32_t>(size_t(e.params().lwe_dimension) + 1)};
  if (ct.data.size() != e.params().lwe_dimension_post_pbs() + 1) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "LWE length");
  e.check(tfhe_key_switch_batch(e.raw(), ct.data.data(), 1, out.data.data()));
  return out;
}

// external_product(): ggsw.rs:132-161
inline GlweCiphertext external_product(Engine& e, const GgswCiphertext& ggsw, const GlweCiphertext& glwe) {
  GlweCiphertext out{std::vector<uint32_t>(glwe.data.size())};
  e.check(tfhe_external_product_batch(e.raw(), ggsw.data.data(), 1, glwe.data.data(), 1, out.data.data()));
  return out;
}
// cmux(): ggsw.rs:164-178 -- mutates glwe_ciphertext1 (ct1 -= ct0) like the reference
inline GlweCiphertext cmux(Engine& e, const GgswCiphertext& ggsw, const GlweCiphertext& glwe_ciphertext0,
                           GlweCiphertext& glwe_ciphertext1) {
  GlweCiphertext out{std::vector<uint32_t>(glwe_ciphertext0.data.size())};
  e.check(tfhe_cmux_batch(e.raw(), ggsw.data.data(), 1, glwe_ciphertext0.data.data(),
                          glwe_ciphertext1.data.data(), 1, out.data.data()));
  return out;
}
// &GlweCiphertext * &Monomial: glwe.rs:20-34
inline GlweCiphertext operator_mul(Engine& e, const GlweCiphertext& glwe, const Monomial& m) {
  GlweCiphertext out{std::vector<uint32_t>(glwe.data.size())};
  e.check(tfhe_glwe_mul_monomial_batch(e.raw(), glwe.data.data(), 1, &m.index, out.data.data()));
  return out;
}
// sample_extract(): bootstrapping.rs:122-156
inline LweCiphertext sample_extract(Engine& e, const GlweCiphertext& glwe, size_t sample_index) {
  LweCiphertext out{std::vector<uint32_t>(e.params().lwe_dimension_post_pbs() + 1)};
  e.check(tfhe_sample_extract_batch(e.raw(), glwe.data.data(), 1, sample_index, out.data.data()));
  return out;
}

// SignedDecomposer: decomposer.rs:18-96 (decompose runs on the device; round/recompose are trivial)
class SignedDecomposer {
 public:
  SignedDecomposer(Engine& e, int which) : e_(e), which_(which) {}
  std::vector<uint32_t> decompose(uint32_t value) const {  // decomposer.rs:42-80
    const auto& d = which_ == TFHE_DECOMPOSER_PBS ? e_.params().pbs_decomposer : e_.params().ks_decomposer;
    std::vector<uint32_t> out(d.levels);
    e_.check(tfhe_decompose(e_.raw(), which_, &value, 1, out.data()));
    return out;
  }

 private:
  Engine& e_;
  int which_;
};

// and()/or(): boolean.rs:9-53, plus the gates the reference leaves to the closure hook
inline LweCiphertext gate(Engine& e, const uint32_t truth[4], const LweCiphertext& ct0, const LweCiphertext& ct1) {
  LweCiphertext out{std::vector<uint32_t>(ct0.data.size())};
  e.check(tfhe_gate_batch(e.raw(), truth, ct0.data.data(), ct1.data.data(), 1, out.data.data()));
  return out;
}
// a stream of gates with one truth table, sharded over the engine's devices (boolean.rs:9-53 per pair)
inline std::vector<LweCiphertext> gate_batch(Engine& e, const uint32_t truth[4], const std::vector<LweCiphertext>& ct0,
                                             const std::vector<LweCiphertext>& ct1) {
  if (ct0.size() != ct1.size() || ct0.empty()) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "gate_batch: equal, non-zero counts");
  const size_t w = ct0[0].data.size();
  std::vector<uint32_t> a(ct0.size() * w), b(ct0.size() * w), out(ct0.size() * w);
  for (size_t i = 0; i < ct0.size(); ++i) {
    if (ct0[i].data.size() != w || ct1[i].data.size() != w) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "LWE length");
    std::copy(ct0[i].data.begin(), ct0[i].data.end(), a.begin() + i * w);
    std::copy(ct1[i].data.begin(), ct1[i].data.end(), b.begin() + i * w);
  }
  e.check_pool(tfhe_pool_gate_batch(e.raw_pool(), truth, a.data(), b.data(), ct0.size(), out.data()));
  std::vector<LweCiphertext> res(ct0.size());
  for (size_t i = 0; i < ct0.size(); ++i) res[i].data.assign(out.begin() + i * w, out.begin() + (i + 1) * w);
  return res;
}
inline LweCiphertext and_(Engine& e, const LweCiphertext& ct0, const LweCiphertext& ct1) {
  const uint32_t t[4] = {0, 0, 0, 1};
  return gate(e, t, ct0, ct1);
}
inline LweCiphertext or_(Engine& e, const LweCiphertext& ct0, const LweCiphertext& ct1) {
  const uint32_t t[4] = {0, 1, 1, 1};
  return gate(e, t, ct0, ct1);
}
inline LweCiphertext nand(Engine& e, const LweCiphertext& ct0, const LweCiphertext& ct1) {
  const uint32_t t[4] = {1, 1, 1, 0};
  return gate(e, t, ct0, ct1);
}
inline LweCiphertext xor_(Engine& e, const LweCiphertext& ct0, const LweCiphertext& ct1) {
  const uint32_t t[4] = {0, 1, 1, 0};
  return gate(e, t, ct0, ct1);
}
// Gates beyond the reference's two, by notes/Boolean Gates.md:2-11.
// NOT needs no bootstrap: (-a, enc(1) - b)
inline LweCiphertext not_(Engine& e, const LweCiphertext& ct) {
  LweCiphertext out{std::vector<uint32_t>(ct.data.size())};
  e.check(tfhe_lwe_not_batch(e.raw(), ct.data.data(), 1, out.data.data()));
  return out;
}
// gate of m inputs: one PBS of sum_i 2^i * cts[i] (cts[0] = rightmost input), truth has 2^m entries;
// needs params.log_p >= m
inline LweCiphertext lut_gate(Engine& e, const std::vector<uint32_t>& truth, const std::vector<const LweCiphertext*>& cts) {
  if (cts.empty() || truth.size() != (size_t(1) << cts.size())) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "truth table size");
  std::vector<const uint32_t*> ptrs;
  for (const LweCiphertext* c : cts) ptrs.push_back(c->data.data());
  LweCiphertext out{std::vector<uint32_t>(cts[0]->data.size())};
  e.check(tfhe_lut_gate_batch(e.raw(), truth.data(), (uint32_t)cts.size(), ptrs.data(), 1, out.data.data()));
  return out;
}
// sel ? a : b with two bootstraps at any log_p >= 2: AND(sel, a) + AND(NOT sel, b)
inline LweCiphertext mux(Engine& e, const LweCiphertext& sel, const LweCiphertext& a, const LweCiphertext& b) {
  const uint32_t t_and[4] = {0, 0, 0, 1}, t_andnot[4] = {0, 1, 0, 0};  // truth[(lhs << 1) | rhs], lhs = sel
  return gate(e, t_and, a, sel) + gate(e, t_andnot, b, sel);
}

// ------------------------------------------------------------------------------------------------
// Encryption side.  The reference threads `rng: &mut R` through keygen and
// encryption; here `Rng` is any C++ UniformRandomBitGenerator.  The draws happen on the host, in
// the reference's order (masks with sample_uniform_array, errors with sample_gaussian_array), and
// land in the output buffer; the GPU then adds the <mask, key> terms and the messages.
// ------------------------------------------------------------------------------------------------
// utils.rs:36-41, but two-sided: the reference's `frac as u32` saturates negative errors to 0
// (a one-sided error distribution); a negative error wraps mod 2^32 here.
inline uint32_t f64_to_torus_unsigned_representation(double v) {
  double frac = v - std::round(v);
  frac = std::round(frac * 4294967296.0);
  return static_cast<uint32_t>(static_cast<uint64_t>(static_cast<int64_t>(frac)));
}
template <class Rng>
void sample_uniform_slice(Rng& rng, uint32_t* out, size_t len) {  // utils.rs:56-66
  std::uniform_int_distribution<uint32_t> d;
  for (size_t i = 0; i < len; ++i) out[i] = d(rng);
}
template <class Rng>
void sample_gaussian_slice(double std_dev, Rng& rng, uint32_t* out, size_t len) {  // utils.rs:43-54
  std::normal_distribution<double> d(0.0, std_dev);
  for (size_t i = 0; i < len; ++i) out[i] = f64_to_torus_unsigned_representation(d(rng));
}
template <class Rng>
void sample_binary_slice(Rng& rng, uint32_t* out, size_t len) {  // utils.rs:68-93
  std::uniform_int_distribution<unsigned> byte(0, 255);
  unsigned cur = byte(rng), bit = 0;
  for (size_t i = 0; i < len; ++i) {
    out[i] = (cur >> bit) & 1u;
    if (++bit == 8) {
      cur = byte(rng);
      bit = 0;
    }
  }
}

// lwe.rs:47-60
struct LweSecretKey {
  std::vector<uint32_t> data;
  template <class Rng>
  static LweSecretKey random(size_t lwe_dimension, Rng& rng) {
    LweSecretKey sk{std::vector<uint32_t>(lwe_dimension)};
    sample_binary_slice(rng, sk.data.data(), sk.data.size());
    return sk;
  }
};
// glwe.rs:171-182: (k, N) row-major
struct GlweSecretKey {
  std::vector<uint32_t> data;
  template <class Rng>
  static GlweSecretKey random(const TfheParams& p, Rng& rng) {
    GlweSecretKey sk{std::vector<uint32_t>(p.glwe_dimension * p.degree())};
    sample_binary_slice(rng, sk.data.data(), sk.data.size());
    return sk;
  }
};
// LweSecretKey::from(&GlweSecretKey) lwe.rs:62-73: the row-major flattening
inline LweSecretKey lwe_secret_key_from(const GlweSecretKey& sk) { return LweSecretKey{sk.data}; }

// LweCleartext::encode_message lwe.rs:81-92 / LwePlaintext::decode lwe.rs:100-107
inline uint32_t encode_message(uint32_t m, const TfheParams& p) {
  if (m >= (1u << p.log_p)) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "assertion failed: m < 1 << log_p");
  return m << (p.log_q - (p.log_p + p.padding_bits));
}
inline uint32_t decode_plaintext(uint32_t pt, const TfheParams& p) {
  return pt >> (p.log_q - (p.log_p + p.padding_bits));
}

// encrypt_lwe_plaintext lwe.rs:138-160 (draw order: error, then mask); `plaintext` is encoded
template <class Rng>
LweCiphertext encrypt_lwe_plaintext(Engine& e, double std_dev, const LweSecretKey& sk, uint32_t plaintext,
                                    Rng& rng) {
  const size_t n = sk.data.size();
  LweCiphertext ct{std::vector<uint32_t>(n + 1)};
  sample_gaussian_slice(std_dev, rng, &ct.data[n], 1);
  sample_uniform_slice(rng, ct.data.data(), n);
  e.check(tfhe_lwe_encrypt_batch(e.raw(), sk.data.data(), n, &plaintext, ct.data.data(), 1));
  return ct;
}
// decrypt_lwe lwe.rs:162-173 -> encoded plaintext
inline uint32_t decrypt_lwe(Engine& e, const LweSecretKey& sk, const LweCiphertext& ct) {
  uint32_t pt = 0;
  e.check(tfhe_lwe_decrypt_batch(e.raw(), sk.data.data(), sk.data.size(), ct.data.data(), 1, &pt));
  return pt;
}

// encrypt_glwe_zero glwe.rs:190-209 (draw order: masks, then errors)
template <class Rng>
void fill_glwe_samples(const TfheParams& p, Rng& rng, uint32_t* row) {
  const size_t N = p.degree(), k = p.glwe_dimension;
  sample_uniform_slice(rng, row, k * N);
  sample_gaussian_slice(p.glwe_std_dev, rng, row + k * N, N);
}
template <class Rng>
GlweCiphertext encrypt_glwe_zero(Engine& e, const GlweSecretKey& sk, Rng& rng) {
  const TfheParams& p = e.params();
  GlweCiphertext ct{std::vector<uint32_t>((p.glwe_dimension + 1) * p.degree())};
  fill_glwe_samples(p, rng, ct.data.data());
  e.check(tfhe_glwe_encrypt_zero_batch(e.raw(), sk.data.data(), ct.data.data(), 1));
  return ct;
}
// encrypt_glwe_plaintext glwe.rs:211-230: message polynomial (encoded) added to the body
template <class Rng>
GlweCiphertext encrypt_glwe_plaintext(Engine& e, const std::vector<uint32_t>& plaintext, const GlweSecretKey& sk,
                                      Rng& rng) {
  const TfheParams& p = e.params();
  if (plaintext.size() != p.degree()) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "plaintext shape");
  GlweCiphertext ct = encrypt_glwe_zero(e, sk, rng);
  uint32_t* body = ct.data.data() + p.glwe_dimension * p.degree();
  for (size_t i = 0; i < p.degree(); ++i) body[i] += plaintext[i];
  return ct;
}
// decrypt_glwe_ciphertext glwe.rs:245-265 -> encoded plaintext polynomial
inline std::vector<uint32_t> decrypt_glwe_ciphertext(Engine& e, const GlweSecretKey& sk, const GlweCiphertext& ct) {
  std::vector<uint32_t> pt(e.params().degree());
  e.check(tfhe_glwe_decrypt_batch(e.raw(), sk.data.data(), ct.data.data(), 1, pt.data()));
  return pt;
}
// encrypt_ggsw_plaintext ggsw.rs:76-130
template <class Rng>
GgswCiphertext encrypt_ggsw_plaintext(Engine& e, uint32_t message, const GlweSecretKey& sk, Rng& rng) {
  const TfheParams& p = e.params();
  const size_t row_words = (p.glwe_dimension + 1) * p.degree();
  GgswCiphertext g{std::vector<uint32_t>(p.ggsw_rows() * row_words)};
  for (size_t r = 0; r < p.ggsw_rows(); ++r) fill_glwe_samples(p, rng, g.data.data() + r * row_words);
  e.check(tfhe_ggsw_encrypt_batch(e.raw(), sk.data.data(), &message, g.data.data(), 1));
  return g;
}
// KeySwitchingKey::generate_ksk key_switching.rs:20-60 (the engine's ks_decomposer)
template <class Rng>
KeySwitchingKey generate_ksk(Engine& e, const LweSecretKey& from_lwe_sk, const LweSecretKey& to_lwe_sk,
                             double to_std_dev, Rng& rng) {
  const size_t to_n = to_lwe_sk.data.size();
  const size_t rows = from_lwe_sk.data.size() * e.params().ks_decomposer.levels;
  KeySwitchingKey ksk{std::vector<uint32_t>(rows * (to_n + 1))};
  for (size_t r = 0; r < rows; ++r) {  // encrypt_lwe_zero lwe.rs:117-136: error, then mask
    uint32_t* row = ksk.data.data() + r * (to_n + 1);
    sample_gaussian_slice(to_std_dev, rng, row + to_n, 1);
    sample_uniform_slice(rng, row, to_n);
  }
  e.check(tfhe_generate_ksk(e.raw(), from_lwe_sk.data.data(), from_lwe_sk.data.size(), to_lwe_sk.data.data(),
                            to_n, ksk.data.data()));
  return ksk;
}
// bootstrapping_key_gen bootstrapping.rs:23-56; the generated key is also installed in the engine.
// bmmp = true makes the key of the unrolled blind rotation instead (notes/BMMP Bootstrapping.md:22-24:
// GGSW(s s'), GGSW(s (1-s')), GGSW(s' (1-s)) per pair of key bits; N = 512, even n).
template <class Rng>
BootstrappingKey bootstrapping_key_gen(Engine& e, const LweSecretKey& lwe_secret_key,
                                       const GlweSecretKey& glwe_secret_key, Rng& rng, bool bmmp = false) {
  const TfheParams& p = e.params();
  const size_t n = p.lwe_dimension, row_words = (p.glwe_dimension + 1) * p.degree();
  const size_t ggsw_words = p.ggsw_rows() * row_words;
  const size_t ggsws = bmmp ? n / 2 * 3 : n;
  if (lwe_secret_key.data.size() != n) throw TfheError(TFHE_ERR_INVALID_ARGUMENT, "lwe secret key shape");
  std::vector<uint32_t> bsk(ggsws * ggsw_words);
  for (size_t r = 0; r < ggsws * p.ggsw_rows(); ++r) fill_glwe_samples(p, rng, bsk.data() + r * row_words);
  const size_t ks_rows = p.lwe_dimension_post_pbs() * p.ks_decomposer.levels;
  KeySwitchingKey ksk{std::vector<uint32_t>(ks_rows * (n + 1))};
  for (size_t r = 0; r < ks_rows; ++r) {
    uint32_t* row = ksk.data.data() + r * (n + 1);
    sample_gaussian_slice(p.lwe_std_dev, rng, row + n, 1);
    sample_uniform_slice(rng, row, n);
  }
  // (the pool entry point: generated on the first device, installed on EVERY device of the engine)
  e.check_pool((bmmp ? tfhe_pool_bootstrapping_key_gen_bmmp : tfhe_pool_bootstrapping_key_gen)(
      e.raw_pool(), lwe_secret_key.data.data(), glwe_secret_key.data.data(), bsk.data(), ksk.data.data(), 1));
  BootstrappingKey bk;
  bk.lwe_sk_ggsw_enc.reserve(ggsws);
  for (size_t i = 0; i < ggsws; ++i)
    bk.lwe_sk_ggsw_enc.push_back(
        GgswCiphertext{std::vector<uint32_t>(bsk.begin() + i * ggsw_words, bsk.begin() + (i + 1) * ggsw_words)});
  bk.ksk = std::move(ksk);
  return bk;
}

}  // namespace tfhe_amd

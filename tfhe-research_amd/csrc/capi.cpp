// capi.cpp -- implementation of the C ABI declared in include/tfhe_hip.h.
//
// Host-side only logic: parameter validation (mirrors what makes the reference panic), device
// memory management, key preparation and kernel sequencing.  No CPU implementation of the hot
// path lives here: without a GPU every compute entry point fails with TFHE_ERR_NO_DEVICE.
#include "context.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <vector>

namespace {

using tfhe::host::fail;
using tfhe::host::hip_fail;

// Bit just above the most significant limb: gadget factor of level i is 2^{top - log_base*(i+1)} and
// the lowest kept limb starts at top - log_base*levels.  Literal (reference, decomposer.rs:48-70 /
// ggsw.rs:98): limbs are counted from bit 0, so top = log_base*floor(32/log_base).  Aligned: 32.
u32 gadget_top(const tfhe_context* ctx, u32 log_base) {
  return ctx->aligned ? 32u : log_base * (32u / log_base);
}

// one workgroup per sample: the grid's x dimension is a 32-bit count
constexpr size_t kMaxBatch = 0x7FFFFFFFull;

int decomposer_validate(const tfhe_decomposer_params& d) {
  if (d.log_q != 32) return 1;                        // the reference is hard-typed to u32
  if (d.log_base == 0 || d.log_base >= 32) return 2;  // 1 << (log_base - 1), 1 << log_base
  if (d.levels == 0) return 3;
  if (d.log_base * d.levels > d.log_q) return 4;      // usize underflow, decomposer.rs:28
  if (d.levels > d.log_q / d.log_base) return 5;      // truncation loop never ends, :74-77
  return 0;
}

template <typename T>
int ensure(tfhe_context* ctx, T** ptr, size_t* have, size_t want_elems) {
  if (*have >= want_elems && *ptr) return TFHE_OK;
  if (*ptr) HIP_TRY(ctx, hipFree(*ptr));
  *ptr = nullptr;
  *have = 0;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(ptr), want_elems * sizeof(T)));
  *have = want_elems;
  return TFHE_OK;
}

int reserve(tfhe_context* ctx, size_t batch) {
  if (batch <= ctx->ws_batch) return TFHE_OK;
  // every LWE buffer can hold either boundary dimension (n for the reference's PBS-then-KS order,
  // k*N for KS-then-PBS), so switching the order never reallocates
  const size_t n1 = std::max((size_t)ctx->params.lwe_dimension, (size_t)ctx->big_n) + 1;
  const size_t glwe = (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  u32** ptrs[] = {&ctx->d_lwe_in, &ctx->d_lwe_in2, &ctx->d_lwe_big, &ctx->d_lwe_out, &ctx->d_lwe_ks,
                  &ctx->d_glwe_a, &ctx->d_glwe_b,   &ctx->d_glwe_c,  &ctx->d_tv};
  const size_t sizes[] = {batch * n1, batch * n1, batch * n1, batch * n1, batch * n1,
                          batch * glwe, batch * glwe, batch * glwe, batch * ctx->N};
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  // the old workspace is gone from here on: a failed allocation below must not leave ws_batch
  // claiming buffers that are null (a later, smaller call would launch kernels on them)
  ctx->ws_batch = 0;
  for (int i = 0; i < 9; ++i) {
    if (*ptrs[i]) {
      hipError_t e = hipFree(*ptrs[i]);
      *ptrs[i] = nullptr;
      if (e != hipSuccess) return hip_fail(ctx, e, "hipFree(workspace)");
    }
  }
  for (int i = 0; i < 9; ++i)
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(ptrs[i]), sizes[i] * sizeof(u32)));
  ctx->ws_batch = batch;
  return TFHE_OK;
}

int ensure_misc(tfhe_context* ctx, size_t bytes) {
  if (bytes <= ctx->misc_bytes) return TFHE_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->misc_bytes = 0;
  if (ctx->d_misc) {
    hipError_t e = hipFree(ctx->d_misc);
    ctx->d_misc = nullptr;
    if (e != hipSuccess) return hip_fail(ctx, e, "hipFree(misc)");
  }
  HIP_TRY(ctx, hipMalloc(&ctx->d_misc, bytes));
  ctx->misc_bytes = bytes;
  return TFHE_OK;
}

int check_ctx(tfhe_context* ctx) {
  if (!ctx) return TFHE_ERR_INVALID_ARGUMENT;
  hipError_t e = hipSetDevice(ctx->device);
  if (e != hipSuccess) return hip_fail(ctx, e, "hipSetDevice");
  return TFHE_OK;
}

// construct_test_from_lut: test_vector.rs:38-67
int test_from_lut(const tfhe_params* p, const u32* lut, size_t lut_len, u32* out) {
  const u32 plaintext_modulus = 1u << p->log_p;
  if (lut_len != plaintext_modulus) return TFHE_ERR_INVALID_ARGUMENT;  // assert! :41
  const size_t n = (size_t)1 << p->glwe_poly_degree;
  const size_t repetition = n / ((size_t)1 << p->log_p);
  std::vector<u32> tv;
  tv.reserve(repetition * lut_len);
  for (size_t v = 0; v < lut_len; ++v)
    for (size_t r = 0; r < repetition; ++r) tv.push_back(lut[v]);
  for (size_t i = 0; i < repetition / 2; ++i)
    if (tv[i] != 0) tv[i] = plaintext_modulus - tv[i];
  const size_t mid = repetition / 2, len = tv.size();
  for (size_t i = 0; i < len; ++i) out[i] = tv[(i + mid) % len];  // rotate_left(mid)
  return TFHE_OK;
}

int check_tv_host(tfhe_context* ctx, const u32* tv, size_t words) {
  // glwe.rs:144 assert!(*m < 1 << log_p)
  if (ctx->params.log_p >= 32) return TFHE_OK;
  const u32 lim = 1u << ctx->params.log_p;
  for (size_t i = 0; i < words; ++i)
    if (tv[i] >= lim)
      return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "test vector value >= 2^log_p (glwe.rs:144)");
  return TFHE_OK;
}

// Enqueue the whole PBS on device buffers: blind rotation (+ fused sample extract), key switch.
// words of one ciphertext at the bootstrap boundary: n+1 in the reference's order (PBS then KS,
// bootstrapping.rs:58-120), k*N+1 when the key switch comes first
using tfhe::host::io_words;

// the blind rotation the loaded key calls for: bootstrapping.rs:79-105, or the unrolled loop of
// notes/BMMP Bootstrapping.md with a BMMP key
hipError_t enqueue_blind_rotate(tfhe_context* ctx, const u32* lwe_in, size_t batch, const u32* tv,
                                size_t tv_count, u32* glwe_out, u32* lwe_extracted) {
  if (ctx->bmmp)
    return launch::blind_rotate_bmmp(ctx->stream, ctx->field, ctx->pbs, ctx->d_tw, lwe_in, batch, tv,
                                     tv_count == 1 ? 0 : ctx->N, ctx->d_bsk, glwe_out, lwe_extracted);
  // accumulators between the launches of a segmented rotation: the caller's output, or the workspace (sized by reserve)
  u32* state = glwe_out ? glwe_out : (batch <= ctx->ws_batch ? ctx->d_glwe_c : nullptr);
  return launch::blind_rotate(ctx->stream, ctx->field, ctx->pbs, ctx->d_tw, lwe_in, batch, tv, tv_count == 1 ? 0 : ctx->N,
                              ctx->d_bsk, glwe_out, lwe_extracted, state, &ctx->side, ctx->shape);
}

// d_lwe_big: [batch][k*N+1] scratch of the reference order (unused when the key switch comes first)
int enqueue_bootstrap(tfhe_context* ctx, const u32* d_lwe_in, size_t batch, const u32* d_tv,
                      size_t tv_count, u32* d_lwe_big, u32* d_lwe_out) {
  const u32* br_in = d_lwe_in;
  u32* br_out = d_lwe_big;
  if (ctx->timing) {  // next slot of the ring
    ctx->ev_slot = (ctx->ev_slot + 1) % tfhe_context::kTimingSlots;
    ctx->ev = ctx->ev_ring[ctx->ev_slot];
    ++ctx->timed_bootstraps;
  }
  if (ctx->ks_first) {  // notes/TFHE.md:367-400: key switch k*N -> n, then PBS back to k*N
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    HIP_TRY(ctx, launch::key_switch(ctx->stream, ctx->ks, ctx->big_n, ctx->params.lwe_dimension,
                                    d_lwe_in, batch, ctx->d_ksk, ctx->d_lwe_ks));
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    br_in = ctx->d_lwe_ks;
    br_out = d_lwe_out;
  }
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  HIP_TRY(ctx, enqueue_blind_rotate(ctx, br_in, batch, d_tv, tv_count, nullptr, br_out));
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
  if (!ctx->ks_first) {
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
    HIP_TRY(ctx, launch::key_switch(ctx->stream, ctx->ks, ctx->big_n, ctx->params.lwe_dimension,
                                    d_lwe_big, batch, ctx->d_ksk, d_lwe_out));
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
  }
  if (ctx->timing) ctx->ev_valid_br = ctx->ev_valid_ks = true;
  return TFHE_OK;
}

int check_batch_args(tfhe_context* ctx, const void* a, const void* b, const void* c, size_t batch,
                     size_t tv_count) {
  if (!a || !b || !c) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if (batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "empty batch");
  if (batch > kMaxBatch) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "batch exceeds 2^31 - 1 (one workgroup per sample)");
  if (tv_count != 1 && tv_count != batch)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "tv_count must be 1 or batch");
  return TFHE_OK;
}

size_t ggsw_words(const tfhe_context* ctx) {
  return (size_t)ctx->R * (ctx->params.glwe_dimension + 1) * ctx->N;
}

// log2 of the worst-case |integer convolution value| one inverse transform has to lift:
// R * N * max|digit| (= B, decomposer.rs:54-63) * max|key operand|
double convolution_bits(const tfhe_params* p, double key_bits) {
  return std::log2((double)(p->glwe_dimension + 1) * p->pbs_decomposer.levels) + p->glwe_poly_degree +
         p->pbs_decomposer.log_base + key_bits;
}

template <class F>
hipError_t upload_twiddles(tfhe_context* ctx) {
  // (the complex transform has N/2 points: F::kLogShrink = 1)
  std::vector<typename F::elem> tw(ntt_twiddle_words((int)(ctx->N >> F::kLogShrink)));
  F::fill_twiddles((int)ctx->params.glwe_poly_degree - F::kLogShrink, tw.data());
  hipError_t e = hipMalloc(&ctx->d_tw, tw.size() * sizeof(typename F::elem));
  if (e != hipSuccess) return e;
  return hipMemcpy(ctx->d_tw, tw.data(), tw.size() * sizeof(typename F::elem), hipMemcpyHostToDevice);
}

}  // namespace

namespace tfhe {
namespace host {

size_t io_words(const tfhe_context* ctx) {
  return (ctx->ks_first ? (size_t)ctx->big_n : (size_t)ctx->params.lwe_dimension) + 1;
}

int adopt_prepared_key(tfhe_context* dst, const tfhe_context* src) {
  if (!dst || !src || !src->have_key) return fail(dst, TFHE_ERR_NO_KEY, "source context holds no key");
  if (std::memcmp(&dst->params, &src->params, sizeof(tfhe_params)) != 0 || dst->field != src->field ||
      dst->aligned != src->aligned)
    return fail(dst, TFHE_ERR_INVALID_PARAMS, "pool members must share parameters, backend and decomposer alignment");
  // from here on dst has NO key until the copies are enqueued: a failure half way must not leave it answering under the
  // key it held before
  dst->have_key = false;
  HIP_TRY(dst, hipSetDevice(dst->device));
  const size_t ggsws = src->bsk_ggsws;
  const size_t bsk_bytes = ggsws * src->R * (src->params.glwe_dimension + 1) * (size_t)src->parts * src->N * sizeof(u64);
  const size_t ksk_bytes = (size_t)src->big_n * src->ks.levels * ((size_t)src->params.lwe_dimension + 1) * sizeof(u32);
  if (dst->d_bsk && dst->bsk_ggsws != ggsws) {
    HIP_TRY(dst, hipStreamSynchronize(dst->stream));
    hipError_t e = hipFree(dst->d_bsk);
    dst->d_bsk = nullptr;
    if (e != hipSuccess) return hip_fail(dst, e, "hipFree(bsk)");
  }
  if (!dst->d_bsk) {
    HIP_TRY(dst, hipMalloc(&dst->d_bsk, bsk_bytes));
    dst->bsk_ggsws = ggsws;
  }
  if (!dst->d_ksk) HIP_TRY(dst, hipMalloc(reinterpret_cast<void**>(&dst->d_ksk), ksk_bytes));
  if (dst->device == src->device) {
    HIP_TRY(dst, hipMemcpyAsync(dst->d_bsk, src->d_bsk, bsk_bytes, hipMemcpyDeviceToDevice, dst->stream));
    HIP_TRY(dst, hipMemcpyAsync(dst->d_ksk, src->d_ksk, ksk_bytes, hipMemcpyDeviceToDevice, dst->stream));
  } else {
    // direct xGMI copies where the link allows peer access; hipMemcpyPeerAsync stages through the host otherwise
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, dst->device, src->device) == hipSuccess && can) {
      hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return hip_fail(dst, e, "hipDeviceEnablePeerAccess");
      (void)hipGetLastError();
    }
    HIP_TRY(dst, hipMemcpyPeerAsync(dst->d_bsk, dst->device, src->d_bsk, src->device, bsk_bytes, dst->stream));
    HIP_TRY(dst, hipMemcpyPeerAsync(dst->d_ksk, dst->device, src->d_ksk, src->device, ksk_bytes, dst->stream));
  }
  dst->have_key = true;
  dst->bmmp = src->bmmp;
  return TFHE_OK;
}

}  // namespace host
}  // namespace tfhe

extern "C" {

// The shipped library's string carries no '[': a dev build (TFHE_DEV_BUILD: the only kind that may compile a WRONG-BITS
// timing probe in, csrc/dev_switches.h), the rounding-margin probe build and one-shape / one-field builds all say so.
const char* tfhe_version(void) {
  return "tfhe-research_amd 0.5 (gfx950; exact backends: fp64-fft, fp64-p49, fp64-p42, goldilocks, goldilocks-split)"
#if defined(TFHE_DEV_BUILD)
         " [DEV BUILD: not for use"
#if TFHE_PROBE_HOT_KEY
         "; WRONG BITS: TFHE_PROBE_HOT_KEY"
#endif
#if TFHE_PROBE_NO_EXCHANGE_READS
         "; WRONG BITS: TFHE_PROBE_NO_EXCHANGE_READS"
#endif
#if TFHE_PROBE_NO_TRANSPOSE
         "; WRONG BITS: TFHE_PROBE_NO_TRANSPOSE"
#endif
#if TFHE_PROBE_NO_TEAM_SYNC
         "; WRONG BITS: TFHE_PROBE_NO_TEAM_SYNC"
#endif
         "]"
#endif
#if defined(TFHE_FFT_TRACK_ERROR)
         " [probe build: fp64-fft rounding margin tracked]"
#endif
#if defined(TFHE_DEV_CFG2_ONLY) || defined(TFHE_DEV_FIELD_FP_ONLY) || defined(TFHE_DEV_FIELD_FP49_ONLY) || defined(TFHE_DEV_FIELD_FFT_ONLY)
         " [subset build: not every shape / backend]"
#endif
      ;
}

const char* tfhe_status_string(int status) {
  switch (status) {
    case TFHE_OK: return "ok";
    case TFHE_ERR_INVALID_PARAMS: return "invalid parameter set";
    case TFHE_ERR_UNSUPPORTED: return "unsupported shape";
    case TFHE_ERR_NO_KEY: return "bootstrapping key not loaded";
    case TFHE_ERR_HIP: return "HIP runtime error";
    case TFHE_ERR_INVALID_ARGUMENT: return "invalid argument";
    case TFHE_ERR_NO_DEVICE: return "no GPU device (this library has no CPU path)";
    case TFHE_ERR_EXACTNESS: return "parameter set exceeds the exact-NTT bound";
    case TFHE_ERR_IO: return "file missing, truncated, not in this format, or corrupt";
    default: return "unknown status";
  }
}

void tfhe_params_default(tfhe_params* p, int cfg_test) {
  p->glwe_dimension = 2;
  p->glwe_poly_degree = 9;
  p->lwe_dimension = cfg_test ? 4 : 722;
  p->padding_bits = 1;
  p->log_p = 2;
  p->log_q = 32;
  p->ks_decomposer = {4, 5, 32};
  p->pbs_decomposer = {4, 6, 32};
}

int tfhe_params_validate(const tfhe_params* p) {
  if (!p) return TFHE_ERR_INVALID_ARGUMENT;
  if (p->log_q != 32) return TFHE_ERR_INVALID_PARAMS;
  if (decomposer_validate(p->pbs_decomposer) || decomposer_validate(p->ks_decomposer))
    return TFHE_ERR_INVALID_PARAMS;
  if (p->glwe_poly_degree == 0 || p->glwe_poly_degree + 1 >= 32) return TFHE_ERR_INVALID_PARAMS;
  if (p->log_p + p->padding_bits > 32) return TFHE_ERR_INVALID_PARAMS;  // glwe.rs:145
  if (p->log_p > p->glwe_poly_degree) return TFHE_ERR_INVALID_PARAMS;   // test_vector.rs:46
  if (p->lwe_dimension == 0 || p->glwe_dimension == 0) return TFHE_ERR_INVALID_PARAMS;
  return TFHE_OK;
}

const char* tfhe_last_error(const tfhe_context* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int tfhe_context_create_with_backend(const tfhe_params* params, int device, int backend,
                                     tfhe_context** out) {
  if (!params || !out) return TFHE_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int st = tfhe_params_validate(params);
  if (st != TFHE_OK) return st;
  if (!launch::shape_supported(params->glwe_poly_degree, params->glwe_dimension))
    return TFHE_ERR_UNSUPPORTED;
  // exactness of the integer convolution in the chosen field
  // (the fp64 field also relies on |digit| <= B <= 2^kSmallBits for its reduction-free first stage)
  const bool fp_ok = convolution_bits(params, FpField::key_bits()) < FpField::exact_bits() &&
                     params->pbs_decomposer.log_base <= (uint32_t)FpField::kSmallBits;
  const bool fp49_ok = convolution_bits(params, Fp49Field::key_bits()) < Fp49Field::exact_bits() &&
                       (params->glwe_dimension + 1) * params->pbs_decomposer.levels <= (uint32_t)Fp49Field::kMaxRows;
  const bool gl_ok = convolution_bits(params, GlField::key_bits()) < GlField::exact_bits();
  const bool gls_ok = convolution_bits(params, GlSplitField::key_bits()) < GlSplitField::exact_bits();
  // the complex transform is exact while the proven rounding error of an output coefficient stays below
  // FftField::kMaxError (field_fft.h)
  const bool fft_ok = launch::field_shape_supported(launch::kFieldFft, params->glwe_poly_degree) &&
                      params->pbs_decomposer.log_base <= (uint32_t)FftField::kMaxLogBase &&
                      FftField::error_bound((int)params->glwe_poly_degree,
                                            (int)((params->glwe_dimension + 1) * params->pbs_decomposer.levels),
                                            (int)params->pbs_decomposer.log_base) < FftField::kMaxError;
  int field = 0;
  if (backend == TFHE_BACKEND_AUTO) {
    const char* env = std::getenv("TFHE_HIP_BACKEND");
    if (env && std::strcmp(env, "goldilocks") == 0) backend = TFHE_BACKEND_GOLDILOCKS;
    else if (env && std::strcmp(env, "fp64") == 0) backend = TFHE_BACKEND_FP64;
    else if (env && std::strcmp(env, "fp64-p49") == 0) backend = TFHE_BACKEND_FP64_P49;
    else if (env && std::strcmp(env, "goldilocks-split") == 0) backend = TFHE_BACKEND_GOLDILOCKS_SPLIT;
    else if (env && std::strcmp(env, "fp64-fft") == 0) backend = TFHE_BACKEND_FP64_FFT;
  }
  // AUTO: the complex transform wherever its rounding bound holds.  (Round 2 kept the single-spectrum 49-bit field for
  // products of more than 8 digit rows, where fp64-fft's two spectra per key polynomial made the key stream decide.
  // Since round 3 -- six-FMA butterflies, twiddles fetched a transpose ahead, two samples per team at N = 512 with
  // k = 2 and at N = 2048 -- it is ahead or level there too, blind rotation of 2048 samples, profiles/r03_auto_choice.txt:
  // the reference's default parameters (18 rows) 37.5 against 40.8 ms; N = 1024, k = 1, l = 10 (20 rows) 58.2 against
  // 65.3; N = 512, k = 1, l = 6 (12 rows) 19.8 against 20.0; N = 2048, k = 2, l = 5 (15 rows, 512 samples) 34.3 against 34.2.)
  if (backend == TFHE_BACKEND_AUTO)
    field = fft_ok ? launch::kFieldFft
            : fp49_ok ? launch::kFieldFp49
            : fp_ok ? launch::kFieldFp64
            : gl_ok ? launch::kFieldGoldilocks
                    : launch::kFieldGoldilocksSplit;
  else if (backend == TFHE_BACKEND_GOLDILOCKS) field = launch::kFieldGoldilocks;
  else if (backend == TFHE_BACKEND_FP64) field = launch::kFieldFp64;
  else if (backend == TFHE_BACKEND_GOLDILOCKS_SPLIT) field = launch::kFieldGoldilocksSplit;
  else if (backend == TFHE_BACKEND_FP64_P49) field = launch::kFieldFp49;
  else if (backend == TFHE_BACKEND_FP64_FFT) field = launch::kFieldFft;
  else return TFHE_ERR_INVALID_ARGUMENT;
  if (field == launch::kFieldFft && !launch::field_shape_supported(field, params->glwe_poly_degree)) return TFHE_ERR_UNSUPPORTED;
  if ((field == launch::kFieldFft && !fft_ok) || (field == launch::kFieldFp49 && !fp49_ok) || (field == launch::kFieldFp64 && !fp_ok) || (field == launch::kFieldGoldilocks && !gl_ok) ||
      (field == launch::kFieldGoldilocksSplit && !gls_ok))
    return TFHE_ERR_EXACTNESS;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count)
    return TFHE_ERR_NO_DEVICE;

  tfhe_context* ctx = new (std::nothrow) tfhe_context();
  if (!ctx) return TFHE_ERR_HIP;
  ctx->params = *params;
  ctx->device = device;
  ctx->field = field;
  ctx->parts = launch::field_parts(field);
  ctx->N = 1u << params->glwe_poly_degree;
  ctx->R = (params->glwe_dimension + 1) * params->pbs_decomposer.levels;
  ctx->big_n = ctx->N * params->glwe_dimension;  // lib.rs:60

  ctx->pbs.n = params->lwe_dimension;
  ctx->pbs.k = params->glwe_dimension;
  ctx->pbs.log_n = params->glwe_poly_degree;
  ctx->pbs.tv_shift = 32 - params->log_p - params->padding_bits;
  ctx->pbs.log_base = params->pbs_decomposer.log_base;
  ctx->pbs.levels = params->pbs_decomposer.levels;
  ctx->pbs.ignored_bits = 32 - ctx->pbs.log_base * ctx->pbs.levels;
  ctx->pbs.first_shift = gadget_top(ctx, ctx->pbs.log_base) - ctx->pbs.log_base * ctx->pbs.levels;
  ctx->ks.log_base = params->ks_decomposer.log_base;
  ctx->ks.levels = params->ks_decomposer.levels;
  ctx->ks.ignored_bits = 32 - ctx->ks.log_base * ctx->ks.levels;
  ctx->ks.first_shift = gadget_top(ctx, ctx->ks.log_base) - ctx->ks.log_base * ctx->ks.levels;

  auto bail = [&](hipError_t e, const char* what) {
    std::fprintf(stderr, "tfhe_context_create: %s: %s\n", what, hipGetErrorString(e));
    tfhe_context_destroy(ctx);
    return TFHE_ERR_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return bail(e, "hipStreamCreate");
  ctx->own_stream = true;
  if ((e = hipStreamCreateWithFlags(&ctx->side.stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->side.fork, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&ctx->side.join, hipEventDisableTiming)) != hipSuccess)
    return bail(e, "side stream");
  for (auto& slot : ctx->ev_ring)
    for (auto& ev : slot)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return bail(e, "hipEventCreate");
  e = field == launch::kFieldFp64   ? upload_twiddles<FpField>(ctx)
      : field == launch::kFieldFp49 ? upload_twiddles<Fp49Field>(ctx)
      : field == launch::kFieldFft  ? upload_twiddles<FftField>(ctx)
                                    : upload_twiddles<GlField>(ctx);  // both Goldilocks fields share the table
  if (e != hipSuccess) return bail(e, "twiddle upload");
  if ((e = hipMalloc(reinterpret_cast<void**>(&ctx->d_queue), 2 * sizeof(unsigned long long))) != hipSuccess ||
      (e = hipMemset(ctx->d_queue, 0, 2 * sizeof(unsigned long long))) != hipSuccess)
    return bail(e, "work queue");
  // TFHE_KERNEL_SHAPE=wide|team: the starting value of tfhe_context_set_kernel_shape (test sweeps: the whole suite under
  // one shape); unset or anything else: TFHE_SHAPE_AUTO
  if (const char* shape = std::getenv("TFHE_KERNEL_SHAPE")) {
    if (std::strcmp(shape, "wide") == 0) ctx->shape = TFHE_SHAPE_WIDE;
    else if (std::strcmp(shape, "team") == 0) ctx->shape = TFHE_SHAPE_TEAM;
  }
  *out = ctx;
  return TFHE_OK;
}

int tfhe_context_create(const tfhe_params* params, int device, tfhe_context** out) {
  return tfhe_context_create_with_backend(params, device, TFHE_BACKEND_AUTO, out);
}

const char* tfhe_context_backend(const tfhe_context* ctx) {
  if (!ctx) return "";
  return ctx->field == launch::kFieldFp64     ? "fp64-p42"
         : ctx->field == launch::kFieldFp49   ? "fp64-p49"
         : ctx->field == launch::kFieldFft    ? "fp64-fft"
         : ctx->field == launch::kFieldGoldilocks ? "goldilocks"
                                                  : "goldilocks-split";
}

int tfhe_prepared_ggsw_words(const tfhe_context* ctx, size_t* words) {
  if (!ctx || !words) return TFHE_ERR_INVALID_ARGUMENT;
  *words = ggsw_words(ctx) * ctx->parts;
  return TFHE_OK;
}

void tfhe_context_destroy(tfhe_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  void* ptrs[] = {ctx->d_queue,  ctx->d_tw,     ctx->d_bsk,    ctx->d_ksk,    ctx->d_lwe_in, ctx->d_lwe_in2,
                  ctx->d_lwe_big, ctx->d_lwe_out, ctx->d_lwe_ks, ctx->d_glwe_a, ctx->d_glwe_b, ctx->d_glwe_c,
                  ctx->d_tv,     ctx->d_misc,   ctx->d_ggsw_tmp, ctx->d_ggsw_raw,
                  ctx->d_key_tmp};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (auto& g : ctx->gate_tvs)
    if (g.d_tv) (void)hipFree(g.d_tv);
  for (auto& slot : ctx->ev_ring)
    for (auto& ev : slot)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->side.stream) {
    (void)hipStreamSynchronize(ctx->side.stream);
    (void)hipStreamDestroy(ctx->side.stream);
  }
  if (ctx->side.fork) (void)hipEventDestroy(ctx->side.fork);
  if (ctx->side.join) (void)hipEventDestroy(ctx->side.join);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int tfhe_context_set_stream(tfhe_context* ctx, void* hip_stream) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (ctx->stream || !ctx->own_stream) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->own_stream && ctx->stream) HIP_TRY(ctx, hipStreamDestroy(ctx->stream));
  // a null handle is HIP's default stream, which is what torch.cuda.current_stream().cuda_stream
  // is unless the caller switched streams
  ctx->stream = reinterpret_cast<hipStream_t>(hip_stream);
  ctx->own_stream = false;
  return TFHE_OK;
}

int tfhe_context_use_own_stream(tfhe_context* ctx) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (ctx->own_stream) return TFHE_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  ctx->own_stream = true;
  return TFHE_OK;
}

int tfhe_context_set_decomposer_alignment(tfhe_context* ctx, int aligned) {
  int st = check_ctx(ctx);
  if (st) return st;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->aligned = aligned != 0;
  ctx->pbs.first_shift = gadget_top(ctx, ctx->pbs.log_base) - ctx->pbs.log_base * ctx->pbs.levels;
  ctx->ks.first_shift = gadget_top(ctx, ctx->ks.log_base) - ctx->ks.log_base * ctx->ks.levels;
  return TFHE_OK;
}

int tfhe_context_set_kernel_shape(tfhe_context* ctx, int shape) {
  if (!ctx) return TFHE_ERR_INVALID_ARGUMENT;
  if (shape != TFHE_SHAPE_AUTO && shape != TFHE_SHAPE_WIDE && shape != TFHE_SHAPE_TEAM)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "kernel shape: TFHE_SHAPE_AUTO, TFHE_SHAPE_WIDE or TFHE_SHAPE_TEAM");
  ctx->shape = shape;
  return TFHE_OK;
}

int tfhe_context_set_bootstrap_order(tfhe_context* ctx, int ks_first) {
  int st = check_ctx(ctx);
  if (st) return st;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->ks_first = ks_first != 0;
  return TFHE_OK;
}

int tfhe_context_synchronize(tfhe_context* ctx) {
  int st = check_ctx(ctx);
  if (st) return st;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_context_reserve(tfhe_context* ctx, size_t max_batch) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (max_batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "max_batch == 0");
  return reserve(ctx, max_batch);
}

int tfhe_context_set_timing(tfhe_context* ctx, int enable) {
  if (!ctx) return TFHE_ERR_INVALID_ARGUMENT;
  ctx->timing = enable != 0;
  ctx->ev_valid_br = ctx->ev_valid_ks = false;
  ctx->timed_bootstraps = 0;
  return TFHE_OK;
}

int tfhe_measure_hbm_copy(tfhe_context* ctx, size_t bytes, int reps, double* gb_per_s) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!gb_per_s || bytes < 16 || reps <= 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "bytes >= 16, reps >= 1");
  bytes &= ~(size_t)15;
  void *src = nullptr, *dst = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&src, bytes);
  if (e == hipSuccess) e = hipMalloc(&dst, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(src, 1, bytes, ctx->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  for (int i = 0; i < 2 && e == hipSuccess; ++i) e = launch::stream_copy(ctx->stream, src, dst, bytes);
  if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
  for (int i = 0; i < reps && e == hipSuccess; ++i) e = launch::stream_copy(ctx->stream, src, dst, bytes);
  if (e == hipSuccess) e = hipEventRecord(e1, ctx->stream);
  if (e == hipSuccess) e = hipEventSynchronize(e1);
  float ms = 0.f;
  if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (src) (void)hipFree(src);
  if (dst) (void)hipFree(dst);
  if (e != hipSuccess) return hip_fail(ctx, e, "hbm copy probe");
  *gb_per_s = 2.0 * (double)bytes * reps / ((double)ms * 1e-3) / 1e9;  // read + write
  return TFHE_OK;
}

int tfhe_debug_fft_margin(tfhe_context* ctx, double* worst, int reset) {
  int st = check_ctx(ctx);
  if (st) return st;
#if defined(TFHE_FFT_TRACK_ERROR)
  HIP_TRY(ctx, launch::fft_margin(worst, reset != 0));
  return TFHE_OK;
#else
  (void)worst;
  (void)reset;
  return fail(ctx, TFHE_ERR_UNSUPPORTED, "this build carries no rounding-margin probe (libtfhe_hip_probe.so does)");
#endif
}

int tfhe_debug_blind_rotate_plan(tfhe_context* ctx, size_t batch, size_t* samples_per_group, unsigned* segments,
                                 unsigned* streams, size_t* resident_samples) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!samples_per_group || !segments || !streams || !resident_samples) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  launch::BlindRotatePlanInfo plan{};
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  // the plan of tfhe_bootstrap_batch[_device], which reserve the workspace the accumulators are parked in
  HIP_TRY(ctx, launch::blind_rotate_plan(ctx->field, ctx->pbs, batch, !ctx->bmmp, ctx->side.stream != nullptr, &plan, ctx->shape));
  *samples_per_group = plan.chunk;
  *segments = ctx->bmmp ? 1u : plan.segments;
  *streams = ctx->bmmp ? 1u : (unsigned)plan.streams;
  *resident_samples = plan.resident_samples;
  return TFHE_OK;
}

int tfhe_debug_blind_rotate_shape(tfhe_context* ctx, size_t batch, unsigned* waves_per_sample, unsigned* samples_per_team) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!waves_per_sample || !samples_per_team) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  launch::BlindRotatePlanInfo plan{};
  HIP_TRY(ctx, launch::blind_rotate_plan(ctx->field, ctx->pbs, batch, !ctx->bmmp, ctx->side.stream != nullptr, &plan, ctx->shape));
  *waves_per_sample = (unsigned)plan.waves_per_sample;
  *samples_per_team = (unsigned)plan.samples_per_team;
  return TFHE_OK;
}

int tfhe_kernel_ms_ago(tfhe_context* ctx, unsigned steps_ago, float* blind_rotate_ms, float* key_switch_ms) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!blind_rotate_ms || !key_switch_ms) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if (steps_ago >= (unsigned)tfhe_context::kTimingSlots || (unsigned long long)steps_ago >= ctx->timed_bootstraps)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "no timed bootstrap that far back");
  hipEvent_t* ev = ctx->ev_ring[(ctx->ev_slot + tfhe_context::kTimingSlots - (int)steps_ago) % tfhe_context::kTimingSlots];
  HIP_TRY(ctx, hipEventSynchronize(ev[1]));
  HIP_TRY(ctx, hipEventElapsedTime(blind_rotate_ms, ev[0], ev[1]));
  HIP_TRY(ctx, hipEventSynchronize(ev[3]));
  HIP_TRY(ctx, hipEventElapsedTime(key_switch_ms, ev[2], ev[3]));
  return TFHE_OK;
}

int tfhe_last_kernel_ms(tfhe_context* ctx, float* blind_rotate_ms, float* key_switch_ms) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (blind_rotate_ms) *blind_rotate_ms = -1.0f;
  if (key_switch_ms) *key_switch_ms = -1.0f;
  if (ctx->ev_valid_br && blind_rotate_ms) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[1]));
    HIP_TRY(ctx, hipEventElapsedTime(blind_rotate_ms, ctx->ev[0], ctx->ev[1]));
  }
  if (ctx->ev_valid_ks && key_switch_ms) {
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
    HIP_TRY(ctx, hipEventElapsedTime(key_switch_ms, ctx->ev[2], ctx->ev[3]));
  }
  return TFHE_OK;
}

// ---------------------------------------------------------------------------------- keys
// GGSWs of a bootstrapping key: one per key bit, or three per pair of key bits (BMMP)
static size_t key_ggsws(const tfhe_context* ctx, bool bmmp) {
  const size_t n = ctx->params.lwe_dimension;
  return bmmp ? n / 2 * 3 : n;
}

static int check_bmmp(tfhe_context* ctx) {
  if (!launch::shape_supported_bmmp(ctx->pbs.log_n, ctx->pbs.k) || (ctx->params.lwe_dimension & 1u))
    return fail(ctx, TFHE_ERR_UNSUPPORTED, "the unrolled (BMMP) blind rotation needs N = 512 and an even lwe_dimension");
  if (!launch::field_supported_bmmp(ctx->field))
    return fail(ctx, TFHE_ERR_UNSUPPORTED,
                std::string("the unrolled (BMMP) blind rotation is offered in the goldilocks and fp64-p49 backends only (its three "
                            "accumulator sets spill 50-172 registers in the others and it runs slower than the loop there); this "
                            "context uses ") + tfhe_context_backend(ctx) + ": create it with TFHE_BACKEND_GOLDILOCKS or "
                            "TFHE_BACKEND_FP64_P49, or load an ordinary key");
  return TFHE_OK;
}

static int load_key_common(tfhe_context* ctx, const u32* d_bsk_raw, const u32* d_ksk_raw,
                           bool ksk_needs_copy, bool bmmp = false) {
  const size_t ggsws = key_ggsws(ctx, bmmp);
  const size_t bsk_polys = ggsws * ctx->R * (ctx->params.glwe_dimension + 1);
  const size_t ksk_words = (size_t)ctx->big_n * ctx->ks.levels * ((size_t)ctx->params.lwe_dimension + 1);
  // the key buffers are overwritten in place: until the new key is complete the context holds none (a failure half way
  // must not leave it bootstrapping under a half-written key)
  ctx->have_key = false;
  if (ctx->d_bsk && ctx->bsk_ggsws != ggsws) {  // the other kind of key was loaded before
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipError_t e = hipFree(ctx->d_bsk);
    ctx->d_bsk = nullptr;
    if (e != hipSuccess) return hip_fail(ctx, e, "hipFree(bsk)");
  }
  if (!ctx->d_bsk) {
    HIP_TRY(ctx, hipMalloc(&ctx->d_bsk, bsk_polys * ctx->parts * ctx->N * sizeof(u64)));
    ctx->bsk_ggsws = ggsws;
  }
  if (!ctx->d_ksk)
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->d_ksk), ksk_words * sizeof(u32)));
  HIP_TRY(ctx, launch::bsk_prepare(ctx->stream, ctx->field, ctx->pbs.log_n, ctx->pbs.k, ctx->d_tw, d_bsk_raw, bsk_polys, ctx->d_bsk));
  if (ksk_needs_copy)
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ksk, d_ksk_raw, ksk_words * sizeof(u32),
                                hipMemcpyDeviceToDevice, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_key = true;
  ctx->bmmp = bmmp;
  return TFHE_OK;
}

static int load_key_host(tfhe_context* ctx, const uint32_t* bsk, const uint32_t* ksk, bool bmmp) {
  if (!bsk || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null key pointer");
  const size_t bsk_words = key_ggsws(ctx, bmmp) * ggsw_words(ctx);
  const size_t ksk_words = (size_t)ctx->big_n * ctx->ks.levels * ((size_t)ctx->params.lwe_dimension + 1);
  u32* d_raw = nullptr;
  HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_raw), bsk_words * sizeof(u32)));
  hipError_t e = hipMemcpy(d_raw, bsk, bsk_words * sizeof(u32), hipMemcpyHostToDevice);
  if (e == hipSuccess && !ctx->d_ksk)
    e = hipMalloc(reinterpret_cast<void**>(&ctx->d_ksk), ksk_words * sizeof(u32));
  // the key-switching key is overwritten in place by a copy that does not order itself behind the context's stream:
  // whatever was enqueued under the old key finishes first, and from here on the context holds no key
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e == hipSuccess) {
    ctx->have_key = false;
    e = hipMemcpy(ctx->d_ksk, ksk, ksk_words * sizeof(u32), hipMemcpyHostToDevice);
  }
  if (e != hipSuccess) {
    (void)hipFree(d_raw);
    return hip_fail(ctx, e, "key upload");
  }
  int st = load_key_common(ctx, d_raw, nullptr, false, bmmp);
  (void)hipFree(d_raw);
  return st;
}

int tfhe_load_bootstrapping_key(tfhe_context* ctx, const uint32_t* bsk, const uint32_t* ksk) {
  int st = check_ctx(ctx);
  if (st) return st;
  return load_key_host(ctx, bsk, ksk, false);
}

int tfhe_load_bootstrapping_key_device(tfhe_context* ctx, const uint32_t* bsk, const uint32_t* ksk) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!bsk || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null key pointer");
  return load_key_common(ctx, bsk, ksk, true);
}

int tfhe_load_bootstrapping_key_bmmp(tfhe_context* ctx, const uint32_t* bsk_bmmp, const uint32_t* ksk) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_bmmp(ctx))) return st;
  return load_key_host(ctx, bsk_bmmp, ksk, true);
}

int tfhe_load_bootstrapping_key_bmmp_device(tfhe_context* ctx, const uint32_t* bsk_bmmp, const uint32_t* ksk) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_bmmp(ctx))) return st;
  if (!bsk_bmmp || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null key pointer");
  return load_key_common(ctx, bsk_bmmp, ksk, true, true);
}

int tfhe_context_uses_bmmp(const tfhe_context* ctx) { return ctx && ctx->have_key && ctx->bmmp ? 1 : 0; }

// ---------------------------------------------------------------------------------- bootstrap
int tfhe_bootstrap_batch_device(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch,
                                const uint32_t* tv, size_t tv_count, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_batch_args(ctx, lwe_in, tv, lwe_out, batch, tv_count))) return st;
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if ((st = reserve(ctx, batch))) return st;
  return enqueue_bootstrap(ctx, lwe_in, batch, tv, tv_count, ctx->d_lwe_big, lwe_out);
}

int tfhe_bootstrap_batch(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch, const uint32_t* tv,
                         size_t tv_count, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_batch_args(ctx, lwe_in, tv, lwe_out, batch, tv_count))) return st;
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if ((st = check_tv_host(ctx, tv, tv_count * ctx->N))) return st;
  if ((st = reserve(ctx, batch))) return st;
  const size_t n1 = io_words(ctx);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lwe_in, lwe_in, batch * n1 * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tv, tv, tv_count * ctx->N * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = enqueue_bootstrap(ctx, ctx->d_lwe_in, batch, ctx->d_tv, tv_count, ctx->d_lwe_big, ctx->d_lwe_out)))
    return st;
  HIP_TRY(ctx, hipMemcpyAsync(lwe_out, ctx->d_lwe_out, batch * n1 * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_blind_rotate_batch_device(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch,
                                   const uint32_t* tv, size_t tv_count, uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_batch_args(ctx, lwe_in, tv, glwe_out, batch, tv_count))) return st;
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  HIP_TRY(ctx, enqueue_blind_rotate(ctx, lwe_in, batch, tv, tv_count, glwe_out, nullptr));
  if (ctx->timing) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->ev_valid_br = true;
    ctx->ev_valid_ks = false;
  }
  return TFHE_OK;
}

int tfhe_blind_rotate_batch(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch,
                            const uint32_t* tv, size_t tv_count, uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if ((st = check_batch_args(ctx, lwe_in, tv, glwe_out, batch, tv_count))) return st;
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if ((st = check_tv_host(ctx, tv, tv_count * ctx->N))) return st;
  if ((st = reserve(ctx, batch))) return st;
  const size_t n1 = (size_t)ctx->params.lwe_dimension + 1;
  const size_t glwe = (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lwe_in, lwe_in, batch * n1 * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_tv, tv, tv_count * ctx->N * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_blind_rotate_batch_device(ctx, ctx->d_lwe_in, batch, ctx->d_tv, tv_count, ctx->d_glwe_a)))
    return st;
  HIP_TRY(ctx, hipMemcpyAsync(glwe_out, ctx->d_glwe_a, batch * glwe * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_sample_extract_batch(tfhe_context* ctx, const uint32_t* glwe, size_t batch,
                              size_t sample_index, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (sample_index >= ctx->N)  // assert!(sample_index < degree), bootstrapping.rs:127
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "sample_index >= N (bootstrapping.rs:127)");
  if ((st = reserve(ctx, batch))) return st;
  const size_t glwe_w = (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  const size_t out_w = (size_t)ctx->big_n + 1;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_glwe_a, glwe, batch * glwe_w * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::sample_extract(ctx->stream, ctx->pbs.log_n, ctx->pbs.k, ctx->d_glwe_a, batch,
                                      (u32)sample_index, ctx->d_lwe_big));
  HIP_TRY(ctx, hipMemcpyAsync(lwe_out, ctx->d_lwe_big, batch * out_w * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_key_switch_batch_device(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch,
                                 uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_in || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], ctx->stream));
  HIP_TRY(ctx, launch::key_switch(ctx->stream, ctx->ks, ctx->big_n, ctx->params.lwe_dimension, lwe_in,
                                  batch, ctx->d_ksk, lwe_out));
  if (ctx->timing) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    ctx->ev_valid_ks = true;
    ctx->ev_valid_br = false;
  }
  return TFHE_OK;
}

int tfhe_key_switch_batch(tfhe_context* ctx, const uint32_t* lwe_in, size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_in || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if ((st = reserve(ctx, batch))) return st;
  const size_t in_w = (size_t)ctx->big_n + 1, out_w = (size_t)ctx->params.lwe_dimension + 1;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_lwe_big, lwe_in, batch * in_w * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_key_switch_batch_device(ctx, ctx->d_lwe_big, batch, ctx->d_lwe_out))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(lwe_out, ctx->d_lwe_out, batch * out_w * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

// ---------------------------------------------------------------------------------- ggsw.rs
int tfhe_prepare_ggsw_device(tfhe_context* ctx, const uint32_t* ggsw, size_t ggsw_count,
                             void* ggsw_prepared) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ggsw || !ggsw_prepared || ggsw_count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / zero count");
  const size_t polys = ggsw_count * ctx->R * (ctx->params.glwe_dimension + 1);
  HIP_TRY(ctx, launch::bsk_prepare(ctx->stream, ctx->field, ctx->pbs.log_n, ctx->pbs.k, ctx->d_tw, ggsw, polys, ggsw_prepared));
  return TFHE_OK;
}

int tfhe_external_product_prepared_device(tfhe_context* ctx, const void* ggsw_prepared,
                                          size_t ggsw_count, const uint32_t* glwe_in, size_t batch,
                                          uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ggsw_prepared || !glwe_in || !glwe_out || batch == 0)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (ggsw_count != 1 && ggsw_count != batch)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "ggsw_count must be 1 or batch");
  if (batch > kMaxBatch) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "batch exceeds 2^31 - 1");
  if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], ctx->stream));
  HIP_TRY(ctx, launch::external_product(ctx->stream, ctx->field, ctx->pbs, ctx->d_tw, ggsw_prepared,
                                        ggsw_count == 1 ? 0 : ggsw_words(ctx) * ctx->parts, glwe_in,
                                        nullptr, nullptr, batch, glwe_out, ctx->d_queue));
  if (ctx->timing) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev[1], ctx->stream));
    ctx->ev_valid_br = true;
    ctx->ev_valid_ks = false;
  }
  return TFHE_OK;
}

static int upload_and_prepare_ggsw(tfhe_context* ctx, const u32* ggsw, size_t ggsw_count) {
  const size_t words = ggsw_count * ggsw_words(ctx);
  int st;
  if ((st = ensure(ctx, &ctx->d_ggsw_raw, &ctx->ggsw_raw_words, words))) return st;
  if ((st = ensure(ctx, &ctx->d_ggsw_tmp, &ctx->ggsw_tmp_words, words * ctx->parts))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ggsw_raw, ggsw, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  return tfhe_prepare_ggsw_device(ctx, ctx->d_ggsw_raw, ggsw_count, ctx->d_ggsw_tmp);
}

int tfhe_external_product_batch(tfhe_context* ctx, const uint32_t* ggsw, size_t ggsw_count,
                                const uint32_t* glwe_in, size_t batch, uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ggsw || !glwe_in || !glwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (ggsw_count != 1 && ggsw_count != batch) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "ggsw_count must be 1 or batch");
  if ((st = reserve(ctx, batch))) return st;
  if ((st = upload_and_prepare_ggsw(ctx, ggsw, ggsw_count))) return st;
  const size_t glwe = (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_glwe_a, glwe_in, batch * glwe * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_external_product_prepared_device(ctx, ctx->d_ggsw_tmp, ggsw_count,
                                                  ctx->d_glwe_a, batch, ctx->d_glwe_b)))
    return st;
  HIP_TRY(ctx, hipMemcpyAsync(glwe_out, ctx->d_glwe_b, batch * glwe * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_cmux_batch(tfhe_context* ctx, const uint32_t* ggsw, size_t ggsw_count, const uint32_t* ct0,
                    uint32_t* ct1, size_t batch, uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ggsw || !ct0 || !ct1 || !glwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (ggsw_count != 1 && ggsw_count != batch) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "ggsw_count must be 1 or batch");
  if ((st = reserve(ctx, batch))) return st;
  if ((st = upload_and_prepare_ggsw(ctx, ggsw, ggsw_count))) return st;
  const size_t glwe = (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_glwe_a, ct0, batch * glwe * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_glwe_b, ct1, batch * glwe * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::external_product(ctx->stream, ctx->field, ctx->pbs, ctx->d_tw, ctx->d_ggsw_tmp,
                                        ggsw_count == 1 ? 0 : ggsw_words(ctx) * ctx->parts, nullptr,
                                        ctx->d_glwe_b, ctx->d_glwe_a, batch, ctx->d_glwe_c, ctx->d_queue));
  HIP_TRY(ctx, hipMemcpyAsync(glwe_out, ctx->d_glwe_c, batch * glwe * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ct1, ctx->d_glwe_b, batch * glwe * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

// ---------------------------------------------------------------------------------- small ops
int tfhe_decompose(tfhe_context* ctx, int which, const uint32_t* values, size_t count, uint32_t* digits_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!values || !digits_out || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / zero count");
  if (which != TFHE_DECOMPOSER_PBS && which != TFHE_DECOMPOSER_KS) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "bad decomposer selector");
  const tfhe_decomposer_params& d = which == TFHE_DECOMPOSER_PBS ? ctx->params.pbs_decomposer : ctx->params.ks_decomposer;
  const size_t in_b = count * sizeof(u32), out_b = count * d.levels * sizeof(u32);
  if ((st = ensure_misc(ctx, in_b + out_b))) return st;
  u32* d_in = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_out = d_in + count;
  HIP_TRY(ctx, hipMemcpyAsync(d_in, values, in_b, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::decompose_words(ctx->stream, d.log_base, d.levels,
                                       gadget_top(ctx, d.log_base) - d.log_base * d.levels, d_in, count, d_out));
  HIP_TRY(ctx, hipMemcpyAsync(digits_out, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_decompose_glwe_batch(tfhe_context* ctx, const uint32_t* glwe, size_t batch, uint32_t* digits_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe || !digits_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const u32 polys = ctx->params.glwe_dimension + 1;
  const size_t in_w = batch * polys * ctx->N, out_w = in_w * ctx->pbs.levels;
  if ((st = ensure_misc(ctx, (in_w + out_w) * sizeof(u32)))) return st;
  u32* d_in = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_out = d_in + in_w;
  HIP_TRY(ctx, hipMemcpyAsync(d_in, glwe, in_w * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::decompose_glwe(ctx->stream, ctx->pbs.log_base, ctx->pbs.levels, ctx->pbs.first_shift, polys, ctx->N, d_in, batch, d_out));
  HIP_TRY(ctx, hipMemcpyAsync(digits_out, d_out, out_w * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_switch_modulus(tfhe_context* ctx, const uint32_t* values, size_t count, uint32_t log_from,
                        uint32_t log_to, uint32_t* out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!values || !out || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / zero count");
  // `1 << (log_from - log_to)` and `1 << log_to` on u32 (utils.rs:27-28)
  if (log_from > 32 || log_to > log_from || log_from - log_to >= 32 || log_to >= 32)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "switch_modulus shift out of range");
  if ((st = ensure_misc(ctx, 2 * count * sizeof(u32)))) return st;
  u32* d_in = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_out = d_in + count;
  HIP_TRY(ctx, hipMemcpyAsync(d_in, values, count * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::switch_modulus(ctx->stream, d_in, count, log_from, log_to, d_out));
  HIP_TRY(ctx, hipMemcpyAsync(out, d_out, count * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_glwe_mul_monomial_batch(tfhe_context* ctx, const uint32_t* glwe_in, size_t batch,
                                 const int64_t* monomial_index, uint32_t* glwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_in || !monomial_index || !glwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const u32 polys = ctx->params.glwe_dimension + 1;
  const size_t w = batch * polys * ctx->N;
  const size_t idx_b = ((batch * sizeof(i64) + 15) / 16) * 16;
  if ((st = ensure_misc(ctx, idx_b + 2 * w * sizeof(u32)))) return st;
  i64* d_idx = reinterpret_cast<i64*>(ctx->d_misc);
  u32* d_in = reinterpret_cast<u32*>(reinterpret_cast<char*>(ctx->d_misc) + idx_b);
  u32* d_out = d_in + w;
  HIP_TRY(ctx, hipMemcpyAsync(d_idx, monomial_index, batch * sizeof(i64), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(d_in, glwe_in, w * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::glwe_mul_monomial(ctx->stream, ctx->pbs.log_n, polys, d_in, batch, d_idx, d_out));
  HIP_TRY(ctx, hipMemcpyAsync(glwe_out, d_out, w * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

// ---------------------------------------------------------------------------------- lwe.rs
int tfhe_lwe_linear_batch_device(tfhe_context* ctx, uint32_t c0, const uint32_t* ct0, uint32_t c1,
                                 const uint32_t* ct1, size_t batch, size_t words_per_ct, uint32_t* out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ct0 || !out || batch == 0 || words_per_ct == 0 || (c1 != 0 && !ct1))
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  HIP_TRY(ctx, launch::lwe_linear(ctx->stream, c0, ct0, c1, c1 ? ct1 : nullptr, batch * words_per_ct, out));
  return TFHE_OK;
}

int tfhe_lwe_linear_batch(tfhe_context* ctx, uint32_t c0, const uint32_t* ct0, uint32_t c1,
                          const uint32_t* ct1, size_t batch, size_t words_per_ct, uint32_t* out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ct0 || !out || batch == 0 || words_per_ct == 0 || (c1 != 0 && !ct1))
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = batch * words_per_ct;
  if ((st = ensure_misc(ctx, 3 * words * sizeof(u32)))) return st;
  u32* d0 = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d1 = d0 + words;
  u32* dout = d1 + words;
  HIP_TRY(ctx, hipMemcpyAsync(d0, ct0, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if (c1) HIP_TRY(ctx, hipMemcpyAsync(d1, ct1, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_lwe_linear_batch_device(ctx, c0, d0, c1, c1 ? d1 : nullptr, batch, words_per_ct, dout))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(out, dout, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

// ---------------------------------------------------------------------------------- encryption side
// keygen / encrypt / decrypt with the caller's randomness already in the buffers (tfhe_hip.h)
namespace {

int check_binary(tfhe_context* ctx, const u32* sk, size_t words, const char* what) {
  for (size_t i = 0; i < words; ++i)
    if (sk[i] > 1u)
      return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, std::string(what) + " must be binary (sample_binary)");
  return TFHE_OK;
}

int ensure_key_tmp(tfhe_context* ctx, size_t words) {
  if (words <= ctx->key_tmp_words) return TFHE_OK;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return ensure(ctx, &ctx->d_key_tmp, &ctx->key_tmp_words, words);
}

int to_key_tmp(tfhe_context* ctx, const u32* host, size_t words, size_t at) {
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_key_tmp + at, host, words * sizeof(u32), hipMemcpyHostToDevice,
                              ctx->stream));
  return TFHE_OK;
}

// rows of GLWE ciphertexts [rows][k+1][N] on the device; sk already at d_key_tmp[0 .. kN)
int glwe_rows_add_mask_dot_key(tfhe_context* ctx, u32* d_rows, size_t rows) {
  const u32 k = ctx->params.glwe_dimension;
  HIP_TRY(ctx, launch::glwe_body(ctx->stream, ctx->field, ctx->pbs.log_n, ctx->d_tw, k, d_rows, rows,
                                 ctx->d_key_tmp, d_rows + (size_t)k * ctx->N,
                                 (size_t)(k + 1) * ctx->N, false));
  return TFHE_OK;
}

// generate_ksk (key_switching.rs:20-60) on device rows; both keys are host pointers
int ksk_gen_device(tfhe_context* ctx, const u32* from_sk, size_t from_dim, const u32* to_sk,
                   size_t to_dim, u32* d_ksk) {
  int st;
  const u32 levels = ctx->ks.levels, log_base = ctx->ks.log_base;
  const u32 top = gadget_top(ctx, log_base);
  const size_t rows = from_dim * levels;
  // row s*levels + level carries s_bit * 2^{log_base*(l - (level+1))} in its b slot (:36-45)
  std::vector<u32> factor(rows);
  for (size_t s = 0; s < from_dim; ++s)
    for (u32 level = 0; level < levels; ++level)
      factor[s * levels + level] = (1u << (top - log_base * (level + 1))) * from_sk[s];
  if ((st = ensure_key_tmp(ctx, to_dim + rows))) return st;
  if ((st = to_key_tmp(ctx, to_sk, to_dim, 0))) return st;
  if ((st = to_key_tmp(ctx, factor.data(), rows, to_dim))) return st;
  HIP_TRY(ctx, launch::lwe_body(ctx->stream, d_ksk, rows, (u32)to_dim, ctx->d_key_tmp,
                                ctx->d_key_tmp + to_dim, d_ksk + to_dim, to_dim + 1, false));
  // `factor` is pageable host memory: the async copy has staged it before returning
  return TFHE_OK;
}

}  // namespace

int tfhe_glwe_encrypt_zero_batch_device(tfhe_context* ctx, const uint32_t* glwe_sk, uint32_t* glwe,
                                        size_t count) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_sk || !glwe || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t kn = (size_t)ctx->params.glwe_dimension * ctx->N;
  if ((st = check_binary(ctx, glwe_sk, kn, "glwe secret key"))) return st;
  if ((st = ensure_key_tmp(ctx, kn))) return st;
  if ((st = to_key_tmp(ctx, glwe_sk, kn, 0))) return st;
  return glwe_rows_add_mask_dot_key(ctx, glwe, count);
}

int tfhe_glwe_encrypt_zero_batch(tfhe_context* ctx, const uint32_t* glwe_sk, uint32_t* glwe, size_t count) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_sk || !glwe || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = count * (size_t)(ctx->params.glwe_dimension + 1) * ctx->N;
  if ((st = ensure_misc(ctx, words * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  HIP_TRY(ctx, hipMemcpyAsync(d, glwe, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_glwe_encrypt_zero_batch_device(ctx, glwe_sk, d, count))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(glwe, d, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_glwe_decrypt_batch(tfhe_context* ctx, const uint32_t* glwe_sk, const uint32_t* glwe, size_t count,
                            uint32_t* plaintext_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_sk || !glwe || !plaintext_out || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const u32 k = ctx->params.glwe_dimension;
  const size_t kn = (size_t)k * ctx->N;
  const size_t words = count * (size_t)(k + 1) * ctx->N;
  if ((st = check_binary(ctx, glwe_sk, kn, "glwe secret key"))) return st;
  if ((st = ensure_key_tmp(ctx, kn))) return st;
  if ((st = ensure_misc(ctx, (words + count * ctx->N) * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_out = d + words;
  if ((st = to_key_tmp(ctx, glwe_sk, kn, 0))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(d, glwe, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, launch::glwe_body(ctx->stream, ctx->field, ctx->pbs.log_n, ctx->d_tw, k, d, count,
                                 ctx->d_key_tmp, d_out, ctx->N, true));
  HIP_TRY(ctx, hipMemcpyAsync(plaintext_out, d_out, count * ctx->N * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_ggsw_encrypt_batch_device(tfhe_context* ctx, const uint32_t* glwe_sk, const uint32_t* messages,
                                   uint32_t* ggsw, size_t count) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_sk || !messages || !ggsw || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const u32 k = ctx->params.glwe_dimension;
  const size_t kn = (size_t)k * ctx->N;
  if ((st = check_binary(ctx, glwe_sk, kn, "glwe secret key"))) return st;
  if ((st = ensure_key_tmp(ctx, kn + count))) return st;
  if ((st = to_key_tmp(ctx, glwe_sk, kn, 0))) return st;
  if ((st = to_key_tmp(ctx, messages, count, kn))) return st;
  if ((st = glwe_rows_add_mask_dot_key(ctx, ggsw, count * ctx->R))) return st;
  HIP_TRY(ctx, launch::ggsw_add_gadget(ctx->stream, ggsw, count, k, ctx->pbs.log_n, ctx->pbs.levels,
                                       ctx->pbs.log_base, gadget_top(ctx, ctx->pbs.log_base),
                                       ctx->d_key_tmp + kn));
  return TFHE_OK;
}

int tfhe_ggsw_encrypt_batch(tfhe_context* ctx, const uint32_t* glwe_sk, const uint32_t* messages,
                            uint32_t* ggsw, size_t count) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!glwe_sk || !messages || !ggsw || count == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = count * ggsw_words(ctx);
  if ((st = ensure_misc(ctx, words * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  HIP_TRY(ctx, hipMemcpyAsync(d, ggsw, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_ggsw_encrypt_batch_device(ctx, glwe_sk, messages, d, count))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(ggsw, d, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_lwe_encrypt_batch_device(tfhe_context* ctx, const uint32_t* lwe_sk, size_t dimension,
                                  const uint32_t* plaintexts, uint32_t* lwe, size_t batch) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !lwe || batch == 0 || dimension == 0 || dimension >= (1u << 31))
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch / bad dimension");
  if ((st = check_binary(ctx, lwe_sk, dimension, "lwe secret key"))) return st;
  if ((st = ensure_key_tmp(ctx, dimension))) return st;
  if ((st = to_key_tmp(ctx, lwe_sk, dimension, 0))) return st;
  HIP_TRY(ctx, launch::lwe_body(ctx->stream, lwe, batch, (u32)dimension, ctx->d_key_tmp, plaintexts,
                                lwe + dimension, dimension + 1, false));
  return TFHE_OK;
}

int tfhe_lwe_encrypt_batch(tfhe_context* ctx, const uint32_t* lwe_sk, size_t dimension,
                           const uint32_t* plaintexts, uint32_t* lwe, size_t batch) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !lwe || batch == 0 || dimension == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = batch * (dimension + 1);
  if ((st = ensure_misc(ctx, (words + batch) * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_pt = d + words;
  HIP_TRY(ctx, hipMemcpyAsync(d, lwe, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if (plaintexts)
    HIP_TRY(ctx, hipMemcpyAsync(d_pt, plaintexts, batch * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_lwe_encrypt_batch_device(ctx, lwe_sk, dimension, plaintexts ? d_pt : nullptr, d, batch))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(lwe, d, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_lwe_decrypt_batch_device(tfhe_context* ctx, const uint32_t* lwe_sk, size_t dimension,
                                  const uint32_t* lwe, size_t batch, uint32_t* plaintext_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !lwe || !plaintext_out || batch == 0 || dimension == 0 || dimension >= (1u << 31))
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch / bad dimension");
  if ((st = check_binary(ctx, lwe_sk, dimension, "lwe secret key"))) return st;
  if ((st = ensure_key_tmp(ctx, dimension))) return st;
  if ((st = to_key_tmp(ctx, lwe_sk, dimension, 0))) return st;
  HIP_TRY(ctx, launch::lwe_body(ctx->stream, lwe, batch, (u32)dimension, ctx->d_key_tmp, nullptr,
                                plaintext_out, 1, true));
  return TFHE_OK;
}

int tfhe_lwe_decrypt_batch(tfhe_context* ctx, const uint32_t* lwe_sk, size_t dimension, const uint32_t* lwe,
                           size_t batch, uint32_t* plaintext_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !lwe || !plaintext_out || batch == 0 || dimension == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = batch * (dimension + 1);
  if ((st = ensure_misc(ctx, (words + batch) * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  u32* d_out = d + words;
  HIP_TRY(ctx, hipMemcpyAsync(d, lwe, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_lwe_decrypt_batch_device(ctx, lwe_sk, dimension, d, batch, d_out))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(plaintext_out, d_out, batch * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_generate_ksk(tfhe_context* ctx, const uint32_t* from_sk, size_t from_dimension,
                      const uint32_t* to_sk, size_t to_dimension, uint32_t* ksk) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!from_sk || !to_sk || !ksk || from_dimension == 0 || to_dimension == 0 || to_dimension >= (1u << 31))
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / bad dimension");
  if ((st = check_binary(ctx, from_sk, from_dimension, "from secret key"))) return st;
  if ((st = check_binary(ctx, to_sk, to_dimension, "to secret key"))) return st;
  const size_t words = from_dimension * ctx->ks.levels * (to_dimension + 1);
  if ((st = ensure_misc(ctx, words * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  HIP_TRY(ctx, hipMemcpyAsync(d, ksk, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = ksk_gen_device(ctx, from_sk, from_dimension, to_sk, to_dimension, d))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(ksk, d, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_bootstrapping_key_gen_device(tfhe_context* ctx, const uint32_t* lwe_sk, const uint32_t* glwe_sk,
                                      uint32_t* bsk, uint32_t* ksk, int load) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !glwe_sk || !bsk || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  const size_t n = ctx->params.lwe_dimension;
  if ((st = check_binary(ctx, lwe_sk, n, "lwe secret key"))) return st;
  // encrypt each bit of the lwe secret key (bootstrapping.rs:32-38)
  if ((st = tfhe_ggsw_encrypt_batch_device(ctx, glwe_sk, lwe_sk, bsk, n))) return st;
  // key switching key from the flattened GLWE key to the LWE key (:41-51, lwe.rs:62-73)
  if ((st = ksk_gen_device(ctx, glwe_sk, ctx->big_n, lwe_sk, n, ksk))) return st;
  if (load) return load_key_common(ctx, bsk, ksk, true);
  return TFHE_OK;
}

int tfhe_bootstrapping_key_gen(tfhe_context* ctx, const uint32_t* lwe_sk, const uint32_t* glwe_sk,
                               uint32_t* bsk, uint32_t* ksk, int load) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !glwe_sk || !bsk || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  const size_t bsk_words = (size_t)ctx->params.lwe_dimension * ggsw_words(ctx);
  const size_t ksk_words = (size_t)ctx->big_n * ctx->ks.levels * ((size_t)ctx->params.lwe_dimension + 1);
  u32 *d_bsk = nullptr, *d_ksk = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_bsk), bsk_words * sizeof(u32));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ksk), ksk_words * sizeof(u32));
  if (e == hipSuccess) e = hipMemcpy(d_bsk, bsk, bsk_words * sizeof(u32), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ksk, ksk, ksk_words * sizeof(u32), hipMemcpyHostToDevice);
  st = (e == hipSuccess) ? tfhe_bootstrapping_key_gen_device(ctx, lwe_sk, glwe_sk, d_bsk, d_ksk, load)
                         : hip_fail(ctx, e, "key buffers");
  if (st == TFHE_OK) {
    e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(bsk, d_bsk, bsk_words * sizeof(u32), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ksk, d_ksk, ksk_words * sizeof(u32), hipMemcpyDeviceToHost);
    if (e != hipSuccess) st = hip_fail(ctx, e, "key download");
  }
  if (d_bsk) (void)hipFree(d_bsk);
  if (d_ksk) (void)hipFree(d_ksk);
  return st;
}

// notes/BMMP Bootstrapping.md:22-24: the three GGSW messages of key-bit pair j
static std::vector<u32> bmmp_messages(const u32* lwe_sk, size_t n) {
  std::vector<u32> m(n / 2 * 3);
  for (size_t j = 0; j < n / 2; ++j) {
    const u32 s0 = lwe_sk[2 * j], s1 = lwe_sk[2 * j + 1];
    m[3 * j + 0] = s0 * s1;
    m[3 * j + 1] = s0 * (1u - s1);
    m[3 * j + 2] = s1 * (1u - s0);
  }
  return m;
}

int tfhe_bootstrapping_key_gen_bmmp_device(tfhe_context* ctx, const uint32_t* lwe_sk, const uint32_t* glwe_sk,
                                           uint32_t* bsk_bmmp, uint32_t* ksk, int load) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !glwe_sk || !bsk_bmmp || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if ((st = check_bmmp(ctx))) return st;
  const size_t n = ctx->params.lwe_dimension;
  if ((st = check_binary(ctx, lwe_sk, n, "lwe secret key"))) return st;
  const std::vector<u32> messages = bmmp_messages(lwe_sk, n);
  if ((st = tfhe_ggsw_encrypt_batch_device(ctx, glwe_sk, messages.data(), bsk_bmmp, messages.size()))) return st;
  if ((st = ksk_gen_device(ctx, glwe_sk, ctx->big_n, lwe_sk, n, ksk))) return st;
  // `messages` is pageable host memory: the async copy inside has staged it before returning
  if (load) return load_key_common(ctx, bsk_bmmp, ksk, true, true);
  return TFHE_OK;
}

int tfhe_bootstrapping_key_gen_bmmp(tfhe_context* ctx, const uint32_t* lwe_sk, const uint32_t* glwe_sk,
                                    uint32_t* bsk_bmmp, uint32_t* ksk, int load) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!lwe_sk || !glwe_sk || !bsk_bmmp || !ksk) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer");
  if ((st = check_bmmp(ctx))) return st;
  const size_t bsk_words = key_ggsws(ctx, true) * ggsw_words(ctx);
  const size_t ksk_words = (size_t)ctx->big_n * ctx->ks.levels * ((size_t)ctx->params.lwe_dimension + 1);
  u32 *d_bsk = nullptr, *d_ksk = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_bsk), bsk_words * sizeof(u32));
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&d_ksk), ksk_words * sizeof(u32));
  if (e == hipSuccess) e = hipMemcpy(d_bsk, bsk_bmmp, bsk_words * sizeof(u32), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_ksk, ksk, ksk_words * sizeof(u32), hipMemcpyHostToDevice);
  st = (e == hipSuccess) ? tfhe_bootstrapping_key_gen_bmmp_device(ctx, lwe_sk, glwe_sk, d_bsk, d_ksk, load)
                         : hip_fail(ctx, e, "key buffers");
  if (st == TFHE_OK) {
    e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = hipMemcpy(bsk_bmmp, d_bsk, bsk_words * sizeof(u32), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ksk, d_ksk, ksk_words * sizeof(u32), hipMemcpyDeviceToHost);
    if (e != hipSuccess) st = hip_fail(ctx, e, "key download");
  }
  if (d_bsk) (void)hipFree(d_bsk);
  if (d_ksk) (void)hipFree(d_ksk);
  return st;
}

// ---------------------------------------------------------------------------------- on-disk format
namespace {

constexpr char kFileMagic[8] = {'T', 'F', 'H', 'E', 'A', 'M', 'D', '\1'};
constexpr size_t kFileHeaderBytes = 104;

struct FileHeader {
  u32 kind = 0, flags = 0;
  u32 params[12] = {};
  u32 ndims = 0, dims[4] = {1, 1, 1, 1};
  u64 words = 0, checksum = 0;
};

u64 fnv1a64(const unsigned char* p, size_t len, u64 h = 0xcbf29ce484222325ull) {
  for (size_t i = 0; i < len; ++i) {
    h ^= p[i];
    h *= 0x100000001b3ull;
  }
  return h;
}

void put_u32(unsigned char* p, u32 v) { for (int i = 0; i < 4; ++i) p[i] = (unsigned char)(v >> (8 * i)); }
void put_u64(unsigned char* p, u64 v) { for (int i = 0; i < 8; ++i) p[i] = (unsigned char)(v >> (8 * i)); }
u32 get_u32(const unsigned char* p) { u32 v = 0; for (int i = 0; i < 4; ++i) v |= (u32)p[i] << (8 * i); return v; }
u64 get_u64(const unsigned char* p) { u64 v = 0; for (int i = 0; i < 8; ++i) v |= (u64)p[i] << (8 * i); return v; }

void params_to_words(const tfhe_params& p, u32 out[12]) {
  const u32 w[12] = {p.glwe_dimension, p.glwe_poly_degree, p.lwe_dimension, p.padding_bits, p.log_p, p.log_q,
                     p.ks_decomposer.log_base, p.ks_decomposer.levels, p.ks_decomposer.log_q,
                     p.pbs_decomposer.log_base, p.pbs_decomposer.levels, p.pbs_decomposer.log_q};
  std::memcpy(out, w, sizeof(w));
}

void words_to_params(const u32 w[12], tfhe_params* p) {
  p->glwe_dimension = w[0];
  p->glwe_poly_degree = w[1];
  p->lwe_dimension = w[2];
  p->padding_bits = w[3];
  p->log_p = w[4];
  p->log_q = w[5];
  p->ks_decomposer = {w[6], w[7], w[8]};
  p->pbs_decomposer = {w[9], w[10], w[11]};
}

int read_header(std::FILE* f, FileHeader* h) {
  unsigned char b[kFileHeaderBytes];
  if (std::fread(b, 1, sizeof(b), f) != sizeof(b)) return TFHE_ERR_IO;
  if (std::memcmp(b, kFileMagic, 8) != 0) return TFHE_ERR_IO;
  h->kind = get_u32(b + 8);
  h->flags = get_u32(b + 12);
  for (int i = 0; i < 12; ++i) h->params[i] = get_u32(b + 16 + 4 * i);
  h->ndims = get_u32(b + 64);
  for (int i = 0; i < 4; ++i) h->dims[i] = get_u32(b + 68 + 4 * i);
  h->words = get_u64(b + 88);
  h->checksum = get_u64(b + 96);
  if (h->kind < TFHE_FILE_BSK || h->kind > TFHE_FILE_WORDS || h->ndims == 0 || h->ndims > 4) return TFHE_ERR_IO;
  u64 prod = 1;
  for (u32 i = 0; i < 4; ++i) {
    if (h->dims[i] == 0 || (i >= h->ndims && h->dims[i] != 1)) return TFHE_ERR_IO;
    if (prod > (~0ull) / h->dims[i]) return TFHE_ERR_IO;
    prod *= h->dims[i];
  }
  if (prod != h->words) return TFHE_ERR_IO;
  // words * 4 + header must not wrap (a crafted header with words = 2^62 would otherwise pass the
  // size check below as 0 and make the caller allocate from attacker-chosen dims)
  if (h->words > ((u64)SIZE_MAX - kFileHeaderBytes) / sizeof(u32)) return TFHE_ERR_IO;
  return TFHE_OK;
}

}  // namespace

int tfhe_file_write(const char* path, uint32_t kind, const tfhe_params* params, uint32_t flags,
                    const uint32_t* dims, uint32_t ndims, const uint32_t* data) {
  if (!path || !params || !dims || !data || ndims == 0 || ndims > 4 || kind < TFHE_FILE_BSK || kind > TFHE_FILE_WORDS)
    return TFHE_ERR_INVALID_ARGUMENT;
  u64 words = 1;
  for (u32 i = 0; i < ndims; ++i) {
    if (dims[i] == 0 || words > (~0ull) / dims[i]) return TFHE_ERR_INVALID_ARGUMENT;
    words *= dims[i];
  }
  unsigned char b[kFileHeaderBytes] = {};
  std::memcpy(b, kFileMagic, 8);
  put_u32(b + 8, kind);
  put_u32(b + 12, flags);
  u32 pw[12];
  params_to_words(*params, pw);
  for (int i = 0; i < 12; ++i) put_u32(b + 16 + 4 * i, pw[i]);
  put_u32(b + 64, ndims);
  for (u32 i = 0; i < 4; ++i) put_u32(b + 68 + 4 * i, i < ndims ? dims[i] : 1u);
  put_u64(b + 88, words);
  // payload is written as little-endian u32 words; this library only targets little-endian hosts,
  // so the in-memory bytes are the file bytes
  put_u64(b + 96, fnv1a64(reinterpret_cast<const unsigned char*>(data), (size_t)words * sizeof(u32)));
  std::FILE* f = std::fopen(path, "wb");
  if (!f) return TFHE_ERR_IO;
  bool ok = std::fwrite(b, 1, sizeof(b), f) == sizeof(b) &&
            std::fwrite(data, sizeof(u32), (size_t)words, f) == (size_t)words;
  ok = (std::fclose(f) == 0) && ok;
  return ok ? TFHE_OK : TFHE_ERR_IO;
}

int tfhe_file_read_header(const char* path, uint32_t* kind, tfhe_params* params, uint32_t* flags,
                          uint32_t dims[4], uint32_t* ndims, uint64_t* words) {
  if (!path) return TFHE_ERR_INVALID_ARGUMENT;
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return TFHE_ERR_IO;
  FileHeader h;
  int st = read_header(f, &h);
  if (st == TFHE_OK) {  // the payload must be all there, and nothing after it
    if (std::fseek(f, 0, SEEK_END) != 0) st = TFHE_ERR_IO;
    const long end = std::ftell(f);
    if (end < 0 || (u64)end != kFileHeaderBytes + h.words * sizeof(u32)) st = TFHE_ERR_IO;
  }
  std::fclose(f);
  if (st) return st;
  if (kind) *kind = h.kind;
  if (params) words_to_params(h.params, params);
  if (flags) *flags = h.flags;
  if (dims) std::memcpy(dims, h.dims, sizeof(h.dims));
  if (ndims) *ndims = h.ndims;
  if (words) *words = h.words;
  return TFHE_OK;
}

int tfhe_file_read(const char* path, uint32_t* data, uint64_t words) {
  if (!path || !data) return TFHE_ERR_INVALID_ARGUMENT;
  std::FILE* f = std::fopen(path, "rb");
  if (!f) return TFHE_ERR_IO;
  FileHeader h;
  int st = read_header(f, &h);
  if (st == TFHE_OK && h.words != words) st = TFHE_ERR_INVALID_ARGUMENT;
  if (st == TFHE_OK && std::fread(data, sizeof(u32), (size_t)words, f) != (size_t)words) st = TFHE_ERR_IO;
  unsigned char extra;
  if (st == TFHE_OK && std::fread(&extra, 1, 1, f) != 0) st = TFHE_ERR_IO;
  std::fclose(f);
  if (st == TFHE_OK &&
      fnv1a64(reinterpret_cast<const unsigned char*>(data), (size_t)words * sizeof(u32)) != h.checksum)
    st = TFHE_ERR_IO;
  return st;
}

// ---------------------------------------------------------------------------------- test vectors / gates
int tfhe_construct_test_from_lut(const tfhe_params* params, const uint32_t* lut, size_t lut_len, uint32_t* out) {
  if (!params || !lut || !out) return TFHE_ERR_INVALID_ARGUMENT;
  int st = tfhe_params_validate(params);
  if (st) return st;
  return test_from_lut(params, lut, lut_len, out);
}

int tfhe_construct_test_vector_boolean(const tfhe_params* params, const uint32_t truth[4], uint32_t* out) {
  if (!params || !truth || !out) return TFHE_ERR_INVALID_ARGUMENT;
  int st = tfhe_params_validate(params);
  if (st) return st;
  const u32 pm = 1u << params->log_p;
  std::vector<u32> lut(pm);
  for (u32 i = 0; i < pm; ++i) lut[i] = truth[(((i >> 1) & 1u) << 1) | (i & 1u)];  // test_vector.rs:16
  return test_from_lut(params, lut.data(), pm, out);
}

// notes/Boolean Gates.md:2-11: a gate of m inputs is one PBS of c_in = sum_i 2^i * cts[i] with the
// test vector of lut[x] = truth[x mod 2^m]; and()/or() (boolean.rs:9-53) are the m = 2 case.
static int lut_gate_device(tfhe_context* ctx, const u32* truth, u32 inputs, const u32* const* cts,
                           size_t batch, u32* lwe_out) {
  int st;
  if (!truth || !cts || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  if (inputs == 0 || inputs > ctx->params.log_p || inputs > 8)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "gate inputs must be 1..min(log_p, 8): the plaintext space holds log_p bits");
  for (u32 i = 0; i < inputs; ++i)
    if (!cts[i]) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null input ciphertext");
  if (!ctx->have_key) return fail(ctx, TFHE_ERR_NO_KEY, "load the bootstrapping key first");
  if ((st = reserve(ctx, batch))) return st;
  const size_t words = batch * io_words(ctx);
  const u32 entries = 1u << inputs;
  tfhe_context::GateTv* slot = nullptr;
  for (auto& g : ctx->gate_tvs)
    if (g.truth.size() == entries && std::memcmp(g.truth.data(), truth, entries * sizeof(u32)) == 0) slot = &g;
  if (!slot) {
    // first use of this truth table: build its test vector on the host and upload it (this one
    // call synchronises; later calls with a table already seen do not).  At most kMaxGateTvs
    // tables are kept; the least recently used one is replaced.
    constexpr size_t kMaxGateTvs = 64;
    const u32 pm = 1u << ctx->params.log_p;
    std::vector<u32> lut(pm), tv(ctx->N);
    for (u32 x = 0; x < pm; ++x) lut[x] = truth[x & (entries - 1)];  // test_vector.rs:16 for m = 2
    if ((st = test_from_lut(&ctx->params, lut.data(), pm, tv.data())))
      return fail(ctx, st, "truth table / plaintext space mismatch");
    if ((st = check_tv_host(ctx, tv.data(), ctx->N))) return st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->gate_tvs.size() < kMaxGateTvs) {
      ctx->gate_tvs.emplace_back();
      slot = &ctx->gate_tvs.back();
      HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&slot->d_tv), ctx->N * sizeof(u32)));
    } else {
      slot = &ctx->gate_tvs[0];
      for (auto& g : ctx->gate_tvs)
        if (g.last_use < slot->last_use) slot = &g;
    }
    slot->truth.clear();  // not a valid entry until the upload has succeeded
    HIP_TRY(ctx, hipMemcpy(slot->d_tv, tv.data(), ctx->N * sizeof(u32), hipMemcpyHostToDevice));
    slot->truth.assign(truth, truth + entries);
  }
  slot->last_use = ++ctx->gate_clock;
  const u32* d_in = cts[0];
  for (u32 i = 1; i < inputs; ++i) {  // 2*ct1 + ct0 (boolean.rs:18), then + 4*ct2, ...
    HIP_TRY(ctx, launch::lwe_linear(ctx->stream, 1u, d_in, 1u << i, cts[i], words, ctx->d_lwe_in2));
    d_in = ctx->d_lwe_in2;
  }
  return enqueue_bootstrap(ctx, d_in, batch, slot->d_tv, 1, ctx->d_lwe_big, lwe_out);
}

int tfhe_gate_batch_device(tfhe_context* ctx, const uint32_t truth[4], const uint32_t* ct0,
                           const uint32_t* ct1, size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ct0 || !ct1) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const u32* cts[2] = {ct0, ct1};
  return lut_gate_device(ctx, truth, 2, cts, batch, lwe_out);
}

int tfhe_lut_gate_batch_device(tfhe_context* ctx, const uint32_t* truth, uint32_t inputs,
                               const uint32_t* const* cts, size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  return lut_gate_device(ctx, truth, inputs, cts, batch, lwe_out);
}

int tfhe_lut_gate_batch(tfhe_context* ctx, const uint32_t* truth, uint32_t inputs, const uint32_t* const* cts,
                        size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!truth || !cts || !lwe_out || batch == 0 || inputs == 0 || inputs > 8)
    return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch / bad input count");
  const size_t words = batch * io_words(ctx);
  if ((st = ensure_misc(ctx, (size_t)(inputs + 1) * words * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  const u32* d_cts[8];
  for (u32 i = 0; i < inputs; ++i) {
    if (!cts[i]) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null input ciphertext");
    HIP_TRY(ctx, hipMemcpyAsync(d + i * words, cts[i], words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
    d_cts[i] = d + i * words;
  }
  u32* d_out = d + (size_t)inputs * words;
  if ((st = lut_gate_device(ctx, truth, inputs, d_cts, batch, d_out))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(lwe_out, d_out, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

int tfhe_gate_batch(tfhe_context* ctx, const uint32_t truth[4], const uint32_t* ct0,
                    const uint32_t* ct1, size_t batch, uint32_t* lwe_out) {
  const uint32_t* cts[2] = {ct0, ct1};
  if (!ct0 || !ct1) return ctx ? fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch") : TFHE_ERR_INVALID_ARGUMENT;
  return tfhe_lut_gate_batch(ctx, truth, 2, cts, batch, lwe_out);
}

// NOT needs no bootstrap: an encryption of 1 - m is (-a, enc(1) - b), enc(1) = 1 << (32 - log_p - padding)
int tfhe_lwe_not_batch_device(tfhe_context* ctx, const uint32_t* ct, size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ct || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t n1 = io_words(ctx);
  const u32 one = 1u << (32 - ctx->params.log_p - ctx->params.padding_bits);
  HIP_TRY(ctx, launch::lwe_linear(ctx->stream, 0xFFFFFFFFu, ct, 0u, nullptr, batch * n1, lwe_out, n1, one));
  return TFHE_OK;
}

int tfhe_lwe_not_batch(tfhe_context* ctx, const uint32_t* ct, size_t batch, uint32_t* lwe_out) {
  int st = check_ctx(ctx);
  if (st) return st;
  if (!ct || !lwe_out || batch == 0) return fail(ctx, TFHE_ERR_INVALID_ARGUMENT, "null pointer / empty batch");
  const size_t words = batch * io_words(ctx);
  if ((st = ensure_misc(ctx, 2 * words * sizeof(u32)))) return st;
  u32* d = reinterpret_cast<u32*>(ctx->d_misc);
  HIP_TRY(ctx, hipMemcpyAsync(d, ct, words * sizeof(u32), hipMemcpyHostToDevice, ctx->stream));
  if ((st = tfhe_lwe_not_batch_device(ctx, d, batch, d + words))) return st;
  HIP_TRY(ctx, hipMemcpyAsync(lwe_out, d + words, words * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return TFHE_OK;
}

}  // extern "C"

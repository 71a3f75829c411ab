// emu.cpp -- host SIMT emulator for the per-wavefront kernel bodies (CPU test harness only).
//
// Compiles tfhe-research_amd/csrc/{goldilocks,wave_ntt,pbs_wave}.h with g++ and runs a 64-lane
// "wavefront" as 64 OS threads; Ctx::sync() is a pthread barrier, LDS is a heap buffer.  This lets
// `pytest -m "not gpu"` check the exact device code (layouts, twiddles, swizzles, decomposer,
// rotation signs) bit-for-bit against the oracle without a GPU.  It is a test of the product's
// source, not a fallback: nothing in the shipped library can reach it.
#include <pthread.h>

#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "pbs_wave.h"

using namespace tfhe;

namespace {

struct HostWave {
  int lane_;
  pthread_barrier_t* bar_;
  u64* scratch_;
  u32* acc_;
  const u64* tw_;
  int lane() const { return lane_; }
  void sync() const { pthread_barrier_wait(bar_); }
  u64* scratch() const { return scratch_; }
  u32* acc() const { return acc_; }
  const u64* twiddles() const { return tw_; }
  u32 uniform(u32 v) const { return v; }
};

// run body(ctx) on 64 lanes
void run_wave(int logn, int k, const std::function<void(const HostWave&)>& body) {
  const int n = 1 << logn;
  std::vector<u64> scratch(n), tw(n);
  std::vector<u32> acc((size_t)(k + 1) * n);
  ntt_fill_twiddles(logn, tw.data());
  pthread_barrier_t bar;
  pthread_barrier_init(&bar, nullptr, kWave);
  std::vector<std::thread> th;
  for (int l = 0; l < kWave; ++l)
    th.emplace_back([&, l] {
      HostWave w{l, &bar, scratch.data(), acc.data(), tw.data()};
      body(w);
    });
  for (auto& t : th) t.join();
  pthread_barrier_destroy(&bar);
}

template <int LOGN>
void poly_ntt(const u64* in, u64* out, int inverse) {
  constexpr int E = NttShape<LOGN>::kE;
  run_wave(LOGN, 0, [&](const HostWave& w) {
    u64 x[E];
    if (!inverse) {
      for (int r = 0; r < E; ++r) x[r] = in[r * 64 + w.lane()];
      ntt_forward<LOGN>(w, x);
      for (int r = 0; r < E; ++r) out[w.lane() * E + r] = x[r];
    } else {
      for (int r = 0; r < E; ++r) x[r] = in[w.lane() * E + r];
      ntt_inverse<LOGN>(w, x);
      for (int r = 0; r < E; ++r) out[r * 64 + w.lane()] = x[r];
    }
  });
}

template <int LOGN>
void bsk_prepare(size_t polys, const u32* src, u64* dst) {
  constexpr int N = 1 << LOGN;
  const u64 n_inv = gl::inv((u64)N);
  run_wave(LOGN, 0, [&](const HostWave& w) {
    for (size_t i = 0; i < polys; ++i) bsk_prepare_wave<LOGN>(w, src + i * N, dst + i * N, n_inv);
  });
}

template <int LOGN, int K>
void blind_rotate(const PbsParams& P, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                  const u64* bsk, u32* out_glwe, u32* out_lwe) {
  constexpr int N = 1 << LOGN;
  constexpr int E = NttShape<LOGN>::kE;
  run_wave(LOGN, K, [&](const HostWave& w) {
    for (size_t b = 0; b < batch; ++b) {
      blind_rotate_wave<LOGN, K>(w, P, lwe + b * (P.n + 1), tv + b * tv_stride, bsk);
      if (out_glwe)
        for (int p = 0; p <= K; ++p)
          for (int r = 0; r < E; ++r)
            out_glwe[(b * (K + 1) + p) * N + r * 64 + w.lane()] = w.acc()[p * N + r * 64 + w.lane()];
      if (out_lwe) sample_extract_wave<LOGN, K>(w, out_lwe + b * ((size_t)K * N + 1));
      w.sync();
    }
  });
}

template <int LOGN, int K>
void ext_product(const PbsParams& P, const u64* ggsw, const u32* glwe, u32* out) {
  constexpr int N = 1 << LOGN;
  run_wave(LOGN, K, [&](const HostWave& w) {
    auto src = [&](int p, int j) -> u32 { return glwe[p * N + j]; };
    auto dst = [&](int p, int j, u32 v) { out[p * N + j] = v; };
    external_product_wave<LOGN, K>(w, P, ggsw, src, dst);
  });
}

PbsParams make_params(u32 n, u32 k, u32 log_n, u32 log_p, u32 padding, u32 log_base, u32 levels) {
  PbsParams P;
  P.n = n;
  P.k = k;
  P.log_n = log_n;
  P.tv_shift = 32 - log_p - padding;
  P.log_base = log_base;
  P.levels = levels;
  P.ignored_bits = 32 - log_base * levels;
  P.first_shift = log_base * (32 / log_base - levels);
  return P;
}

}  // namespace

#define DISPATCH_LOGN(logn, CALL)            \
  switch (logn) {                            \
    case 9: { constexpr int L = 9; CALL; break; }   \
    case 10: { constexpr int L = 10; CALL; break; } \
    case 11: { constexpr int L = 11; CALL; break; } \
    default: return 1;                       \
  }

extern "C" {

int emu_poly_ntt(int logn, const u64* in, u64* out, int inverse) {
  DISPATCH_LOGN(logn, poly_ntt<L>(in, out, inverse));
  return 0;
}

int emu_twiddles(int logn, u64* out) {
  ntt_fill_twiddles(logn, out);
  return 0;
}

u64 emu_gl_mul(u64 a, u64 b) { return gl::mul(a, b); }
u64 emu_gl_add(u64 a, u64 b) { return gl::add(a, b); }
u64 emu_gl_sub(u64 a, u64 b) { return gl::sub(a, b); }
u64 emu_gl_from_i32(u32 d) { return gl::from_i32(d); }
u32 emu_gl_lift(u64 v) { return gl::lift_mod_2_32(v); }
void emu_gl_mul_many(const u64* a, const u64* b, u64* out, size_t len) {
  for (size_t i = 0; i < len; ++i) out[i] = gl::mul(a[i], b[i]);
}

int emu_bsk_prepare(int logn, size_t polys, const u32* src, u64* dst) {
  DISPATCH_LOGN(logn, bsk_prepare<L>(polys, src, dst));
  return 0;
}

int emu_blind_rotate(u32 n, u32 k, u32 logn, u32 log_p, u32 padding, u32 log_base, u32 levels,
                     size_t batch, const u32* lwe, const u32* tv, size_t tv_stride, const u64* bsk,
                     u32* out_glwe, u32* out_lwe) {
  PbsParams P = make_params(n, k, logn, log_p, padding, log_base, levels);
  if (k == 1) { DISPATCH_LOGN(logn, (blind_rotate<L, 1>(P, batch, lwe, tv, tv_stride, bsk, out_glwe, out_lwe))); }
  else if (k == 2) { DISPATCH_LOGN(logn, (blind_rotate<L, 2>(P, batch, lwe, tv, tv_stride, bsk, out_glwe, out_lwe))); }
  else return 2;
  return 0;
}

int emu_external_product(u32 k, u32 logn, u32 log_base, u32 levels, const u64* ggsw,
                         const u32* glwe, u32* out) {
  PbsParams P = make_params(0, k, logn, 2, 1, log_base, levels);
  if (k == 1) { DISPATCH_LOGN(logn, (ext_product<L, 1>(P, ggsw, glwe, out))); }
  else if (k == 2) { DISPATCH_LOGN(logn, (ext_product<L, 2>(P, ggsw, glwe, out))); }
  else return 2;
  return 0;
}

}  // extern "C"

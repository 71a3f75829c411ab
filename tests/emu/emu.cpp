// emu.cpp -- host SIMT emulator for the per-wavefront kernel bodies (CPU test harness only).
//
// Compiles tfhe-research_amd/csrc/{goldilocks,wave_ntt,pbs_wave}.h with g++ and runs a 64-lane
// "wavefront" as 64 OS threads; Ctx::sync() is a pthread barrier, LDS is a heap buffer.  This lets
// `pytest -m "not gpu"` check the exact device code (layouts, twiddles, swizzles, decomposer,
// rotation signs) bit-for-bit against the oracle without a GPU.  It is a test of the product's
// source, not a fallback: nothing in the shipped library can reach it.
#define TFHE_FFT_TRACK_ERROR 1  // record how far the complex transform's lifted values are from integers
#include <pthread.h>

#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "pbs_wave.h"

using namespace tfhe;

namespace {

int g_exchange_buffers = 1;  // exchange buffers per group (kernels.hip::ExchangeBuffersOf)
int g_segments = 1;  // launches a blind rotation is cut into (kernels.hip::blind_rotate_plan): resumes from parked accumulators
int g_samples_per_team = 1;  // samples a team rotates at once (kernels.hip::SamplesPerTeam): each needs a buffer and an accumulator

template <class Elem>
struct HostTeam {
  int n, g;  // ring degree, waves per polynomial group
  int ns;    // transform elements per polynomial (n, or n/2 for the complex transform)
  int exb, groups, samples;
  std::vector<Elem> scratch, tw, tw_natural;
  std::vector<u32> acc;
  pthread_barrier_t team_bar;
  std::vector<pthread_barrier_t> group_bars;
  std::vector<pthread_barrier_t> wave_bars;  // lanes are free-running threads: lockstep = a barrier
};

template <class Elem>
struct HostWave {
  int lane_, wave_;
  HostTeam<Elem>* t_;
  int buf_ = 0;  // exchange buffer in use: buffer b of group c is at scratch[(b*groups + c)*n]
  int exchange_buffers() const { return t_->exb; }
  HostWave with_exchange_buffer(int i) const {
    HostWave w = *this;
    w.buf_ = i;
    return w;
  }
  int tid() const { return (wave_ % t_->g) * 64 + lane_; }
  int group() const { return wave_ / t_->g; }
  void poly_sync() const { pthread_barrier_wait(&t_->group_bars[group()]); }
  void wave_sync() const { pthread_barrier_wait(&t_->wave_bars[wave_]); }
  void team_sync() const { pthread_barrier_wait(&t_->team_bar); }
  Elem* scratch() const { return t_->scratch.data() + ((size_t)buf_ * t_->groups + group()) * t_->ns; }
  const Elem* scratch_of(int s) const { return t_->scratch.data() + ((size_t)buf_ * t_->groups + s) * t_->ns; }
  u32* acc(int s = 0) const { return t_->acc.data() + ((size_t)group() * t_->samples + s) * t_->n; }
  const Elem* twiddles() const { return t_->tw.data(); }
  const Elem* twiddles_uniform() const { return t_->tw_natural.data(); }
  u32 uniform(u32 v) const { return v; }
  void lds_add(u32* p, u32 v) const { *p += v; }
  void compiler_fence() const {}
};

// run body(ctx) on a team of `groups` polynomial groups x g waves x 64 lanes (one OS thread per lane)
template <class F>
void run_team(int logn, int groups, int g, const std::function<void(const HostWave<typename F::elem>&)>& body) {
  typedef typename F::elem elem;
  HostTeam<elem> team;
  team.n = 1 << logn;
  team.g = g;
  team.samples = g_samples_per_team;
  team.exb = g_exchange_buffers > team.samples ? g_exchange_buffers : team.samples;
  team.groups = groups;
  const int lt = logn - F::kLogShrink;  // log2 of the transform size
  team.ns = 1 << lt;
  team.scratch.resize((size_t)groups * team.ns * team.exb);
  team.acc.resize((size_t)groups * team.samples * team.n);
  // natural-order table -> the working copy's layout, as the kernels stage it into LDS
  std::vector<elem>& natural = team.tw_natural;
  natural.resize(ntt_twiddle_words(team.ns));
  F::fill_twiddles(lt, natural.data());
  team.tw.resize(natural.size());
  for (int tid = 0; tid < 64; ++tid) {
    switch (lt * 8 + g) {
      case 9 * 8 + 1: ntt_stage_twiddles<9, 1>(team.tw.data(), natural.data(), tid, 64); break;
      case 10 * 8 + 1: ntt_stage_twiddles<10, 1>(team.tw.data(), natural.data(), tid, 64); break;
      case 11 * 8 + 1: ntt_stage_twiddles<11, 1>(team.tw.data(), natural.data(), tid, 64); break;
      case 11 * 8 + 2: ntt_stage_twiddles<11, 2>(team.tw.data(), natural.data(), tid, 64); break;
      case 11 * 8 + 4: ntt_stage_twiddles<11, 4>(team.tw.data(), natural.data(), tid, 64); break;
      case 8 * 8 + 1: ntt_stage_twiddles<8, 1>(team.tw.data(), natural.data(), tid, 64); break;  // complex transform, N = 512
      case 10 * 8 + 4: ntt_stage_twiddles<10, 4>(team.tw.data(), natural.data(), tid, 64); break;  // ... over four waves
      case 10 * 8 + 2: ntt_stage_twiddles<10, 2>(team.tw.data(), natural.data(), tid, 64); break;  // complex transform, N = 2048
      default: std::abort();
    }
  }
  const int waves = groups * g;
  pthread_barrier_init(&team.team_bar, nullptr, waves * kWave);
  team.group_bars.resize(groups);
  for (auto& b : team.group_bars) pthread_barrier_init(&b, nullptr, g * kWave);
  team.wave_bars.resize(waves);
  for (auto& b : team.wave_bars) pthread_barrier_init(&b, nullptr, kWave);
  std::vector<std::thread> th;
  for (int w = 0; w < waves; ++w)
    for (int l = 0; l < kWave; ++l)
      th.emplace_back([&, w, l] {
        HostWave<elem> ctx{l, w, &team, 0};
        body(ctx);
      });
  for (auto& t : th) t.join();
}

// shapes a transform policy can run in the emulator: any for the prime fields; the complex transform holds two
// coefficients per element, so it needs N/2 >= 512 points per G waves x 8 elements
template <class F, int LOGN, int G>
constexpr bool shape_ok() {
  return F::kLogShrink == 0 || ((LOGN == 9 || LOGN == 10) && G == 1) || LOGN == 11;
}

template <class F, int LOGN, int G>
void poly_ntt(const typename F::elem* in, typename F::elem* out, int inverse) {
  typedef typename F::elem elem;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  if constexpr (F::kLogShrink != 0) std::abort();  // (the raw transform test is for the prime fields)
  else
  run_team<F>(LOGN, 1, G, [&](const HostWave<elem>& w) {
    elem x[E];
    if (!inverse) {
      for (int r = 0; r < E; ++r) x[r] = in[r * T + w.tid()];
      ntt_forward<F, LOGN, G>(w, x);
      for (int r = 0; r < E; ++r) out[w.tid() * E + r] = x[r];
    } else {
      for (int r = 0; r < E; ++r) x[r] = in[w.tid() * E + r];
      ntt_inverse<F, LOGN, G>(w, x);
      for (int r = 0; r < E; ++r) out[r * T + w.tid()] = x[r];
    }
  });
}

int g_key_k = 0;  // GLWE dimension of the key being prepared: selects its layout (pbs_wave.h::key_layout_e); 0 = natural

template <class F, int LOGN, int G>
void bsk_prepare(size_t polys, const u32* src, typename F::elem* dst) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  if constexpr (!shape_ok<F, LOGN, G>()) std::abort();
  else {
    const elem n_inv = F::n_inv(LOGN - F::kLogShrink);
    constexpr int PAIR_E = key_layout_e<F, LOGN, 1>();
    run_team<F>(LOGN, 1, G, [&](const HostWave<elem>& w) {
      for (size_t i = 0; i < polys; ++i) {
        if constexpr (PAIR_E != 0) {
          if (g_key_k == 1) {
            bsk_prepare_wave<F, LOGN, G, PAIR_E>(w, src + i * N, dst + i * (N >> F::kLogShrink) * F::kParts, n_inv);
            continue;
          }
        }
        bsk_prepare_wave<F, LOGN, G>(w, src + i * N, dst + i * (N >> F::kLogShrink) * F::kParts, n_inv);
      }
    });
  }
}

template <class F, int LOGN, int K, int G, int NS = 1>
void blind_rotate(const PbsParams& P, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                  const typename F::elem* bsk, u32* out_glwe, u32* out_lwe) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  std::vector<u32> state((size_t)NS * (K + 1) * N);
  if constexpr (!shape_ok<F, LOGN, G>()) std::abort();
  else
  run_team<F>(LOGN, K + 1, G, [&](const HostWave<elem>& w) {
    // teams of NS samples, as blind_rotate_kernel forms them: an odd batch's last team redoes its last sample
    for (size_t b0 = 0; b0 < batch; b0 += NS) {
      size_t idx[NS];
      const u32* lwes[NS];
      const u32* tvs[NS];
      for (int s = 0; s < NS; ++s) {
        idx[s] = b0 + s < batch ? b0 + s : batch - 1;
        lwes[s] = lwe + idx[s] * (P.n + 1);
        tvs[s] = tv + idx[s] * tv_stride;
      }
      // the rotation in g_segments pieces, the accumulators parked in `state` in between -- as blind_rotate_kernel does
      const u32 per = (P.n + g_segments - 1) / g_segments;
      const u32* resume[NS];
      for (int s = 0; s < NS; ++s) resume[s] = state.data() + (size_t)s * (K + 1) * N;
      for (u32 i0 = 0; i0 < P.n; i0 += per) {
        const u32 i1 = i0 + per < P.n ? i0 + per : P.n;
        blind_rotate_team_multi<F, LOGN, K, G, NS>(w, P, lwes, tvs, bsk, i0, i1, resume);
        if (i1 < P.n) {
          for (int s = 0; s < NS; ++s)
            for (int r = 0; r < E; ++r)
              state[((size_t)s * (K + 1) + w.group()) * N + r * T + w.tid()] = w.acc(s)[r * T + w.tid()];
          w.team_sync();
        }
      }
      for (int s = 0; s < NS && b0 + s < batch; ++s) {
        const size_t b = idx[s];
        if (out_glwe)
          for (int r = 0; r < E; ++r)
            out_glwe[(b * (K + 1) + w.group()) * N + r * T + w.tid()] = w.acc(s)[r * T + w.tid()];
        if (out_lwe) sample_extract_team<LOGN, K, G>(w, out_lwe + b * ((size_t)K * N + 1), s);
      }
      w.team_sync();
    }
  });
}

// the unrolled blind rotation (pbs_wave.h::blind_rotate_bmmp_team); bsk: prepared [n/2][3] GGSWs
template <class F, int LOGN, int K, int G>
void blind_rotate_bmmp(const PbsParams& P, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                       const typename F::elem* bsk, u32* out_glwe, u32* out_lwe) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int E = NttShape<LOGN, G>::kE;
  constexpr int T = NttShape<LOGN, G>::kThreads;
  if constexpr (!shape_ok<F, LOGN, G>()) std::abort();
  else
  run_team<F>(LOGN, K + 1, G, [&](const HostWave<elem>& w) {
    for (size_t b = 0; b < batch; ++b) {
      blind_rotate_bmmp_team<F, LOGN, K, G>(w, P, lwe + b * (P.n + 1), tv + b * tv_stride, bsk);
      if (out_glwe)
        for (int r = 0; r < E; ++r)
          out_glwe[(b * (K + 1) + w.group()) * N + r * T + w.tid()] = w.acc()[r * T + w.tid()];
      if (out_lwe) sample_extract_team<LOGN, K, G>(w, out_lwe + b * ((size_t)K * N + 1));
      w.team_sync();
    }
  });
}

template <class F, int LOGN, int K, int G>
void ext_product(const PbsParams& P, const typename F::elem* ggsw, const u32* glwe, u32* out) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  if constexpr (!shape_ok<F, LOGN, G>()) std::abort();
  else
  run_team<F>(LOGN, K + 1, G, [&](const HostWave<elem>& w) {
    const int p = w.group();
    auto src = [&](int j) -> u32 { return glwe[p * N + j]; };
    auto dst = [&](int j, u32 v) { out[p * N + j] = v; };
    external_product_team<F, LOGN, K, G>(w, P, ggsw, src, dst);
  });
}

// body_out = body_in +/- sum_i masks[i] (*) sk[i] for `rows` GLWE rows [k+1][N]
template <class F, int LOGN, int G>
void glwe_body(u32 k, size_t rows, const u32* glwe, const u32* sk, u32* out, int negate) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  if constexpr (!shape_ok<F, LOGN, G>()) std::abort();
  else {
  const elem n_inv = F::n_inv(LOGN - F::kLogShrink);
  run_team<F>(LOGN, 1, G, [&](const HostWave<elem>& w) {
    for (size_t row = 0; row < rows; ++row) {
      const u32* masks = glwe + row * (size_t)(k + 1) * N;
      const u32* body = masks + (size_t)k * N;
      auto dst = [&](int j, u32 dot) { out[row * N + j] = negate ? body[j] - dot : body[j] + dot; };
      glwe_mask_dot_key<F, LOGN, G>(w, k, masks, sk, n_inv, dst);
    }
  });
  }
}

// ---- the wide team (pbs_wave.h::blind_rotate_team_wide): 2 (K+1) waves per sample, one row buffer per digit row ----
template <class Elem>
struct HostWideTeam {
  int n, ns, rows, waves;
  std::vector<Elem> buffers, tw, tw_natural;  // buffers: rows x ns
  std::vector<u32> acc;                       // (K+1) x n
  pthread_barrier_t team_bar;
  std::vector<pthread_barrier_t> wave_bars;
};
template <class Elem>
struct HostWideWave {
  int lane_, wave_;
  HostWideTeam<Elem>* t_;
  int tid() const { return lane_; }
  int group() const { return wave_ >> 1; }
  int half() const { return wave_ & 1; }
  Elem* row_buffer(int r) const { return t_->buffers.data() + (size_t)r * t_->ns; }
  void wave_sync() const { pthread_barrier_wait(&t_->wave_bars[wave_]); }
  void poly_sync() const { wave_sync(); }
  void team_sync() const { pthread_barrier_wait(&t_->team_bar); }
  Elem* scratch() const { return t_->buffers.data() + (size_t)(t_->rows + wave_) * t_->ns; }  // my own transpose buffer
  u32* acc(int = 0) const { return t_->acc.data() + (size_t)group() * t_->n; }
  const Elem* twiddles() const { return t_->tw.data(); }
  const Elem* twiddles_uniform() const { return t_->tw_natural.data(); }
  u32 uniform(u32 v) const { return v; }
  // two waves add to one word here (the halves of a column): the emulator's lanes are OS threads, so this one is atomic
  void lds_add(u32* p, u32 v) const { __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
  void compiler_fence() const {}
};

int g_wide_key_ring = 1;  // 1: the key-ring instantiation where one exists for the level count (2, 3, 4, 6); 0: the generic kernel

template <class F, int LOGN, int K, int LEVELS = 0>
void blind_rotate_wide(const PbsParams& P, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                       const typename F::elem* bsk, u32* out_glwe, u32* out_lwe) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int LT = LOGN - F::kLogShrink;
  HostWideTeam<elem> team;
  team.n = N;
  team.ns = 1 << LT;
  team.waves = 2 * (K + 1);
  team.rows = (K + 1) * (int)P.levels;
  team.buffers.resize((size_t)(team.rows + team.waves) * team.ns);
  team.acc.resize((size_t)(K + 1) * N);
  team.tw_natural.resize(ntt_twiddle_words(team.ns));
  F::fill_twiddles(LT, team.tw_natural.data());
  team.tw.resize(team.tw_natural.size());
  for (int tid = 0; tid < 64; ++tid) ntt_stage_twiddles<LT, 1>(team.tw.data(), team.tw_natural.data(), tid, 64);
  pthread_barrier_init(&team.team_bar, nullptr, team.waves * kWave);
  team.wave_bars.resize(team.waves);
  for (auto& b : team.wave_bars) pthread_barrier_init(&b, nullptr, kWave);
  std::vector<u32> state((size_t)(K + 1) * N);
  auto body = [&](const HostWideWave<elem>& w) {
    constexpr int EC = N / 64;
    const int me = w.group(), q = w.half(), lane = w.tid();
    for (size_t b = 0; b < batch; ++b) {
      const u32 per = (P.n + g_segments - 1) / g_segments;
      for (u32 i0 = 0; i0 < P.n; i0 += per) {
        const u32 i1 = i0 + per < P.n ? i0 + per : P.n;
        blind_rotate_team_wide<F, LOGN, K, LEVELS, wide_ring_rows<LOGN, K, LEVELS>()>(w, P, lwe + b * (P.n + 1), tv + b * tv_stride, bsk, i0, i1,
                                                                                     state.data());
        if (i1 < P.n) {  // park, as blind_rotate_wide_kernel does: each half the words it owns
          for (int r = 0; r < EC / 2; ++r) {
            const int j = (r + q * (EC / 2)) * 64 + lane;
            state[(size_t)me * N + j] = w.acc()[j];
          }
          w.team_sync();
        }
      }
      const u32* acc = w.acc();
      for (int r = 0; r < EC / 2; ++r) {
        const int x = (r + q * (EC / 2)) * 64 + lane;
        if (out_glwe) out_glwe[(b * (K + 1) + me) * N + x] = acc[x];
        if (out_lwe && me < K) out_lwe[b * ((size_t)K * N + 1) + me * N + x] = (x == 0) ? acc[0] : (0u - acc[N - x]);
      }
      if (out_lwe && me == K && q == 0 && lane == 0) out_lwe[b * ((size_t)K * N + 1) + K * N] = acc[0];
      w.team_sync();
    }
  };
  std::vector<std::thread> th;
  for (int wv = 0; wv < team.waves; ++wv)
    for (int l = 0; l < kWave; ++l)
      th.emplace_back([&, wv, l] {
        HostWideWave<elem> ctx{l, wv, &team};
        body(ctx);
      });
  for (auto& t : th) t.join();
}

// ---- the pair kernel (pbs_wave.h::blind_rotate_pair): ONE wave per sample, polynomial c in lanes 32 c .. 32 c + 31 ----
template <class Elem>
struct HostPairTeam {
  int n, ns;
  std::vector<Elem> buffers, tw, tw_natural;  // buffers: 2 x ns
  std::vector<u32> acc;                       // 2 x n
  pthread_barrier_t wave_bar;
};
template <class Elem>
struct HostPairWave {
  int lane_;
  HostPairTeam<Elem>* t_;
  int tid() const { return lane_ & 31; }
  int group() const { return lane_ >> 5; }
  void wave_sync() const { pthread_barrier_wait(&t_->wave_bar); }
  void poly_sync() const { wave_sync(); }
  void team_sync() const { wave_sync(); }
  Elem* scratch() const { return t_->buffers.data() + (size_t)group() * t_->ns; }
  const Elem* scratch_of(int half) const { return t_->buffers.data() + (size_t)half * t_->ns; }
  u32* acc(int = 0) const { return t_->acc.data() + (size_t)group() * t_->n; }
  const Elem* twiddles() const { return t_->tw.data(); }
  const Elem* twiddles_uniform() const { return t_->tw_natural.data(); }
  u32 uniform(u32 v) const { return v; }
  void lds_add(u32* p, u32 v) const { *p += v; }
  void compiler_fence() const {}
};

template <class F, int LOGN>
void blind_rotate_pair_emu(const PbsParams& P, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                           const typename F::elem* bsk, u32* out_glwe, u32* out_lwe) {
  typedef typename F::elem elem;
  constexpr int N = 1 << LOGN;
  constexpr int LT = LOGN - F::kLogShrink;
  constexpr int K = 1;
  HostPairTeam<elem> team;
  team.n = N;
  team.ns = 1 << LT;
  team.buffers.resize((size_t)2 * team.ns);
  team.acc.resize((size_t)2 * N);
  team.tw_natural.resize(ntt_twiddle_words(team.ns));
  F::fill_twiddles(LT, team.tw_natural.data());
  team.tw.resize(team.tw_natural.size());
  for (int tid = 0; tid < 64; ++tid) ntt_stage_twiddles<LT, 0>(team.tw.data(), team.tw_natural.data(), tid, 64);
  pthread_barrier_init(&team.wave_bar, nullptr, kWave);
  std::vector<u32> state((size_t)2 * N);
  auto body = [&](const HostPairWave<elem>& w) {
    constexpr int T = 32, EC = N / T;
    const int me = w.group(), tid = w.tid();
    for (size_t b = 0; b < batch; ++b) {
      const u32 per = (P.n + g_segments - 1) / g_segments;
      for (u32 i0 = 0; i0 < P.n; i0 += per) {
        const u32 i1 = i0 + per < P.n ? i0 + per : P.n;
        blind_rotate_pair<F, LOGN>(w, P, lwe + b * (P.n + 1), tv + b * tv_stride, bsk, i0, i1, state.data());
        if (i1 < P.n) {
          for (int r = 0; r < EC; ++r) state[(size_t)me * N + r * T + tid] = w.acc()[r * T + tid];
          w.wave_sync();
        }
      }
      const u32* acc = w.acc();
      for (int r = 0; r < EC; ++r) {
        const int x = r * T + tid;
        if (out_glwe) out_glwe[(b * (K + 1) + me) * N + x] = acc[x];
        if (out_lwe && me < K) out_lwe[b * ((size_t)K * N + 1) + me * N + x] = (x == 0) ? acc[0] : (0u - acc[N - x]);
      }
      if (out_lwe && me == K && tid == 0) out_lwe[b * ((size_t)K * N + 1) + K * N] = acc[0];
      w.wave_sync();
    }
  };
  std::vector<std::thread> th;
  for (int l = 0; l < kWave; ++l)
    th.emplace_back([&, l] {
      HostPairWave<elem> ctx{l, &team};
      body(ctx);
    });
  for (auto& t : th) t.join();
}

bool g_aligned = false;  // decomposer alignment extension (tfhe_hip.h)

PbsParams make_params(u32 n, u32 k, u32 log_n, u32 log_p, u32 padding, u32 log_base, u32 levels) {
  PbsParams P;
  P.n = n;
  P.k = k;
  P.log_n = log_n;
  P.tv_shift = 32 - log_p - padding;
  P.log_base = log_base;
  P.levels = levels;
  P.ignored_bits = 32 - log_base * levels;
  P.first_shift = (g_aligned ? 32u : log_base * (32 / log_base)) - log_base * levels;
  return P;
}

}  // namespace

// (logn, g): g = waves per polynomial; supported: (9,1) (10,1) (11,1) (11,2) (11,4)
#define DISPATCH_LOGN(logn, g, CALL)                                     \
  if ((logn) == 9 && (g) == 1) { constexpr int L = 9, GG = 1; CALL; }    \
  else if ((logn) == 10 && (g) == 1) { constexpr int L = 10, GG = 1; CALL; } \
  else if ((logn) == 11 && (g) == 1) { constexpr int L = 11, GG = 1; CALL; } \
  else if ((logn) == 11 && (g) == 2) { constexpr int L = 11, GG = 2; CALL; } \
  else if ((logn) == 11 && (g) == 4) { constexpr int L = 11, GG = 4; CALL; } \
  else return 1;

// field: 1 = Goldilocks, 2 = fp64 42-bit prime (double elements), 3 = Goldilocks with split key,
// 4 = fp64 49-bit prime with the key word taken whole; 5 = complex FFT in fp64 (16-byte elements, N/2 of them
// per polynomial); prepared keys are counted in 8-byte words for every field
#define DISPATCH_FIELD(field, CALL)                         \
  if ((field) == 1) { typedef GlField FF; CALL }            \
  else if ((field) == 2) { typedef FpField FF; CALL }       \
  else if ((field) == 3) { typedef GlSplitField FF; CALL }  \
  else if ((field) == 4) { typedef Fp49Field FF; CALL }     \
  else if ((field) == 5) { typedef FftField FF; CALL }      \
  else return 3;

extern "C" {

void emu_set_aligned(int aligned) { g_aligned = aligned != 0; }
void emu_set_exchange_buffers(int n) { g_exchange_buffers = n == 2 ? 2 : 1; }
void emu_set_samples_per_team(int n) { g_samples_per_team = n == 2 ? 2 : 1; }
void emu_set_segments(int n) { g_segments = n > 1 ? n : 1; }
int emu_field_parts(int field) { return (field == 1 || field == 4) ? 1 : 2; }
// 1 if the emulator can run `field` at ring degree 2^logn with g waves per polynomial
int emu_field_shape_ok(int field, int logn, int g) {
  if (field != 5) return 1;
  return ((logn == 9 || logn == 10) && g == 1) || logn == 11;
}
// the hoisted-constant rounding of the hot loop against the literal decomposer.rs:27-40 restatement: number of
// mismatches over `count` words starting at `first` with stride `stride`
unsigned long long emu_round_value_mismatches(unsigned ignored_bits, unsigned first, unsigned stride, unsigned long long count) {
  const RoundConsts rc = round_consts(ignored_bits);
  unsigned long long bad = 0;
  u32 v = first;
  for (unsigned long long i = 0; i < count; ++i, v += stride) bad += round_value_fast(v, rc) != round_value(v, ignored_bits);
  return bad;
}
double emu_fft_error_bound(int logn, int rows, int log_base) { return FftField::error_bound(logn, rows, log_base); }
// largest |value - nearest integer| the complex transform has lifted since the last reset
void emu_fft_error_reset() { fft_error_slot().store(0.0); }
double emu_fft_error_max() { return fft_error_slot().load(); }

int emu_poly_ntt(int field, int logn, int g, const void* in, void* out, int inverse) {
  DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (poly_ntt<FF, L, GG>((const FF::elem*)in, (FF::elem*)out, inverse))));
  return 0;
}

int emu_twiddles(int field, int logn, void* out) {
  DISPATCH_FIELD(field, FF::fill_twiddles(logn, (FF::elem*)out););
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
double emu_fp_p() { return FpField::P; }
void emu_fp_mul_many(const double* a, const double* w, double* out, size_t len) {
  for (size_t i = 0; i < len; ++i) out[i] = FpField::mul(a[i], w[i]);
}
void emu_fp_reduce_many(const double* a, double* out, size_t len) {
  for (size_t i = 0; i < len; ++i) out[i] = FpField::reduce(a[i]);
}
void emu_fp_to_u32_many(const double* a, u32* out, size_t len) {
  for (size_t i = 0; i < len; ++i) out[i] = FpField::to_u32(a[i]);
}
double emu_fp_from_key_word(u32 w, int part) { return FpField::from_key_word(w, part); }

int emu_bsk_prepare(int field, int logn, int g, size_t polys, const u32* src, void* dst) {
  DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (bsk_prepare<FF, L, GG>(polys, src, (FF::elem*)dst))));
  return 0;
}

// N = 512, one wave per polynomial, one exchange buffer (the shape the GPU library offers it for)
int emu_blind_rotate_bmmp(int field, u32 n, u32 k, u32 log_p, u32 padding, u32 log_base, u32 levels, size_t batch,
                          const u32* lwe, const u32* tv, size_t tv_stride, const void* bsk, u32* out_glwe,
                          u32* out_lwe) {
  PbsParams P = make_params(n, k, 9, log_p, padding, log_base, levels);
  if (g_exchange_buffers != 1 || (n & 1u)) return 2;
  if (k == 1) { DISPATCH_FIELD(field, (blind_rotate_bmmp<FF, 9, 1, 1>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe));) }
  else if (k == 2) { DISPATCH_FIELD(field, (blind_rotate_bmmp<FF, 9, 2, 1>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe));) }
  else return 1;
  return 0;
}

int emu_blind_rotate(int field, int g, u32 n, u32 k, u32 logn, u32 log_p, u32 padding, u32 log_base,
                     u32 levels, size_t batch, const u32* lwe, const u32* tv, size_t tv_stride,
                     const void* bsk, u32* out_glwe, u32* out_lwe) {
  PbsParams P = make_params(n, k, logn, log_p, padding, log_base, levels);
  if (g_samples_per_team == 2) {  // two samples per team (emu_set_samples_per_team): the complex transform's kernels use it
    if (field != 5) return 4;
    typedef FftField FF;
    if (k == 1) { DISPATCH_LOGN(logn, g, (blind_rotate<FF, L, 1, GG, 2>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe))); }
    else if (k == 2) { DISPATCH_LOGN(logn, g, (blind_rotate<FF, L, 2, GG, 2>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe))); }
    else return 2;
    return 0;
  }
  if (k == 1) { DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (blind_rotate<FF, L, 1, GG>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe)))); }
  else if (k == 2) { DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (blind_rotate<FF, L, 2, GG>(P, batch, lwe, tv, tv_stride, (const FF::elem*)bsk, out_glwe, out_lwe)))); }
  else return 2;
  return 0;
}

// the wide team: the complex transform, N = 512 / 1024, k = 1 / 2
int emu_blind_rotate_wide(u32 n, u32 k, u32 logn, u32 log_p, u32 padding, u32 log_base, u32 levels, size_t batch,
                          const u32* lwe, const u32* tv, size_t tv_stride, const void* bsk, u32* out_glwe, u32* out_lwe) {
  PbsParams P = make_params(n, k, logn, log_p, padding, log_base, levels);
  typedef FftField FF;
  const FF::elem* key = (const FF::elem*)bsk;
#define WIDE_LEVELS(LOGN_, K_)                                                                                   \
  do {                                                                                                           \
    const int lv = g_wide_key_ring ? (int)levels : 0;                                                            \
    if (lv == 2) blind_rotate_wide<FF, LOGN_, K_, 2>(P, batch, lwe, tv, tv_stride, key, out_glwe, out_lwe);       \
    else if (lv == 3) blind_rotate_wide<FF, LOGN_, K_, 3>(P, batch, lwe, tv, tv_stride, key, out_glwe, out_lwe);  \
    else if (lv == 4) blind_rotate_wide<FF, LOGN_, K_, 4>(P, batch, lwe, tv, tv_stride, key, out_glwe, out_lwe);  \
    else if (lv == 6) blind_rotate_wide<FF, LOGN_, K_, 6>(P, batch, lwe, tv, tv_stride, key, out_glwe, out_lwe);  \
    else blind_rotate_wide<FF, LOGN_, K_, 0>(P, batch, lwe, tv, tv_stride, key, out_glwe, out_lwe);               \
  } while (0)
  if (logn == 9 && k == 1) WIDE_LEVELS(9, 1);
  else if (logn == 9 && k == 2) WIDE_LEVELS(9, 2);
  else if (logn == 10 && k == 1) WIDE_LEVELS(10, 1);
  else if (logn == 10 && k == 2) WIDE_LEVELS(10, 2);
  else return 1;
#undef WIDE_LEVELS
  return 0;
}
void emu_set_wide_key_ring(int on) { g_wide_key_ring = on != 0; }
// the GLWE dimension of the keys emu_bsk_prepare lays out from here on (0: every field's natural layout)
void emu_set_key_k(int k) { g_key_k = k; }
// the pair kernel: the complex transform, N = 512, k = 1
int emu_blind_rotate_pair(u32 n, u32 log_p, u32 padding, u32 log_base, u32 levels, size_t batch, const u32* lwe, const u32* tv,
                          size_t tv_stride, const void* bsk, u32* out_glwe, u32* out_lwe) {
  PbsParams P = make_params(n, 1, 9, log_p, padding, log_base, levels);
  blind_rotate_pair_emu<FftField, 9>(P, batch, lwe, tv, tv_stride, (const FftField::elem*)bsk, out_glwe, out_lwe);
  return 0;
}

int emu_external_product(int field, int g, u32 k, u32 logn, u32 log_base, u32 levels, const void* ggsw,
                         const u32* glwe, u32* out) {
  PbsParams P = make_params(0, k, logn, 2, 1, log_base, levels);
  if (k == 1) { DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (ext_product<FF, L, 1, GG>(P, (const FF::elem*)ggsw, glwe, out)))); }
  else if (k == 2) { DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (ext_product<FF, L, 2, GG>(P, (const FF::elem*)ggsw, glwe, out)))); }
  else return 2;
  return 0;
}

int emu_glwe_body(int field, int logn, int g, u32 k, size_t rows, const u32* glwe, const u32* sk, u32* out,
                  int negate) {
  DISPATCH_FIELD(field, DISPATCH_LOGN(logn, g, (glwe_body<FF, L, GG>(k, rows, glwe, sk, out, negate))));
  return 0;
}

}  // extern "C"

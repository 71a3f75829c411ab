// context.h -- the state behind a tfhe_context handle, shared by the host-side sources of the library
// (capi.cpp: the single-device C ABI; pool.cpp: the multi-device pool built on top of it).  Private to csrc/.
#pragma once
#include "tfhe_hip.h"

#include <hip/hip_runtime.h>

#include <string>
#include <vector>

#include "launch.h"

using namespace tfhe;

struct tfhe_context {
  tfhe_params params;
  PbsParams pbs;
  KsParams ks;
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  launch::SideStream side = {nullptr, nullptr, nullptr};  // the blind rotation's second stream (always the context's own)
  u32 N = 0, R = 0, big_n = 0;

  int field = 0;              // launch::kFieldGoldilocks | launch::kFieldFp64
  int parts = 1;              // spectra per key polynomial in this field
  void* d_tw = nullptr;       // psi_rev[N], 8-byte field elements
  unsigned long long* d_queue = nullptr;  // ticket counter of the external-product kernel's work queue
  void* d_bsk = nullptr;      // prepared BSK [n][R][k+1][parts][N] (spectrum_slot order, x 1/N)
  u32* d_ksk = nullptr;       // [big_n*l_ks][n+1]
  bool have_key = false;
  bool bmmp = false;          // the loaded key is a BMMP key: n/2 * 3 GGSWs (tfhe_load_bootstrapping_key_bmmp)
  size_t bsk_ggsws = 0;       // GGSWs d_bsk was allocated for
  bool aligned = false;       // decomposer alignment (tfhe_context_set_decomposer_alignment)
  bool ks_first = false;      // bootstrap order (tfhe_context_set_bootstrap_order)
  int shape = 0;              // kernel shape of the blind rotation (tfhe_context_set_kernel_shape): launch::kShape*

  // workspace (grown on demand by host-pointer calls or tfhe_context_reserve)
  size_t ws_batch = 0;
  u32* d_lwe_in = nullptr;    // [batch][n+1]
  u32* d_lwe_in2 = nullptr;   // [batch][n+1] second gate operand
  u32* d_lwe_big = nullptr;   // [batch][big_n+1]
  u32* d_lwe_out = nullptr;   // [batch][n+1]
  u32* d_lwe_ks = nullptr;    // [batch][n+1] key-switched input of the KS-then-PBS order
  u32* d_glwe_a = nullptr;    // [batch][k+1][N]
  u32* d_glwe_b = nullptr;
  u32* d_glwe_c = nullptr;
  u32* d_tv = nullptr;        // [batch][N] (or [1][N])
  // test vectors of gate calls, one [N] device buffer per truth table seen (a gate graph alternates
  // between a handful of tables; re-uploading on every switch would synchronise the stream)
  struct GateTv {
    std::vector<u32> truth;  // 2^inputs entries
    u32* d_tv = nullptr;
    unsigned long long last_use = 0;
  };
  std::vector<GateTv> gate_tvs;
  unsigned long long gate_clock = 0;
  // generic scratch for the small entry points
  void* d_misc = nullptr;
  size_t misc_bytes = 0;
  u64* d_ggsw_tmp = nullptr;  // prepared GGSWs of external_product / cmux host calls (8-byte words)
  size_t ggsw_tmp_words = 0;
  u32* d_ggsw_raw = nullptr;
  size_t ggsw_raw_words = 0;
  u32* d_key_tmp = nullptr;   // secret keys / messages of the encryption-side calls
  size_t key_tmp_words = 0;

  bool timing = false;
  // br start/stop, ks start/stop of the current timing slot.  Bootstraps rotate through kTimingSlots sets of
  // events, so that a caller can time K back-to-back steps without a host synchronisation inside the loop and
  // read them all afterwards (tfhe_kernel_ms_ago); the other timed calls use the current set.
  static constexpr int kTimingSlots = 64;
  hipEvent_t ev_ring[kTimingSlots][4] = {};
  hipEvent_t* ev = ev_ring[0];
  int ev_slot = 0;
  unsigned long long timed_bootstraps = 0;
  bool ev_valid_br = false, ev_valid_ks = false;

  std::string last_error;
};

namespace tfhe {
namespace host {

inline int fail(tfhe_context* ctx, int status, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return status;
}

inline int hip_fail(tfhe_context* ctx, hipError_t e, const char* what) {
  return fail(ctx, TFHE_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

// pool.cpp: allocate dst's key buffers like src's and copy the PREPARED bootstrapping key and the key-switching
// key device to device (peer copy over xGMI, or a plain device copy when both contexts sit on one GPU) on dst's
// stream; dst must have been created with the same parameters and backend (capi.cpp)
int adopt_prepared_key(tfhe_context* dst, const tfhe_context* src);
// words of one ciphertext at the bootstrap boundary (n+1, or k*N+1 in KS-first order)
size_t io_words(const tfhe_context* ctx);

}  // namespace host
}  // namespace tfhe

#define HIP_TRY(ctx, expr)                                              \
  do {                                                                  \
    hipError_t _e = (expr);                                             \
    if (_e != hipSuccess) return tfhe::host::hip_fail((ctx), _e, #expr); \
  } while (0)

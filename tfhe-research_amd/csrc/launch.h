// launch.h -- host-callable launchers of the gfx950 kernels (implemented in kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "pbs_wave.h"

namespace tfhe {
namespace launch {

// NTT backends (field policies): values of the `field` argument below
constexpr int kFieldGoldilocks = GlField::kId;  // 1
constexpr int kFieldFp64 = FpField::kId;        // 2
constexpr int kFieldGoldilocksSplit = GlSplitField::kId;  // 3
constexpr int kFieldFp49 = Fp49Field::kId;                // 4
constexpr int kFieldFft = FftField::kId;                  // 5: complex FFT in fp64, exact by a rounding bound

// true if a kernel set is instantiated for (log_n, k)
bool shape_supported(u32 log_n, u32 k);
// spectra per key polynomial for a field (1 or 2)
int field_parts(int field);
// true if the field's kernels exist at this ring degree
bool field_shape_supported(int field, u32 log_n);
// samples a team of the blind-rotation kernel rotates at once for (field, log_n, k): 1 or 2
int samples_per_team(int field, u32 log_n, u32 k);

// twiddle table psi_rev[N] (8-byte field elements) must already be on the device;
// spectra: poly_count x field_parts x N elements; k (the GLWE dimension) selects the key's layout with log_n and the
// field (pbs_wave.h::key_layout_e: the pair kernel's at N = 512, k = 1 in the complex transform)
hipError_t bsk_prepare(hipStream_t s, int field, u32 log_n, u32 k, const void* tw, const u32* polys,
                       size_t poly_count, void* spectra);

// A second stream of the caller's, with the two events that fork it from and join it to the main one: batches larger than
// what the chip holds at once go out as two halves whose segment launches alternate (kernels.hip::blind_rotate_plan).
struct SideStream {
  hipStream_t stream;
  hipEvent_t fork, join;
};

// Kernel shape of a blind rotation (tfhe_context_set_kernel_shape): the launcher's own choice by batch size, the wide
// team (2 (k+1) waves per sample: the latency shape, complex transform up to N = 1024) wherever it is offered, or
// always the throughput team
constexpr int kShapeAuto = 0, kShapeWide = 1, kShapeTeam = 2;

// Blind rotation of `batch` samples.  Optional outputs: glwe_out [batch][k+1][N] and/or
// lwe_extracted [batch][k*N+1] (sample extract at index 0 fused in).
// `state`: [batch][k+1][N] words of scratch that hold the accumulators between the launches of a segmented rotation
// (may be glwe_out itself; null: one launch per rotation).  `side`: null = everything on s.  On return all work is
// ordered on s (the side stream has been joined).
hipError_t blind_rotate(hipStream_t s, int field, const PbsParams& P, const void* tw,
                        const u32* lwe_in, size_t batch, const u32* tv, size_t tv_stride,
                        const void* bsk, u32* glwe_out, u32* lwe_extracted, u32* state = nullptr,
                        const SideStream* side = nullptr, int shape = kShapeAuto);

// How blind_rotate would send out a batch (kernels.hip::blind_rotate_plan), for bench lines and tests
struct BlindRotatePlanInfo {
  size_t chunk;             // samples per group of launches
  u32 segments;             // launches per rotation (key slices)
  int streams;              // 1 or 2
  size_t resident_samples;  // samples the chip rotates at once (teams it holds x samples per team)
  int samples_per_team;
  int waves_per_sample;     // K+1 groups of G waves (shared by samples_per_team samples), or 2 (K+1) for the wide team
};
hipError_t blind_rotate_plan(int field, const PbsParams& P, size_t batch, bool can_park, bool have_side,
                             BlindRotatePlanInfo* out, int shape = kShapeAuto);

// The unrolled blind rotation of notes/BMMP Bootstrapping.md (two key bits per step): bsk holds
// n/2 * 3 prepared GGSWs (pbs_wave.h::blind_rotate_bmmp_team); n even, shape_supported_bmmp only.
bool shape_supported_bmmp(u32 log_n, u32 k);
bool field_supported_bmmp(int field);  // Goldilocks and fp64-p49: the fields whose BMMP kernel does not lose to the loop
hipError_t blind_rotate_bmmp(hipStream_t s, int field, const PbsParams& P, const void* tw,
                             const u32* lwe_in, size_t batch, const u32* tv, size_t tv_stride,
                             const void* bsk, u32* glwe_out, u32* lwe_extracted);

// out = external_product(ggsw[g], glwe[b]) (+ ct0 for the CMUX form).
//   cmux_ct0 == nullptr : src = glwe_in
//   cmux_ct0 != nullptr : src = ct1 - ct0 where ct1 = glwe_inout_ct1 (overwritten with the
//                         difference, ggsw.rs:171), out = product + ct0
hipError_t external_product(hipStream_t s, int field, const PbsParams& P, const void* tw,
                            const void* ggsw, size_t ggsw_stride_words, const u32* glwe_in,
                            u32* ct1_inout, const u32* cmux_ct0, size_t batch, u32* glwe_out,
                            unsigned long long* queue /* one device word of scratch: ticket counter of long batches */);

// key_switch_lwe over a batch: lwe_in [batch][big_n+1], ksk [big_n*levels][n+1], out [batch][n+1]
hipError_t key_switch(hipStream_t s, const KsParams& K, u32 big_n, u32 n, const u32* lwe_in,
                      size_t batch, const u32* ksk, u32* lwe_out);

// elementwise helpers.  first_shift = bit of the lowest kept limb (PbsParams::first_shift)
hipError_t decompose_words(hipStream_t s, u32 log_base, u32 levels, u32 first_shift, const u32* values,
                           size_t count, u32* digits /* [count][levels] */);
hipError_t decompose_glwe(hipStream_t s, u32 log_base, u32 levels, u32 first_shift, u32 polys_per_ct,
                          u32 n_coeff, const u32* glwe, size_t batch,
                          u32* digits /* [batch][polys*levels][N] */);
hipError_t switch_modulus(hipStream_t s, const u32* values, size_t count, u32 log_from, u32 log_to,
                          u32* out);
hipError_t glwe_mul_monomial(hipStream_t s, u32 log_n, u32 polys_per_ct, const u32* glwe,
                             size_t batch, const i64* monomial_index, u32* out);
hipError_t sample_extract(hipStream_t s, u32 log_n, u32 k, const u32* glwe, size_t batch,
                          u32 sample_index, u32* lwe_out);
// out = c0*ct0 + c1*ct1 (ct1 may be null when c1 == 0); lwe.rs:9-23, boolean.rs:18.  With
// words_per_ct != 0, b_add is added to the b slot (last word) of every ciphertext.
hipError_t lwe_linear(hipStream_t s, u32 c0, const u32* ct0, u32 c1, const u32* ct1, size_t words,
                      u32* out, size_t words_per_ct = 0, u32 b_add = 0);

// ---- encryption side (keygen / encrypt / decrypt with caller-supplied randomness)
// dst[row][j] = body(rows[row])[j] +/- sum_i masks(rows[row])[i] (*) sk[i]; rows [row_count][k+1][N],
// sk [k][N] binary, dst rows of dst_stride words (may alias the bodies)
hipError_t glwe_body(hipStream_t s, int field, u32 log_n, const void* tw, u32 k, const u32* rows,
                     size_t row_count, const u32* sk, u32* dst, size_t dst_stride, bool negate);
// dst[row*dst_stride] = rows[row][n] +/- <rows[row][0..n), sk> (+ plaintext[row] if non-null)
hipError_t lwe_body(hipStream_t s, const u32* rows, size_t row_count, u32 n, const u32* sk,
                    const u32* plaintext, u32* dst, size_t dst_stride, bool negate);
// ggsw.rs:96-103 for ggsw_count matrices [(k+1)*levels][k+1][N]
hipError_t ggsw_add_gadget(hipStream_t s, u32* ggsw, size_t ggsw_count, u32 k, u32 log_n, u32 levels,
                           u32 log_base, u32 gadget_top, const u32* messages);

// Probe builds (-DTFHE_FFT_TRACK_ERROR, libtfhe_hip_probe.so): largest |value - nearest integer| the complex
// transform has lifted on the device since the last reset.  The product build has no probe: hipErrorNotSupported.
hipError_t fft_margin(double* worst, bool reset);

// dst[0..bytes) = src[0..bytes) with 16-byte accesses (bytes a multiple of 16): HBM roofline probe
hipError_t stream_copy(hipStream_t s, const void* src, void* dst, size_t bytes);

}  // namespace launch
}  // namespace tfhe

// Sanitizer driver for the device headers run through the host SIMT emulator (CPU only; the GPU
// pool has no ASan): builds emu.cpp with -fsanitize=address,undefined and pushes random operands
// through bsk_prepare, external product, blind rotation + sample extract and the keygen wave
// function for every shipped shape, in both exchange-buffer schemes.  LDS buffers are exact-size
// heap vectors in the emulator, so an out-of-range slot, swizzle or twiddle index trips ASan.
//   g++ -O1 [-g] -std=c++17 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=all
//       -pthread -I tfhe-research_amd/csrc tests/emu/sanitize_main.cpp -o tests/emu/sanitize && tests/emu/sanitize
#include "emu.cpp"

#include <cstdio>
#include <random>

int main() {
  std::mt19937_64 gen(1);
  auto fill = [&](std::vector<u32>& v) { for (auto& x : v) x = (u32)gen(); };
  struct Shape { int logn, g; u32 k, log_base, levels; };
  const Shape shapes[] = {{9, 1, 1, 8, 2}, {9, 1, 2, 4, 6}, {10, 1, 1, 7, 3}, {11, 2, 2, 8, 4}, {11, 4, 2, 8, 4}, {11, 4, 1, 16, 2}};
  for (int exb = 1; exb <= 2; ++exb) {
    emu_set_exchange_buffers(exb);
    for (const Shape& s : shapes) {
      for (int field = 1; field <= 5; ++field) {
        if (field == 2 && s.log_base > 9) continue;  // outside the fp64 field's small-digit bound
        // complex transform: the shapes the GPU library ships (N = 512, 1024; N = 2048 over four waves), inside its bound
        if (field == 5 && (!(s.logn == 9 || s.logn == 10 || (s.logn == 11 && s.g == 4)) || s.log_base > 13)) continue;
        // 49-bit field: only where (k+1) l N B 2^31 < 2^48.25 (here: the reference default shape)
        if (field == 4 && !(s.logn == 9 && s.k == 2 && s.log_base == 4)) continue;
        const size_t N = (size_t)1 << s.logn, R = (size_t)(s.k + 1) * s.levels, n = 3;
        const int parts = emu_field_parts(field);
        std::vector<u32> bsk(n * R * (s.k + 1) * N), glwe((s.k + 1) * N), out((s.k + 1) * N), lwe(2 * (n + 1)),
            tv(N), acc(2 * (s.k + 1) * N), ext(2 * (s.k * N + 1)), sk(s.k * N), body(2 * N);
        fill(bsk); fill(glwe); fill(lwe);
        for (auto& x : tv) x = (u32)(gen() & 3u);
        for (auto& x : sk) x = (u32)(gen() & 1u);
        std::vector<u64> spec(bsk.size() * parts);
        if (emu_bsk_prepare(field, s.logn, s.g, n * R * (s.k + 1), bsk.data(), spec.data())) return 2;
        if (emu_external_product(field, s.g, s.k, s.logn, s.log_base, s.levels, spec.data(), glwe.data(), out.data())) return 3;
        if (emu_blind_rotate(field, s.g, (u32)n, s.k, s.logn, 2, 1, s.log_base, s.levels, 2, lwe.data(), tv.data(), 0,
                             spec.data(), acc.data(), ext.data())) return 4;
        // two samples per team (the complex transform's kernels at N = 512 with k = 2 and at N = 2048): odd batch of 3,
        // per-sample test vectors
        if (field == 5 && exb == 1 && (s.logn == 11 || (s.logn == 9 && s.k == 2))) {
          std::vector<u32> lwe3(3 * (n + 1)), tv3(3 * N), acc3(3 * (s.k + 1) * N), ext3(3 * (s.k * N + 1));
          fill(lwe3);
          for (auto& x : tv3) x = (u32)(gen() & 3u);
          emu_set_samples_per_team(2);
          const int rc = emu_blind_rotate(field, s.g, (u32)n, s.k, s.logn, 2, 1, s.log_base, s.levels, 3, lwe3.data(), tv3.data(), N,
                                          spec.data(), acc3.data(), ext3.data());
          emu_set_samples_per_team(1);
          if (rc) return 6;
          std::printf("ok two samples per team logn=%d g=%d k=%u\n", s.logn, s.g, s.k);
        }
        std::vector<u32> rows(2 * (s.k + 1) * N);
        fill(rows);
        if (emu_glwe_body(field, s.logn, s.g, s.k, 2, rows.data(), sk.data(), body.data(), 0)) return 5;
        std::printf("ok exb=%d logn=%d g=%d k=%u field=%d\n", exb, s.logn, s.g, s.k, field);
      }
    }
  }
  std::puts("sanitized run clean");
  return 0;
}

// fma52_probe.hip — VERDICT r02 next #3(a): would a 5 x 52-bit field product on v_fma_f64 beat the 9 x 29-bit one on
// v_mad_u64_u32?  Measured, not guessed: the same number of waves runs (1) fe29_mul, the shipped product (81 + 17
// multiply-adds + column splits, 150 instructions), and (2) the CORE of a 5 x 52 product in double precision - only the 25
// limb products split into high and low halves (two FMAs with round-toward-zero and one subtraction each, Emmart /
// Zheng / Weems 2018) and their accumulation into the nine 64-bit columns: no carry propagation, no reduction mod p, no
// conversion back to doubles (another ~100 instructions by count).  If (2) alone is not clearly faster than (1) whole,
// the FMA route is dead.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I secp256k1_voi_amd/csrc tools/fma52_probe.hip -o /tmp/fma52_probe && /tmp/fma52_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdint>

#include "fe29.h"
using namespace s2k;

constexpr int ITER = 2000;

__global__ void __launch_bounds__(256) k_fe29(uint32_t* out, uint32_t seed) {
  fe29 a, b;
  for (int i = 0; i < 9; ++i) {
    a.n[i] = (seed * 2654435761u + threadIdx.x * 40503u + i * 977u) & F29_M;
    b.n[i] = (seed * 40503u + blockIdx.x * 2654435761u + i * 31u) & F29_M;
  }
  a.n[8] &= F29_M8;
  b.n[8] &= F29_M8;
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) a = fe29_mul(a, b);
  uint32_t x = 0;
  for (int i = 0; i < 9; ++i) x ^= a.n[i];
  out[blockIdx.x * 256 + threadIdx.x] = x;
}

// core of the 5 x 52 product: columns c[0..9] (biased 64-bit integers) from limbs a[0..4], b[0..4] (doubles holding
// integers below 2^52); rounding mode of the wave: toward zero
__device__ __forceinline__ void fma52_core(const double a[5], const double b[5], uint64_t c[10]) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
#pragma unroll
  for (int k = 0; k < 10; ++k) c[k] = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const double hi = __builtin_fma(a[i], b[j], C1);        // 2^104 + floor(ab / 2^52) 2^52: the mantissa IS the high half
      const double sub = C2 - hi;
      const double lo = __builtin_fma(a[i], b[j], sub);       // 2^52 + (ab mod 2^52): the mantissa IS the low half
      c[i + j + 1] += (uint64_t)__double_as_longlong(hi);
      c[i + j] += (uint64_t)__double_as_longlong(lo);
    }
}
__global__ void __launch_bounds__(256) k_fma52(uint32_t* out, uint32_t seed) {
  __builtin_amdgcn_s_setreg(1 | (2 << 6) | (1 << 11), 3);     // MODE.FP_ROUND[3:2] (f64): round toward zero
  double a[5], b[5];
  for (int i = 0; i < 5; ++i) {
    a[i] = (double)((uint64_t)(seed * 2654435761u + threadIdx.x * 40503u + i) << 19 | 12345u);
    b[i] = (double)((uint64_t)(seed * 40503u + blockIdx.x * 2654435761u + i) << 19 | 54321u);
  }
  uint64_t acc = 0;
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
    uint64_t c[10];
    fma52_core(a, b, c);
    // feed something of every column back so that nothing is dead code; (no carries, no reduction: see the header)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint64_t m = (c[i] ^ c[i + 5]) & ((1ull << 52) - 1);
      a[i] = __longlong_as_double((long long)(m | 0x4330000000000000ull)) - 0x1p52;   // integer below 2^52 -> double
      acc += c[i];
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)acc ^ (uint32_t)(acc >> 32);
}

int main() {
  uint32_t* out;
  const int blocks = 256 * 4 * 3;             // three waves per SIMD, like k_verify_fast
  hipMalloc(&out, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int which = 0; which < 2; ++which) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (which == 0) k_fe29<<<blocks, 256>>>(out, rep); else k_fma52<<<blocks, 256>>>(out, rep);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (rep && ms < best) best = ms;
    }
    const double prods = (double)blocks * 256 * ITER;
    printf("%-44s %8.3f ms  %7.1f G products/s  %6.1f cycles per product per SIMD at 2.1 GHz\n",
           which == 0 ? "fe29_mul (whole product, 9 x 29, v_mad_u64_u32)" : "5 x 52 on v_fma_f64: limb products + columns ONLY", best,
           prods / best / 1e6, best * 1e-3 * 2.1e9 * 1024 * 64 / prods);
  }
  return 0;
}

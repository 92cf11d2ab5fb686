// fe26.h — GF(p), p = 2^256 - 2^32 - 977, as 10 limbs of 26 bits with lazy carries.
//
// Fast-path replacement for the reference's fiat Montgomery field
// (internal/fiat/secp256k1montgomery/secp256k1montgomery.go:87,418,750,802,844) and for the
// saturated 8x32 form in fe.h.  Chosen from measurements on MI355X
// (profiles/r01_valu_instruction_rates.txt, profiles/r01_a_*): v_mad_u64_u32 issues at the
// full VALU rate, plain 32-bit add/and/shift at twice that, while every add-with-carry
// costs a full-rate instruction plus two wait states on VCC.  With 26-bit limbs
//   * a product is 100 v_mad_u64_u32 accumulating ten 64-bit column sums with no carry
//     instruction at all, plus ~70 shift/mask/multiply-by-constant steps of reduction;
//   * add / negate are 10 independent 32-bit adds, no carry chain, no VCC.
// The algorithm is the classic 10x26 one (as in libsecp256k1's field_10x26, restated here
// for HIP): 2^260 = 2^4 * (2^32 + 977) = R1 * 2^26 + R0 with R0 = 0x3D10, R1 = 0x400.
//
// Magnitude discipline (same convention as that field): a value has magnitude m when its
// limbs 0..8 are <= 2m * 0x3FFFFFF and limb 9 <= 2m * 0x03FFFFF.  fe26_mul / fe26_sqr accept
// magnitudes <= 8 and return magnitude 1; add sums magnitudes; negate(a, m) needs
// magnitude(a) <= m and returns m + 1; mul_int multiplies it; half gives m/2 + 1.  The
// formulas in pt26.h carry the magnitude of every intermediate in comments and
// tests/test_magnitudes.py re-derives the bounds.  (The verification ladder itself runs on the
// 9x29 variant of this field, fe29.h.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fe.h"

namespace s2k {

struct fe26 {
  uint32_t n[10];
};

constexpr uint32_t F26_M = 0x3FFFFFFu;
constexpr uint32_t F26_R0 = 0x3D10u;
constexpr uint32_t F26_R1 = 0x400u;
// p in 26-bit limbs
constexpr uint32_t F26_P0 = 0x3FFFC2Fu, F26_P1 = 0x3FFFFBFu, F26_PM = 0x3FFFFFFu, F26_P9 = 0x03FFFFFu;

S2K_DEV fe26 fe26_zero() {
  fe26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = 0;
  return r;
}
S2K_DEV fe26 fe26_one() {
  fe26 r = fe26_zero();
  r.n[0] = 1;
  return r;
}

// 8 x 32-bit little-endian words (value < 2^256) -> magnitude-1 limbs
S2K_DEV fe26 fe26_from_words(const uint32_t w[8]) {
  fe26 r;
  r.n[0] = w[0] & F26_M;
  r.n[1] = ((w[0] >> 26) | (w[1] << 6)) & F26_M;
  r.n[2] = ((w[1] >> 20) | (w[2] << 12)) & F26_M;
  r.n[3] = ((w[2] >> 14) | (w[3] << 18)) & F26_M;
  r.n[4] = ((w[3] >> 8) | (w[4] << 24)) & F26_M;
  r.n[5] = (w[4] >> 2) & F26_M;
  r.n[6] = ((w[4] >> 28) | (w[5] << 4)) & F26_M;
  r.n[7] = ((w[5] >> 22) | (w[6] << 10)) & F26_M;
  r.n[8] = ((w[6] >> 16) | (w[7] << 16)) & F26_M;
  r.n[9] = w[7] >> 10;
  return r;
}

// canonical (fully normalised) limbs -> 8 x 32-bit little-endian words
S2K_DEV void fe26_to_words(uint32_t w[8], const fe26& a) {
  w[0] = a.n[0] | (a.n[1] << 26);
  w[1] = (a.n[1] >> 6) | (a.n[2] << 20);
  w[2] = (a.n[2] >> 12) | (a.n[3] << 14);
  w[3] = (a.n[3] >> 18) | (a.n[4] << 8);
  w[4] = (a.n[4] >> 24) | (a.n[5] << 2) | (a.n[6] << 28);
  w[5] = (a.n[6] >> 4) | (a.n[7] << 22);
  w[6] = (a.n[7] >> 10) | (a.n[8] << 16);
  w[7] = (a.n[8] >> 16) | (a.n[9] << 10);
}

S2K_DEV fe26 fe26_add(const fe26& a, const fe26& b) {
  fe26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] + b.n[i];
  return r;
}
// -a for magnitude(a) <= m; result magnitude m + 1
S2K_DEV fe26 fe26_negate(const fe26& a, uint32_t m) {
  fe26 r;
  const uint32_t k = 2 * (m + 1);
  r.n[0] = F26_P0 * k - a.n[0];
  r.n[1] = F26_P1 * k - a.n[1];
#pragma unroll
  for (int i = 2; i < 9; ++i) r.n[i] = F26_PM * k - a.n[i];
  r.n[9] = F26_P9 * k - a.n[9];
  return r;
}
// a - b for magnitude(b) <= mb; result magnitude(a) + mb + 1
S2K_DEV fe26 fe26_sub(const fe26& a, const fe26& b, uint32_t mb) { return fe26_add(a, fe26_negate(b, mb)); }
S2K_DEV fe26 fe26_mul_int(const fe26& a, uint32_t k) {
  fe26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] * k;
  return r;
}
// a / 2; magnitude m -> m/2 + 1
S2K_DEV fe26 fe26_half(const fe26& a) {
  uint32_t mask = (0u - (a.n[0] & 1u)) >> 6;   // 0x3FFFFFF when odd
  uint32_t t[10];
  t[0] = a.n[0] + (F26_P0 & mask);
  t[1] = a.n[1] + (F26_P1 & mask);
#pragma unroll
  for (int i = 2; i < 9; ++i) t[i] = a.n[i] + mask;
  t[9] = a.n[9] + (mask >> 4);
  fe26 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = (t[i] >> 1) + ((t[i + 1] & 1u) << 25);
  r.n[9] = t[9] >> 1;
  return r;
}
S2K_DEV fe26 fe26_select(bool pick_b, const fe26& a, const fe26& b) {
  // arithmetic select (no v_cndmask on VCC): mask = pick_b ? ~0 : 0
  uint32_t m = 0u - (uint32_t)pick_b;
  fe26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = a.n[i] ^ ((a.n[i] ^ b.n[i]) & m);
  return r;
}

// carry-propagate to magnitude 1 (not canonical); any magnitude whose limbs fit 32 bits
S2K_DEV fe26 fe26_normalize_weak(const fe26& a) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = a.n[i];
  uint32_t x = t[9] >> 22;
  t[9] &= 0x03FFFFFu;
  t[0] += x * 0x3D1u;
  t[1] += x << 6;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    t[i + 1] += t[i] >> 26;
    t[i] &= F26_M;
  }
  fe26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = t[i];
  return r;
}
// canonical representative in [0, p)
S2K_DEV fe26 fe26_normalize(const fe26& a) {
  fe26 r = fe26_normalize_weak(a);
  // after the weak pass limb 9 may still carry one bit above 2^22: fold once more
  uint32_t x = r.n[9] >> 22;
  // r >= p ?  (all middle limbs saturated, limb 1 and limb 0 above p's)
  uint32_t m = r.n[2] & r.n[3] & r.n[4] & r.n[5] & r.n[6] & r.n[7] & r.n[8];
  uint32_t ge = (r.n[9] == 0x03FFFFFu) & (m == F26_M) & ((r.n[1] + 0x40u + ((r.n[0] + 0x3D1u) >> 26)) > F26_M);
  x |= ge;
  // add x * (2^256 - p) and drop bit 256
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = r.n[i];
  t[0] += x * 0x3D1u;
  t[1] += x << 6;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    t[i + 1] += t[i] >> 26;
    t[i] &= F26_M;
  }
  t[9] &= 0x03FFFFFu;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = t[i];
  return r;
}
S2K_DEV bool fe26_is_zero(const fe26& a) {
  fe26 r = fe26_normalize(a);
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) x |= r.n[i];
  return x == 0;
}
// a == b for magnitude(b) <= 1
S2K_DEV bool fe26_eq(const fe26& a, const fe26& b) { return fe26_is_zero(fe26_sub(a, b, 1)); }

// acc += a * k as one v_mad_u64_u32 with a wave-uniform multiplier (SGPR).  The products of a
// column are kept as ONE serial accumulation chain in hand-written asm (fe26_mul_gen.h): left
// to itself hipcc builds several partial sums per column and joins them with extra 64-bit adds
// (+25 % instructions, +50 VGPRs).  The carry-out goes to VCC and is never read (the bounds
// in the header guarantee no 64-bit overflow), so there is no SGPR hazard to pad.
S2K_DEV void mad64s(uint64_t& acc, uint32_t a, uint32_t k) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(k) : "vcc");
}

// Common tail of mul / sqr.  On entry: t[0..8] are 26-bit digits, t[9] the digit of column 9,
// c the carry into column 9 (< 2^39), d the carry out of column 18 (< 2^27, weight 2^(26*19)).
// Everything at or above bit 256 is folded back with 2^256 = 0x3D1 + 0x40 * 2^26, using only
// 32x32 products (the 40-bit fold count is kept as c_lo + d * 2^14).
S2K_DEV fe26 fe26_mul_tail(uint32_t t[10], uint64_t c, uint64_t d) {
  const uint32_t d32 = (uint32_t)d;
  c += t[9];
  mad64s(c, d32, F26_R0);
  fe26 r;
  r.n[9] = (uint32_t)c & (F26_M >> 4);
  c >>= 22;                                   // < 2^19
  const uint32_t clo = (uint32_t)c;           // fold count = clo + d32 * (R1 << 4) = clo + d32 * 2^14
  const uint32_t k0 = F26_R0 >> 4, k0s = (F26_R0 >> 4) << 14, k1 = F26_R1 >> 4, k1s = (F26_R1 >> 4) << 14;
  uint64_t e = t[0];
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0"
      : "+&v"(e) : "v"(clo), "s"(k0), "v"(d32), "s"(k0s) : "vcc");
  r.n[0] = (uint32_t)e & F26_M;
  e >>= 26;
  e += t[1];
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0"
      : "+&v"(e) : "v"(clo), "s"(k1), "v"(d32), "s"(k1s) : "vcc");
  r.n[1] = (uint32_t)e & F26_M;
  e >>= 26;
  r.n[2] = t[2] + (uint32_t)e;
#pragma unroll
  for (int i = 3; i < 9; ++i) r.n[i] = t[i];
  return r;
}

// fe26_mul, fe26_sqr: column sums with 64-bit accumulators; inputs of magnitude <= 8, output
// magnitude 1.  Generated, fully unrolled.
#include "fe26_mul_gen.h"

S2K_DEV fe26 fe26_sqr_n(fe26 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe26_sqr(a);
  return a;
}
// x^(2^223 - 1), shared prefix of the inversion and square-root chains ([1] in, [1] out);
// also returns x^(2^22 - 1) and x^(2^2 - 1)
S2K_DEV fe26 fe26_pow_x223(const fe26& a, fe26& x22, fe26& x2) {
  x2 = fe26_mul(fe26_sqr(a), a);
  fe26 x3 = fe26_mul(fe26_sqr(x2), a);
  fe26 x6 = fe26_mul(fe26_sqr_n(x3, 3), x3);
  fe26 x9 = fe26_mul(fe26_sqr_n(x6, 3), x3);
  fe26 x11 = fe26_mul(fe26_sqr_n(x9, 2), x2);
  x22 = fe26_mul(fe26_sqr_n(x11, 11), x11);
  fe26 x44 = fe26_mul(fe26_sqr_n(x22, 22), x22);
  fe26 x88 = fe26_mul(fe26_sqr_n(x44, 44), x44);
  fe26 x176 = fe26_mul(fe26_sqr_n(x88, 88), x88);
  fe26 x220 = fe26_mul(fe26_sqr_n(x176, 44), x44);
  return fe26_mul(fe26_sqr_n(x220, 3), x3);
}
// a^(p-2) (Invert, internal/field/field_invert.go:11; 0 -> 0).  p - 2 = 2^256 - 2^32 - 979:
// 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1  (binary tail ...101101)
__device__ __noinline__ fe26 fe26_inv(fe26 a) {
  fe26 x22, x2;
  fe26 x223 = fe26_pow_x223(a, x22, x2);
  fe26 t = fe26_mul(fe26_sqr_n(x223, 23), x22);
  t = fe26_mul(fe26_sqr_n(t, 5), a);
  t = fe26_mul(fe26_sqr_n(t, 3), x2);
  return fe26_mul(fe26_sqr_n(t, 2), a);
}
// square root for p = 3 (mod 4): a^((p+1)/4), verified by squaring (Sqrt,
// internal/field/field_sqrt_ratio.go:14).  `a` of magnitude 1.  Returns false when no root exists.
__device__ __noinline__ bool fe26_sqrt(fe26& out, fe26 a) {
  fe26 x22, x2;
  fe26 x223 = fe26_pow_x223(a, x22, x2);
  fe26 t = fe26_mul(fe26_sqr_n(x223, 23), x22);
  t = fe26_mul(fe26_sqr_n(t, 6), x2);
  t = fe26_sqr_n(t, 2);
  out = t;
  return fe26_eq(fe26_sqr(t), a);
}

}  // namespace s2k

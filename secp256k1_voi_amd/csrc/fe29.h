// fe29.h — GF(p), p = 2^256 - 2^32 - 977, as 9 limbs of 29 bits with lazy carries: the field of
// the verification ladder (k_verify_fast).
//
// Same construction as fe26.h (v_mad_u64_u32 column chains, no carry instructions, replacement
// for the reference's fiat field, internal/fiat/secp256k1montgomery/secp256k1montgomery.go:87,418)
// with one limb fewer: a product is 81 + 17 multiply-adds instead of 100 + 19 and every linear
// operation touches 9 words instead of 10 (about 15 % fewer VALU cycles per multiplication,
// measured in profiles/).  2^261 = 2^5 (2^32 + 977) = R1 * 2^29 + R0, R0 = 0x7A20, R1 = 0x100.
//
// The price is headroom.  Bounds are counted in UNITS: a value "has w units" when its limbs
// 0..7 are <= w * 2^29 and limb 8 <= w * (2^24 + 16).
//   * fe29_mul / fe29_sqr / fused forms: a column sums at most 8 full-size products, so the
//     products of the operands' units (summed over the fused terms) must stay <= 7.8; the
//     result has 1 unit (limb 2 may exceed 2^29 by a few thousand).
//   * add sums units; negate(a, w) needs units(a) <= w and gives w + 1; mul_int multiplies;
//     half gives w/2 + 1/2; normalize_weak brings anything whose limbs fit 32 bits to 1 unit.
// jacobian29.h states the units of every intermediate and tests/test_fe29_model.py re-derives
// them with interval arithmetic (and checks the schedule itself on integers).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fe.h"
#include "fe26.h"

namespace s2k {

struct fe29 {
  uint32_t n[9];
};

constexpr uint32_t F29_M = 0x1FFFFFFFu;
constexpr uint32_t F29_M8 = 0x00FFFFFFu;   // nominal width of limb 8: 256 - 8 * 29 = 24 bits
constexpr uint32_t F29_R0 = 0x7A20u;
constexpr uint32_t F29_R1 = 0x100u;
// p in 29-bit limbs
constexpr uint32_t F29_P0 = 0x1FFFFC2Fu, F29_P1 = 0x1FFFFFF7u, F29_PM = 0x1FFFFFFFu, F29_P8 = 0x00FFFFFFu;

S2K_DEV fe29 fe29_zero() {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = 0;
  return r;
}
S2K_DEV fe29 fe29_one() {
  fe29 r = fe29_zero();
  r.n[0] = 1;
  return r;
}

S2K_DEV fe29 fe29_add(const fe29& a, const fe29& b) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = a.n[i] + b.n[i];
  return r;
}
// -a for units(a) <= w; result w + 1 units
S2K_DEV fe29 fe29_negate(const fe29& a, uint32_t w) {
  fe29 r;
  const uint32_t k = w + 1;
  r.n[0] = F29_P0 * k - a.n[0];
  r.n[1] = F29_P1 * k - a.n[1];
#pragma unroll
  for (int i = 2; i < 8; ++i) r.n[i] = F29_PM * k - a.n[i];
  r.n[8] = F29_P8 * k - a.n[8];
  return r;
}
S2K_DEV fe29 fe29_mul_int(const fe29& a, uint32_t k) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = a.n[i] * k;
  return r;
}
// a / 2; w units -> w/2 + 1/2 (+1 bit)
S2K_DEV fe29 fe29_half(const fe29& a) {
  uint32_t mask = 0u - (a.n[0] & 1u);   // all ones when odd: add p first
  uint32_t t[9];
  t[0] = a.n[0] + (F29_P0 & mask);
  t[1] = a.n[1] + (F29_P1 & mask);
#pragma unroll
  for (int i = 2; i < 8; ++i) t[i] = a.n[i] + (F29_PM & mask);
  t[8] = a.n[8] + (F29_P8 & mask);
  fe29 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.n[i] = (t[i] >> 1) + ((t[i + 1] & 1u) << 28);
  r.n[8] = t[8] >> 1;
  return r;
}
S2K_DEV fe29 fe29_select(bool pick_b, const fe29& a, const fe29& b) {
  uint32_t m = 0u - (uint32_t)pick_b;   // arithmetic select: v_cndmask on VCC is slow on gfx950 (profiles/r01_valu_instruction_rates.txt)
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = a.n[i] ^ ((a.n[i] ^ b.n[i]) & m);
  return r;
}

// neg ? 2p - a : a, for a of 1 unit (result <= 2 units), without a select: with M = all ones when
// neg, (a ^ M) + (M & (B + 1)) is ~a + B + 1 = B - a limb by limb (B = 2p's limbs), and a when M = 0
S2K_DEV fe29 fe29_cond_negate1(const fe29& a, bool neg) {
  const uint32_t M = 0u - (uint32_t)neg;
  const uint32_t b0 = M & (2u * F29_P0 + 1u), b1 = M & (2u * F29_P1 + 1u), bm = M & (2u * F29_PM + 1u), b8 = M & (2u * F29_P8 + 1u);
  fe29 r;
  r.n[0] = (a.n[0] ^ M) + b0;
  r.n[1] = (a.n[1] ^ M) + b1;
#pragma unroll
  for (int i = 2; i < 8; ++i) r.n[i] = (a.n[i] ^ M) + bm;
  r.n[8] = (a.n[8] ^ M) + b8;
  return r;
}

// carry-propagate to 1 unit (not canonical); input limbs < 2^32 - 2^18
S2K_DEV fe29 fe29_normalize_weak(const fe29& a) {
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = a.n[i];
  uint32_t x = t[8] >> 24;             // multiples of 2^256 = 977 + 8 * 2^29 (mod p)
  t[8] &= F29_M8;
  t[0] += x * 0x3D1u;
  t[1] += x << 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t[i + 1] += t[i] >> 29;
    t[i] &= F29_M;
  }
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = t[i];
  return r;
}
// canonical representative in [0, p)
S2K_DEV fe29 fe29_normalize(const fe29& a) {
  fe29 r = fe29_normalize_weak(a);
  uint32_t x = r.n[8] >> 24;           // the weak pass can leave one bit above 2^24
  uint32_t m = r.n[2] & r.n[3] & r.n[4] & r.n[5] & r.n[6] & r.n[7];
  uint32_t ge = (r.n[8] == F29_M8) & (m == F29_M) & ((r.n[1] + 8u + ((r.n[0] + 0x3D1u) >> 29)) > F29_M);
  x |= ge;
  uint32_t t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = r.n[i];
  t[0] += x * 0x3D1u;                  // add x * (2^256 - p) and drop bit 256
  t[1] += x << 3;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    t[i + 1] += t[i] >> 29;
    t[i] &= F29_M;
  }
  t[8] &= F29_M8;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = t[i];
  return r;
}
S2K_DEV bool fe29_is_zero(const fe29& a) {
  fe29 r = fe29_normalize(a);
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) x |= r.n[i];
  return x == 0;
}
// a == b for units(b) <= 1
S2K_DEV bool fe29_eq(const fe29& a, const fe29& b) { return fe29_is_zero(fe29_add(a, fe29_negate(b, 1))); }

// Common tail of the products.  On entry t[0..7] are 29-bit digits, t[8] the digit of column 8,
// c the carry into column 8, d the carry out of column 16 (weight 2^(29*17) = 2^232 * 2^261).
// Everything at or above bit 256 goes back in with 2^256 = 0x3D1 + 8 * 2^29, using 32x32
// products only (the fold count is c_lo + d * R1 * 2^5 = c_lo + d * 2^13).
S2K_DEV fe29 fe29_mul_tail(uint32_t t[9], uint64_t c, uint64_t d) {
  const uint32_t d32 = (uint32_t)d;
  c += t[8];
  mad64s(c, d32, F29_R0);
  fe29 r;
  r.n[8] = (uint32_t)c & F29_M8;
  c >>= 24;
  const uint32_t clo = (uint32_t)c;
  const uint32_t k0 = F29_R0 >> 5, k0s = (F29_R0 >> 5) << 13, k1 = F29_R1 >> 5, k1s = (F29_R1 >> 5) << 13;
  uint64_t e = t[0];
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0"
      : "+&v"(e) : "v"(clo), "s"(k0), "v"(d32), "s"(k0s) : "vcc");
  r.n[0] = (uint32_t)e & F29_M;
  e >>= 29;
  e += t[1];
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %3, %4, %0"
      : "+&v"(e) : "v"(clo), "s"(k1), "v"(d32), "s"(k1s) : "vcc");
  r.n[1] = (uint32_t)e & F29_M;
  e >>= 29;
  r.n[2] = t[2] + (uint32_t)e;
#pragma unroll
  for (int i = 3; i < 8; ++i) r.n[i] = t[i];
  return r;
}

// fe29_from_words / fe29_to_words, fe29_mul, fe29_sqr, fe29_mul_add_mul, fe29_mul_add_sqr
#include "fe29_mul_gen.h"

S2K_DEV fe29 fe29_sqr_n(fe29 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe29_sqr(a);
  return a;
}
// x^(2^223 - 1) (1 unit in and out), with x^(2^22 - 1) and x^(2^2 - 1); same chain as the reference field_invert.go
S2K_DEV fe29 fe29_pow_x223(const fe29& a, fe29& x22, fe29& x2) {
  x2 = fe29_mul(fe29_sqr(a), a);
  fe29 x3 = fe29_mul(fe29_sqr(x2), a);
  fe29 x6 = fe29_mul(fe29_sqr_n(x3, 3), x3);
  fe29 x9 = fe29_mul(fe29_sqr_n(x6, 3), x3);
  fe29 x11 = fe29_mul(fe29_sqr_n(x9, 2), x2);
  x22 = fe29_mul(fe29_sqr_n(x11, 11), x11);
  fe29 x44 = fe29_mul(fe29_sqr_n(x22, 22), x22);
  fe29 x88 = fe29_mul(fe29_sqr_n(x44, 44), x44);
  fe29 x176 = fe29_mul(fe29_sqr_n(x88, 88), x88);
  fe29 x220 = fe29_mul(fe29_sqr_n(x176, 44), x44);
  return fe29_mul(fe29_sqr_n(x220, 3), x3);
}
// a^(p-2) (Invert, internal/field/field_invert.go:11; 0 -> 0)
__device__ __noinline__ fe29 fe29_inv(fe29 a) {
  fe29 x22, x2;
  fe29 x223 = fe29_pow_x223(a, x22, x2);
  fe29 t = fe29_mul(fe29_sqr_n(x223, 23), x22);
  t = fe29_mul(fe29_sqr_n(t, 5), a);
  t = fe29_mul(fe29_sqr_n(t, 3), x2);
  return fe29_mul(fe29_sqr_n(t, 2), a);
}
// a^((p+1)/4), checked by squaring (Sqrt, internal/field/field_sqrt_ratio.go:14); a of 1 unit
__device__ __noinline__ bool fe29_sqrt(fe29& out, fe29 a) {
  fe29 x22, x2;
  fe29 x223 = fe29_pow_x223(a, x22, x2);
  fe29 t = fe29_mul(fe29_sqr_n(x223, 23), x22);
  t = fe29_mul(fe29_sqr_n(t, 6), x2);
  t = fe29_sqr_n(t, 2);
  out = t;
  return fe29_eq(fe29_sqr(t), a);
}

}  // namespace s2k

// fe.h — GF(p) arithmetic for p = 2^256 - 2^32 - 977 on gfx950 (CDNA4), one element per lane.
//
// Replaces, for the device path, the reference's fiat 4x64 Montgomery code
// (internal/fiat/secp256k1montgomery/secp256k1montgomery.go:87 Mul, :418 Square, :750 Add,
// :802 Sub, :844 Opp) and the field.Element wrapper (internal/field/field.go:41-197).
// Only canonical values have to agree with the reference (SURVEY.md §8a "parity
// definition"), so the representation is chosen for the hardware:
//
//   * 8 x 32-bit saturated limbs in VGPRs, little-endian, NOT in the Montgomery domain:
//     p is a pseudo-Mersenne prime, 2^256 = C (mod p) with C = 2^32 + 977, so a 512-bit
//     product folds with 8 extra multiply-adds instead of a second 8x8 pass.
//   * "weak" reduction: every fe holds some 256-bit representative (0 <= v < 2^256) of its
//     residue; fe_normalize() produces the canonical one.  All routines accept any
//     representative.
//   * measured on MI355X (tools/valu_rates.hip, gpurun_out/valu_rates_r01.txt):
//     v_mad_u64_u32, v_addc_co_u32, v_mul_hi_u32 all issue at the full VALU rate once
//     two waves share a SIMD, so the cost model is "count instructions", and carry
//     handling costs as much as the multiplies.
//   * gfx950 hazard: a VALU that reads an SGPR/VCC written by the previous VALU needs two
//     wait states.  Hand-written asm blocks below keep one independent instruction (or an
//     s_nop 1) between a carry producer and its consumer; carry chains written with
//     __builtin_addc are padded by the compiler.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define S2K_DEV __device__ __forceinline__

namespace s2k {

struct fe {
  uint32_t v[8];
};

// p, little-endian limbs
__device__ static const uint32_t FE_P[8] = {0xFFFFFC2Fu, 0xFFFFFFFEu, 0xFFFFFFFFu, 0xFFFFFFFFu,
                                            0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};

S2K_DEV fe fe_zero() {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
S2K_DEV fe fe_from_u32(uint32_t x) {
  fe r = fe_zero();
  r.v[0] = x;
  return r;
}

// ---- raw 256-bit add / sub with carry / borrow out -------------------------------
S2K_DEV uint32_t u256_add(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = __builtin_addc(a[i], b[i], c, &c);
  return c;
}
S2K_DEV uint32_t u256_sub(uint32_t r[8], const uint32_t a[8], const uint32_t b[8]) {
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = __builtin_subc(a[i], b[i], c, &c);
  return c;
}
// r += m * C  (m in {0,1}), returns carry out.  C = 2^32 + 977.
S2K_DEV uint32_t u256_add_mC(uint32_t r[8], uint32_t m) {
  unsigned c = 0;
  uint32_t k0 = m ? 977u : 0u, k1 = m;
  r[0] = __builtin_addc(r[0], k0, c, &c);
  r[1] = __builtin_addc(r[1], k1, c, &c);
#pragma unroll
  for (int i = 2; i < 8; ++i) r[i] = __builtin_addc(r[i], 0u, c, &c);
  return c;
}
S2K_DEV uint32_t u256_sub_mC(uint32_t r[8], uint32_t m) {
  unsigned c = 0;
  uint32_t k0 = m ? 977u : 0u, k1 = m;
  r[0] = __builtin_subc(r[0], k0, c, &c);
  r[1] = __builtin_subc(r[1], k1, c, &c);
#pragma unroll
  for (int i = 2; i < 8; ++i) r[i] = __builtin_subc(r[i], 0u, c, &c);
  return c;
}

// ---- add / sub / neg (field.go:61-80; fiat Add/Sub/Opp) --------------------------
// a + b = s + c*2^256 = s + c*C.  A second wrap needs s >= 2^256 - C after the first,
// i.e. both inputs non-canonical; it is handled (the result is then < C, no third wrap).
S2K_DEV fe fe_add(const fe& a, const fe& b) {
  fe r;
  uint32_t c = u256_add(r.v, a.v, b.v);
  uint32_t c2 = u256_add_mC(r.v, c);
  u256_add_mC(r.v, c2);
  return r;
}
S2K_DEV fe fe_sub(const fe& a, const fe& b) {
  fe r;
  uint32_t c = u256_sub(r.v, a.v, b.v);
  uint32_t c2 = u256_sub_mC(r.v, c);
  u256_sub_mC(r.v, c2);
  return r;
}
S2K_DEV fe fe_neg(const fe& a) { return fe_sub(fe_zero(), a); }
// a / 2: (a + (a odd ? p : 0)) >> 1, computed on 257 bits
S2K_DEV fe fe_half(const fe& a) {
  uint32_t m = 0u - (a.v[0] & 1u);
  uint32_t t[8];
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = __builtin_addc(a.v[i], FE_P[i] & m, c, &c);
  fe r;
#pragma unroll
  for (int i = 0; i < 7; ++i) r.v[i] = (t[i] >> 1) | (t[i + 1] << 31);
  r.v[7] = (t[7] >> 1) | (c << 31);
  return r;
}
S2K_DEV fe fe_dbl(const fe& a) { return fe_add(a, a); }

// canonical representative in [0, p)  (reduceSaturated, field_reduce.go:82-102)
S2K_DEV fe fe_normalize(const fe& a) {
  // a >= p  <=>  a + C overflows 2^256; then a - p = a + C - 2^256.
  fe t = a;
  uint32_t c = u256_add_mC(t.v, 1u);
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = c ? t.v[i] : a.v[i];
  return r;
}
// raw 256-bit compare helpers (on representatives, not residues)
S2K_DEV bool u256_is_zero(const uint32_t a[8]) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) x |= a[i];
  return x == 0;
}
S2K_DEV bool u256_eq(const uint32_t a[8], const uint32_t b[8]) {
  uint32_t x = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) x |= a[i] ^ b[i];
  return x == 0;
}
// a < b on raw values
S2K_DEV bool u256_lt(const uint32_t a[8], const uint32_t b[8]) {
  uint32_t t[8];
  return u256_sub(t, a, b) != 0;
}
S2K_DEV bool fe_is_zero(const fe& a) {   // field.go:183
  fe n = fe_normalize(a);
  return u256_is_zero(n.v);
}
S2K_DEV bool fe_eq(const fe& a, const fe& b) {   // field.go:178
  fe x = fe_normalize(a), y = fe_normalize(b);
  return u256_eq(x.v, y.v);
}
S2K_DEV bool fe_is_odd(const fe& a) {   // field.go:191
  fe n = fe_normalize(a);
  return (n.v[0] & 1u) != 0;
}
S2K_DEV bool fe_is_canonical_raw(const uint32_t a[8]) { return u256_lt(a, FE_P); }   // field.go:128
S2K_DEV fe fe_select(bool pick_b, const fe& a, const fe& b) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = pick_b ? b.v[i] : a.v[i];
  return r;
}

// ---- multiply-accumulate blocks ---------------------------------------------------
// macN: (c2 : acc) += sum of N 32x32 products.  acc is a 64-bit VGPR pair, c2 the third
// word.  All N v_mad_u64_u32 are issued first, each with its own carry-out SGPR pair, then
// the N v_addc_co_u32: every carry consumer is at least two instructions behind its
// producer (gfx950 "VALU writes SGPR -> VALU reads it" needs two wait states), padded with
// s_nop only for N < 3.
S2K_DEV void mac1(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0) {
  asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\t"
      "s_nop 1\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2)
      : "v"(a0), "v"(b0)
      : "vcc");
}
S2K_DEV void mac2(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1) {
  uint64_t s0;
  asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t"
      "s_nop 0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1)
      : "vcc");
}
S2K_DEV void mac3(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2) {
  uint64_t s0, s1;
  asm("v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
      "v_mad_u64_u32 %0, %3, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %8, %9, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2)
      : "vcc");
}
S2K_DEV void mac4(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3) {
  uint64_t s0, s1, s2;
  asm("v_mad_u64_u32 %0, %2, %5, %6, %0\n\t"
      "v_mad_u64_u32 %0, %3, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %4, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, %4, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1), "=&s"(s2)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3)
      : "vcc");
}
S2K_DEV void mac5(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4) {
  uint64_t s0, s1, s2, s3;
  asm("v_mad_u64_u32 %0, %2, %6, %7, %0\n\t"
      "v_mad_u64_u32 %0, %3, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %4, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %5, %12, %13, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %14, %15, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, %4, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, %5, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4)
      : "vcc");
}
S2K_DEV void mac6(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5) {
  uint64_t s0, s1, s2, s3, s4;
  asm("v_mad_u64_u32 %0, %2, %7, %8, %0\n\t"
      "v_mad_u64_u32 %0, %3, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %4, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %5, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %6, %15, %16, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, %4, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, %5, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, %6, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5)
      : "vcc");
}
S2K_DEV void mac7(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6) {
  uint64_t s0, s1, s2, s3, s4, s5;
  asm("v_mad_u64_u32 %0, %2, %8, %9, %0\n\t"
      "v_mad_u64_u32 %0, %3, %10, %11, %0\n\t"
      "v_mad_u64_u32 %0, %4, %12, %13, %0\n\t"
      "v_mad_u64_u32 %0, %5, %14, %15, %0\n\t"
      "v_mad_u64_u32 %0, %6, %16, %17, %0\n\t"
      "v_mad_u64_u32 %0, %7, %18, %19, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %20, %21, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, %4, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, %5, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, %6, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, %7, 0, %1, %7\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6)
      : "vcc");
}
S2K_DEV void mac8(uint64_t& acc, uint32_t& c2, uint32_t a0, uint32_t b0, uint32_t a1, uint32_t b1, uint32_t a2, uint32_t b2, uint32_t a3, uint32_t b3, uint32_t a4, uint32_t b4, uint32_t a5, uint32_t b5, uint32_t a6, uint32_t b6, uint32_t a7, uint32_t b7) {
  uint64_t s0, s1, s2, s3, s4, s5, s6;
  asm("v_mad_u64_u32 %0, %2, %9, %10, %0\n\t"
      "v_mad_u64_u32 %0, %3, %11, %12, %0\n\t"
      "v_mad_u64_u32 %0, %4, %13, %14, %0\n\t"
      "v_mad_u64_u32 %0, %5, %15, %16, %0\n\t"
      "v_mad_u64_u32 %0, %6, %17, %18, %0\n\t"
      "v_mad_u64_u32 %0, %7, %19, %20, %0\n\t"
      "v_mad_u64_u32 %0, %8, %21, %22, %0\n\t"
      "v_mad_u64_u32 %0, vcc, %23, %24, %0\n\t"
      "v_addc_co_u32 %1, %2, 0, %1, %2\n\t"
      "v_addc_co_u32 %1, %3, 0, %1, %3\n\t"
      "v_addc_co_u32 %1, %4, 0, %1, %4\n\t"
      "v_addc_co_u32 %1, %5, 0, %1, %5\n\t"
      "v_addc_co_u32 %1, %6, 0, %1, %6\n\t"
      "v_addc_co_u32 %1, %7, 0, %1, %7\n\t"
      "v_addc_co_u32 %1, %8, 0, %1, %8\n\t"
      "v_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+&v"(acc), "+&v"(c2), "=&s"(s0), "=&s"(s1), "=&s"(s2), "=&s"(s3), "=&s"(s4), "=&s"(s5), "=&s"(s6)
      : "v"(a0), "v"(b0), "v"(a1), "v"(b1), "v"(a2), "v"(b2), "v"(a3), "v"(b3), "v"(a4), "v"(b4), "v"(a5), "v"(b5), "v"(a6), "v"(b6), "v"(a7), "v"(b7)
      : "vcc");
}

template <int N>
S2K_DEV void mac_col(uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b) {
  // products a[0]*b[0], a[1]*b[-1], ... a[N-1]*b[-(N-1)]
  if constexpr (N == 1) mac1(acc, c2, a[0], b[0]);
  if constexpr (N == 2) mac2(acc, c2, a[0], b[0], a[1], b[-1]);
  if constexpr (N == 3) mac3(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2]);
  if constexpr (N == 4) mac4(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2], a[3], b[-3]);
  if constexpr (N == 5) mac5(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2], a[3], b[-3], a[4], b[-4]);
  if constexpr (N == 6) mac6(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2], a[3], b[-3], a[4], b[-4], a[5], b[-5]);
  if constexpr (N == 7) mac7(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2], a[3], b[-3], a[4], b[-4], a[5], b[-5], a[6], b[-6]);
  if constexpr (N == 8) mac8(acc, c2, a[0], b[0], a[1], b[-1], a[2], b[-2], a[3], b[-3], a[4], b[-4], a[5], b[-5], a[6], b[-6], a[7], b[-7]);
}

// one column of the schoolbook product: sum over i + j = K
template <int K>
S2K_DEV void mul_column(uint32_t* t, uint64_t& acc, uint32_t& c2, const uint32_t* a, const uint32_t* b) {
  constexpr int lo = K < 8 ? 0 : K - 7;
  constexpr int hi = K < 8 ? K : 7;
  mac_col<hi - lo + 1>(acc, c2, a + lo, b + (K - lo));
  t[K] = (uint32_t)acc;
  acc = (acc >> 32) | ((uint64_t)c2 << 32);
  c2 = 0;
}
// 8x8 -> 16 limb schoolbook product, product scanning (column by column)
S2K_DEV void u256_mul_wide(uint32_t t[16], const uint32_t a[8], const uint32_t b[8]) {
  uint64_t acc = 0;
  uint32_t c2 = 0;
  mul_column<0>(t, acc, c2, a, b);  mul_column<1>(t, acc, c2, a, b);  mul_column<2>(t, acc, c2, a, b);
  mul_column<3>(t, acc, c2, a, b);  mul_column<4>(t, acc, c2, a, b);  mul_column<5>(t, acc, c2, a, b);
  mul_column<6>(t, acc, c2, a, b);  mul_column<7>(t, acc, c2, a, b);  mul_column<8>(t, acc, c2, a, b);
  mul_column<9>(t, acc, c2, a, b);  mul_column<10>(t, acc, c2, a, b); mul_column<11>(t, acc, c2, a, b);
  mul_column<12>(t, acc, c2, a, b); mul_column<13>(t, acc, c2, a, b); mul_column<14>(t, acc, c2, a, b);
  t[15] = (uint32_t)acc;
}

// squaring column: cross products (i < j) once into a side accumulator, doubled, plus the
// diagonal term for even K
template <int K>
S2K_DEV void sqr_column(uint32_t* t, uint64_t& acc, uint32_t& c2, const uint32_t* a) {
  constexpr int lo = K < 8 ? 0 : K - 7;
  constexpr int npairs = (K + 1) / 2 - lo;
  if constexpr (npairs > 0) {
    uint64_t cacc = 0;
    uint32_t cc2 = 0;
    mac_col<npairs>(cacc, cc2, a + lo, a + (K - lo));
    uint32_t d0 = (uint32_t)cacc, d1 = (uint32_t)(cacc >> 32);
    uint32_t e0 = d0 << 1, e1 = (d1 << 1) | (d0 >> 31), e2 = (cc2 << 1) | (d1 >> 31);
    unsigned c = 0;
    uint32_t x0 = __builtin_addc((uint32_t)acc, e0, c, &c);
    uint32_t x1 = __builtin_addc((uint32_t)(acc >> 32), e1, c, &c);
    c2 = __builtin_addc(c2, e2, c, &c);
    acc = ((uint64_t)x1 << 32) | x0;
  }
  if constexpr ((K & 1) == 0) mac1(acc, c2, a[K / 2], a[K / 2]);
  t[K] = (uint32_t)acc;
  acc = (acc >> 32) | ((uint64_t)c2 << 32);
  c2 = 0;
}
S2K_DEV void u256_sqr_wide(uint32_t t[16], const uint32_t a[8]) {
  uint64_t acc = 0;
  uint32_t c2 = 0;
  sqr_column<0>(t, acc, c2, a);  sqr_column<1>(t, acc, c2, a);  sqr_column<2>(t, acc, c2, a);
  sqr_column<3>(t, acc, c2, a);  sqr_column<4>(t, acc, c2, a);  sqr_column<5>(t, acc, c2, a);
  sqr_column<6>(t, acc, c2, a);  sqr_column<7>(t, acc, c2, a);  sqr_column<8>(t, acc, c2, a);
  sqr_column<9>(t, acc, c2, a);  sqr_column<10>(t, acc, c2, a); sqr_column<11>(t, acc, c2, a);
  sqr_column<12>(t, acc, c2, a); sqr_column<13>(t, acc, c2, a); sqr_column<14>(t, acc, c2, a);
  t[15] = (uint32_t)acc;
}

// fold a 512-bit value to a 256-bit representative: L + H*C, twice.
S2K_DEV fe fe_reduce_wide(const uint32_t t[16]) {
  // U = H * 977  (9 limbs)
  uint32_t u[9];
  uint64_t m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m = (uint64_t)t[8 + i] * 977u + (m >> 32);
    u[i] = (uint32_t)m;
  }
  u[8] = (uint32_t)(m >> 32);
  // r = L + U[0..7]  (carry c1), then r += H << 32 (limbs 1..7, carry c2)
  fe r;
  unsigned c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = __builtin_addc(t[i], u[i], c, &c);
  uint32_t top = u[8] + c;                       // weight 2^256, < 2^11
  c = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i) r.v[i] = __builtin_addc(r.v[i], t[8 + i - 1], c, &c);
  // V = top + t[15] + c  (weight 2^256), up to 33 bits
  uint64_t V = (uint64_t)top + t[15] + c;
  uint32_t v0 = (uint32_t)V, v1 = (uint32_t)(V >> 32);   // v1 in {0,1}
  // r += V * C = V*977 + (V << 32)
  uint64_t w = (uint64_t)v0 * 977u;                 // < 2^42
  uint32_t w0 = (uint32_t)w;
  uint64_t w1 = (w >> 32) + (v1 ? 977u : 0u) + v0;  // limb 1 contribution, < 2^33+...
  uint32_t w1lo = (uint32_t)w1;
  uint32_t w2 = (uint32_t)(w1 >> 32) + v1;          // limb 2 contribution
  c = 0;
  r.v[0] = __builtin_addc(r.v[0], w0, c, &c);
  r.v[1] = __builtin_addc(r.v[1], w1lo, c, &c);
  r.v[2] = __builtin_addc(r.v[2], w2, c, &c);
#pragma unroll
  for (int i = 3; i < 8; ++i) r.v[i] = __builtin_addc(r.v[i], 0u, c, &c);
  // a final wrap leaves a value < 2^67, so one more +C cannot wrap again
  u256_add_mC(r.v, c);
  return r;
}

S2K_DEV fe fe_mul(const fe& a, const fe& b) {   // field.go:82 Multiply
  uint32_t t[16];
  u256_mul_wide(t, a.v, b.v);
  return fe_reduce_wide(t);
}
S2K_DEV fe fe_sqr(const fe& a) {   // field.go:90 Square
  uint32_t t[16];
  u256_sqr_wide(t, a.v);
  return fe_reduce_wide(t);
}
// a * k for a small constant k (k < 2^16): used for b3 = 21 (point_projective.go:21)
S2K_DEV fe fe_mul_small(const fe& a, uint32_t k) {
  fe r;
  uint64_t m = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    m = (uint64_t)a.v[i] * k + (m >> 32);
    r.v[i] = (uint32_t)m;
  }
  uint32_t top = (uint32_t)(m >> 32);   // < 2^16, weight 2^256
  uint64_t w = (uint64_t)top * 977u;    // < 2^26
  unsigned c = 0;
  r.v[0] = __builtin_addc(r.v[0], (uint32_t)w, c, &c);
  r.v[1] = __builtin_addc(r.v[1], top, c, &c);
#pragma unroll
  for (int i = 2; i < 8; ++i) r.v[i] = __builtin_addc(r.v[i], 0u, c, &c);
  u256_add_mC(r.v, c);
  return r;
}

S2K_DEV fe fe_sqr_n(fe a, int n) {   // Pow2k, field.go:97
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fe_sqr(a);
  return a;
}

// x^(p-2): same addition chain as internal/field/field_invert.go:11-140 (255 S + 15 M).
// Invert(0) = 0.
__device__ __noinline__ fe fe_inv(fe x) {
  fe t0, t1, t2, t3, t4, t5;
  t0 = fe_sqr(x);
  t1 = fe_sqr(t0);
  t1 = fe_mul(x, t1);          // _101
  t0 = fe_mul(t0, t1);         // _111
  t2 = fe_sqr(t0);             // _1110
  t3 = fe_sqr_n(t2, 2);
  t3 = fe_mul(t0, t3);         // _111111
  t3 = fe_sqr_n(t3, 4);
  t2 = fe_mul(t2, t3);         // i13
  t3 = fe_sqr_n(t2, 2);
  t3 = fe_mul(t0, t3);         // x12
  t3 = fe_sqr_n(t3, 10);
  t2 = fe_mul(t2, t3);
  t4 = fe_mul(x, t2);          // x22
  t2 = fe_sqr(t4);             // i29
  t3 = fe_sqr_n(t2, 2);        // i31
  t5 = fe_sqr_n(t3, 22);
  t3 = fe_mul(t3, t5);         // i54
  t5 = fe_sqr_n(t3, 20);
  t2 = fe_mul(t2, t5);
  t2 = fe_sqr_n(t2, 46);
  t3 = fe_mul(t3, t2);         // i122
  t2 = fe_sqr_n(t3, 110);
  t3 = fe_mul(t3, t2);
  t0 = fe_mul(t0, t3);         // x223
  t0 = fe_sqr_n(t0, 23);
  t4 = fe_mul(t4, t0);
  t4 = fe_sqr_n(t4, 7);
  t4 = fe_mul(t1, t4);
  t4 = fe_sqr_n(t4, 3);
  return fe_mul(t1, t4);
}

// square root for p = 3 (mod 4): candidate a^((p+1)/4), checked by squaring.  Same result
// set as Element.Sqrt (internal/field/field_sqrt_ratio.go:14-63): returns (root, true) or
// (0, false); the caller fixes the parity (point_s11n.go:160-166), so which of the two
// roots is produced does not matter.  Chain for (p+1)/4 = 2^254 - 2^30 - 244:
// x223 << 23 + x22, << 6 + _11, << 2.
S2K_DEV bool fe_sqrt(fe& out, const fe& a) {
  fe x2 = fe_mul(fe_sqr(a), a);                 // 2 ones
  fe x3 = fe_mul(fe_sqr(x2), a);                // 3 ones
  fe x6 = fe_mul(fe_sqr_n(x3, 3), x3);
  fe x9 = fe_mul(fe_sqr_n(x6, 3), x3);
  fe x11 = fe_mul(fe_sqr_n(x9, 2), x2);
  fe x22 = fe_mul(fe_sqr_n(x11, 11), x11);
  fe x44 = fe_mul(fe_sqr_n(x22, 22), x22);
  fe x88 = fe_mul(fe_sqr_n(x44, 44), x44);
  fe x176 = fe_mul(fe_sqr_n(x88, 88), x88);
  fe x220 = fe_mul(fe_sqr_n(x176, 44), x44);
  fe x223 = fe_mul(fe_sqr_n(x220, 3), x3);
  fe t = fe_mul(fe_sqr_n(x223, 23), x22);
  t = fe_mul(fe_sqr_n(t, 6), x2);
  t = fe_sqr_n(t, 2);
  bool ok = fe_eq(fe_sqr(t), a);
  out = ok ? t : fe_zero();
  return ok;
}

// ---- big-endian bytes <-> limbs (internal/helpers/helpers.go:48-66) ----------------
S2K_DEV uint32_t bswap32(uint32_t x) { return __builtin_bswap32(x); }
// p points at 32 big-endian bytes, 16-byte aligned
S2K_DEV void load_be32(uint32_t out[8], const uint8_t* p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
  uint4 hi = q[0], lo = q[1];
  out[7] = bswap32(hi.x); out[6] = bswap32(hi.y); out[5] = bswap32(hi.z); out[4] = bswap32(hi.w);
  out[3] = bswap32(lo.x); out[2] = bswap32(lo.y); out[1] = bswap32(lo.z); out[0] = bswap32(lo.w);
}
S2K_DEV void store_be32(uint8_t* p, const uint32_t in[8]) {
  uint4* q = reinterpret_cast<uint4*>(p);
  q[0] = make_uint4(bswap32(in[7]), bswap32(in[6]), bswap32(in[5]), bswap32(in[4]));
  q[1] = make_uint4(bswap32(in[3]), bswap32(in[2]), bswap32(in[1]), bswap32(in[0]));
}

}  // namespace s2k

// ct_cpu.cpp — the CONSTANT-TIME twins of the reference's scalar multiplications, on the host CPU
// (SURVEY.md §8 a23 / f4).  The GPU kernels of this library are variable-time by design and serve
// public data only (signature verification); anything that touches a secret scalar — ECDH,
// signing, key generation — has to go through these functions instead, exactly as the reference
// keeps two code paths:
//   s2k_ct_scalar_mult       Point.ScalarMult        point_mul_glv.go:257-303
//   s2k_ct_scalar_base_mult  Point.ScalarBaseMult    point_mul_table.go:168-194
//   s2k_ct_ecdh              PrivateKey.ECDH         secec/secec.go:53-56
//   s2k_ct_ecdsa_sign_raw    the arithmetic of sign  secec/ecdsa.go:335-390 (nonce supplied by the caller)
// Constant time means: no branch, no memory address and no loop count depends on a scalar, a
// private key, a nonce or a coordinate derived from one.  Table lookups scan every entry under a
// mask (lookupProjectivePoint / lookupAffinePoint, point_mul_table_ref.go:11-24), selections are
// arithmetic masks, reductions always run their final conditional subtraction as a masked move.
// Arithmetic: 4 x 64-bit saturated limbs with unsigned __int128 products, pseudo-Mersenne folding
// for p = 2^256 - 0x1000003D1 and n = 2^256 - 0x14551231950B75FC4402DA1732FC9BEBF; group law =
// the complete Renes-Costello-Batina formulas (point_projective.go:24,123,208).  Pure host code:
// no HIP call, works without a GPU.
#include <cstdint>
#include <cstring>
#include <mutex>

#include "../../include/secp256k1_voi_amd.h"

namespace {

typedef unsigned __int128 u128;
typedef uint64_t u64;

// ---- masks -----------------------------------------------------------------------------------
// Every mask passes through an empty asm statement: the optimiser must not learn that it is 0 or
// all ones, or it turns the masked moves back into compares, branches and secret-indexed loads
// (clang did exactly that to the table scans before the barrier was added; tests/test_ct_cpu.py
// now counts the conditional branches of the compiled constant-time functions).
inline u64 value_barrier(u64 x) {
  asm volatile("" : "+r"(x));
  return x;
}
inline u64 mask_nonzero(u64 x) { return value_barrier(0 - ((x | (0 - x)) >> 63)); }   // all ones iff x != 0
inline u64 mask_eq(u64 a, u64 b) { return value_barrier(~mask_nonzero(a ^ b)); }

struct u256 {
  u64 v[4];   // little-endian limbs
};

inline u64 adc(u64 a, u64 b, u64& c) {
  u128 t = (u128)a + b + c;
  c = (u64)(t >> 64);
  return (u64)t;
}
inline u64 sbb(u64 a, u64 b, u64& bw) {
  u128 t = (u128)a - b - bw;
  bw = (u64)(t >> 64) & 1;
  return (u64)t;
}
inline u64 add256(u256& r, const u256& a, const u256& b) {
  u64 c = 0;
  for (int i = 0; i < 4; ++i) r.v[i] = adc(a.v[i], b.v[i], c);
  return c;
}
inline u64 sub256(u256& r, const u256& a, const u256& b) {
  u64 bw = 0;
  for (int i = 0; i < 4; ++i) r.v[i] = sbb(a.v[i], b.v[i], bw);
  return bw;
}
inline void cmov256(u256& r, const u256& a, u64 mask) {   // r = mask ? a : r
  mask = value_barrier(mask);
  for (int i = 0; i < 4; ++i) r.v[i] ^= (r.v[i] ^ a.v[i]) & mask;
}
inline u64 is_zero256(const u256& a) { return value_barrier(~mask_nonzero(a.v[0] | a.v[1] | a.v[2] | a.v[3])); }
inline void from_be(u256& r, const uint8_t* b) {
  for (int i = 0; i < 4; ++i) {
    u64 w = 0;
    for (int j = 0; j < 8; ++j) w = (w << 8) | b[(3 - i) * 8 + j];
    r.v[i] = w;
  }
}
inline void to_be(uint8_t* b, const u256& a) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) b[(3 - i) * 8 + j] = (uint8_t)(a.v[i] >> (56 - 8 * j));
}
inline void mul_wide(u64 t[8], const u256& a, const u256& b) {
  for (int i = 0; i < 8; ++i) t[i] = 0;
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 4; ++j) {
      u128 x = (u128)a.v[i] * b.v[j] + t[i + j] + carry;
      t[i + j] = (u64)x;
      carry = (u64)(x >> 64);
    }
    t[i + 4] = carry;
  }
}

// ---- GF(p) ------------------------------------------------------------------------------------
const u256 FP_P = {{0xFFFFFFFEFFFFFC2Full, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull, 0xFFFFFFFFFFFFFFFFull}};
const u64 FP_C = 0x1000003D1ull;   // 2^256 - p
typedef u256 fe;                   // always fully reduced

inline void fe_reduce_once(fe& r, u64 carry) {   // value = r + carry * 2^256 < 2p  ->  [0, p)
  u256 t;
  u64 c = 0;
  t.v[0] = adc(r.v[0], FP_C, c);
  for (int i = 1; i < 4; ++i) t.v[i] = adc(r.v[i], 0, c);
  cmov256(r, t, mask_nonzero(carry | c));
}
inline void fe_add(fe& r, const fe& a, const fe& b) {
  u64 c = add256(r, a, b);
  fe_reduce_once(r, c);
}
inline void fe_sub(fe& r, const fe& a, const fe& b) {
  u64 bw = sub256(r, a, b);
  u256 t;
  add256(t, r, FP_P);
  cmov256(r, t, mask_nonzero(bw));
}
inline void fe_neg(fe& r, const fe& a) {
  fe z = {{0, 0, 0, 0}};
  fe_sub(r, z, a);
}
thread_local uint64_t g_fe_mul_count = 0;   // instrumentation for the tests: the operation count must not depend on secrets
inline void fe_mul(fe& r, const fe& a, const fe& b) {
  ++g_fe_mul_count;
  u64 t[8];
  mul_wide(t, a, b);
  // fold the high half: t_lo + t_hi * C  (< 2^256 * 2^34)
  u64 carry = 0, acc[5];
  for (int i = 0; i < 4; ++i) {
    u128 x = (u128)t[i + 4] * FP_C + t[i] + carry;
    acc[i] = (u64)x;
    carry = (u64)(x >> 64);
  }
  acc[4] = carry;   // < 2^34
  // fold again: acc[4] * C < 2^67
  u128 x = (u128)acc[4] * FP_C + acc[0];
  r.v[0] = (u64)x;
  u64 c = (u64)(x >> 64);
  r.v[1] = adc(acc[1], 0, c);
  r.v[2] = adc(acc[2], 0, c);
  r.v[3] = adc(acc[3], 0, c);
  // value = r + c * 2^256 with c in {0,1}; if c, r is tiny, so one more fold cannot carry again
  u256 t2;
  u64 c2 = 0;
  t2.v[0] = adc(r.v[0], FP_C, c2);
  for (int i = 1; i < 4; ++i) t2.v[i] = adc(r.v[i], 0, c2);
  cmov256(r, t2, mask_nonzero(c));
  fe_reduce_once(r, 0);
}
inline void fe_sqr(fe& r, const fe& a) { fe_mul(r, a, a); }
inline void fe_mul_small(fe& r, const fe& a, u64 k) {
  fe kk = {{k, 0, 0, 0}};
  fe_mul(r, a, kk);
}
void fe_sqr_n(fe& r, const fe& a, int n) {
  r = a;
  for (int i = 0; i < n; ++i) fe_sqr(r, r);
}
// a^(p-2), fixed addition chain (Element.Invert, internal/field/field_invert.go:11); 0 -> 0
void fe_inv(fe& r, const fe& a) {
  fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
  fe_sqr(t, a); fe_mul(x2, t, a);
  fe_sqr(t, x2); fe_mul(x3, t, a);
  fe_sqr_n(t, x3, 3); fe_mul(x6, t, x3);
  fe_sqr_n(t, x6, 3); fe_mul(x9, t, x3);
  fe_sqr_n(t, x9, 2); fe_mul(x11, t, x2);
  fe_sqr_n(t, x11, 11); fe_mul(x22, t, x11);
  fe_sqr_n(t, x22, 22); fe_mul(x44, t, x22);
  fe_sqr_n(t, x44, 44); fe_mul(x88, t, x44);
  fe_sqr_n(t, x88, 88); fe_mul(x176, t, x88);
  fe_sqr_n(t, x176, 44); fe_mul(x220, t, x44);
  fe_sqr_n(t, x220, 3); fe_mul(x223, t, x3);
  fe_sqr_n(t, x223, 23); fe_mul(t, t, x22);
  fe_sqr_n(t, t, 5); fe_mul(t, t, a);
  fe_sqr_n(t, t, 3); fe_mul(t, t, x2);
  fe_sqr_n(t, t, 2); fe_mul(r, t, a);
}
// ---- Z/n --------------------------------------------------------------------------------------
const u256 SC_N = {{0xBFD25E8CD0364141ull, 0xBAAEDCE6AF48A03Bull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull}};
const u64 SC_C[3] = {0x402DA1732FC9BEBFull, 0x4551231950B75FC4ull, 1ull};   // 2^256 - n (129 bits)
const u256 SC_HALF_N = {{0xDFE92F46681B20A0ull, 0x5D576E7357A4501Dull, 0xFFFFFFFFFFFFFFFFull, 0x7FFFFFFFFFFFFFFFull}};
typedef u256 sc;   // always < n

inline void sc_reduce_once(sc& r, u64 carry) {   // r + carry * 2^256 < 2n -> [0, n)
  u256 t;
  u64 bw = sub256(t, r, SC_N);
  cmov256(r, t, mask_nonzero(carry) | ~mask_nonzero(bw));
}
inline void sc_add(sc& r, const sc& a, const sc& b) {
  u64 c = add256(r, a, b);
  sc_reduce_once(r, c);
}
inline void sc_neg(sc& r, const sc& a) {
  u256 t;
  sub256(t, SC_N, a);
  u64 z = is_zero256(a);
  for (int i = 0; i < 4; ++i) r.v[i] = t.v[i] & ~z;
}
inline u64 sc_gt_half_n(const sc& a) {   // mask
  u256 t;
  return mask_nonzero(sub256(t, SC_HALF_N, a));
}
inline void sc_cneg(sc& r, const sc& a, u64 mask) {
  sc t;
  sc_neg(t, a);
  r = a;
  cmov256(r, t, mask);
}
// hi * (2^256 - n) + lo for a value of up to `nh` high limbs
void sc_reduce_wide(sc& r, const u64 t[8]) {
  // first fold: t[4..7] * C (385 bits) + t[0..3]  -> 7 limbs
  u64 a[8] = {t[0], t[1], t[2], t[3], 0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 3; ++j) {
      u128 x = (u128)t[4 + i] * SC_C[j] + a[i + j] + carry;
      a[i + j] = (u64)x;
      carry = (u64)(x >> 64);
    }
    for (int k = i + 3; k < 8; ++k) {
      u128 x = (u128)a[k] + carry;
      a[k] = (u64)x;
      carry = (u64)(x >> 64);
    }
  }
  // second fold: a[4..6] (130 bits) * C + a[0..3] -> 5 limbs (< 2^260)
  u64 b[8] = {a[0], a[1], a[2], a[3], 0, 0, 0, 0};
  for (int i = 0; i < 3; ++i) {
    u64 carry = 0;
    for (int j = 0; j < 3; ++j) {
      u128 x = (u128)a[4 + i] * SC_C[j] + b[i + j] + carry;
      b[i + j] = (u64)x;
      carry = (u64)(x >> 64);
    }
    for (int k = i + 3; k < 8; ++k) {
      u128 x = (u128)b[k] + carry;
      b[k] = (u64)x;
      carry = (u64)(x >> 64);
    }
  }
  // third fold: b[4] (a few bits) * C + b[0..3] < 2^256 + 2^134
  u64 c[5] = {b[0], b[1], b[2], b[3], 0};
  {
    u64 carry = 0;
    for (int j = 0; j < 3; ++j) {
      u128 x = (u128)b[4] * SC_C[j] + c[j] + carry;
      c[j] = (u64)x;
      carry = (u64)(x >> 64);
    }
    for (int k = 3; k < 5; ++k) {
      u128 x = (u128)c[k] + carry;
      c[k] = (u64)x;
      carry = (u64)(x >> 64);
    }
  }
  r.v[0] = c[0]; r.v[1] = c[1]; r.v[2] = c[2]; r.v[3] = c[3];
  sc_reduce_once(r, c[4]);   // value < 2n here: c[4] in {0,1} and then r is small
  sc_reduce_once(r, 0);
}
inline void sc_mul(sc& r, const sc& a, const sc& b) {
  u64 t[8];
  mul_wide(t, a, b);
  sc_reduce_wide(r, t);
}
inline void sc_from_be_reduce(sc& r, const uint8_t* b) {   // Scalar.SetBytes (scalar.go:123): one conditional subtraction
  from_be(r, b);
  sc_reduce_once(r, 0);
}
// a^(n-2) by square-and-multiply over the PUBLIC exponent (fixed sequence: constant time); 0 -> 0
void sc_inv(sc& r, const sc& a) {
  u256 e;
  const u256 two = {{2, 0, 0, 0}};
  sub256(e, SC_N, two);
  sc acc = {{1, 0, 0, 0}};
  for (int i = 255; i >= 0; --i) {
    sc_mul(acc, acc, acc);
    if ((e.v[i >> 6] >> (i & 63)) & 1) sc_mul(acc, acc, a);   // exponent bits are public
  }
  r = acc;
}

// GLV (point_mul_glv.go:37-117): constants, mulGFlooredDiv, splitGLV
const sc SC_NEG_LAMBDA = {{0xE0CFC810B51283CFull, 0xA880B9FC8EC739C2ull, 0x5AD9E3FD77ED9BA4ull, 0xAC9C52B33FA3CF1Full}};
const sc SC_NEG_B1 = {{0x6F547FA90ABFE4C3ull, 0xE4437ED6010E8828ull, 0, 0}};
const sc SC_NEG_B2 = {{0xD765CDA83DB1562Cull, 0x8A280AC50774346Dull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull}};
const u256 SC_G1 = {{0xE893209A45DBB031ull, 0x3DAA8A1471E8CA7Full, 0xE86C90E49284EB15ull, 0x3086D221A7D46BCDull}};
const u256 SC_G2 = {{0x1571B4AE8AC47F71ull, 0x221208AC9DF506C6ull, 0x6F547FA90ABFE4C4ull, 0xE4437ED6010E8828ull}};
const fe FE_BETA = {{0xC1396C28719501EEull, 0x9CF0497512F58995ull, 0x6E64479EAC3434E9ull, 0x7AE96A2B657C0710ull}};

inline void sc_mul_g_floored_div(sc& r, const sc& k, const u256& g) {   // (k*g + 2^383) >> 384
  u64 t[8];
  mul_wide(t, k, g);
  u64 round = t[5] >> 63;
  u64 c = round;
  r.v[0] = adc(t[6], 0, c);
  r.v[1] = adc(t[7], 0, c);
  r.v[2] = c;
  r.v[3] = 0;
}
void sc_split_glv(sc& k1, sc& k2, const sc& k) {
  sc c1, c2, t;
  sc_mul_g_floored_div(c1, k, SC_G1);
  sc_mul_g_floored_div(c2, k, SC_G2);
  sc_mul(k2, c1, SC_NEG_B1);
  sc_mul(t, c2, SC_NEG_B2);
  sc_add(k2, k2, t);
  sc_mul(t, k2, SC_NEG_LAMBDA);
  sc_add(k1, k, t);
}

// ---- group: complete projective formulas, a = 0, b3 = 21 ----------------------------------------
struct pt {
  fe x, y, z;
};
const fe FE_ZERO = {{0, 0, 0, 0}}, FE_ONE = {{1, 0, 0, 0}};
const fe FE_GX = {{0x59F2815B16F81798ull, 0x029BFCDB2DCE28D9ull, 0x55A06295CE870B07ull, 0x79BE667EF9DCBBACull}};
const fe FE_GY = {{0x9C47D08FFB10D4B8ull, 0xFD17B448A6855419ull, 0x5DA4FBFC0E1108A8ull, 0x483ADA7726A3C465ull}};

inline void pt_identity(pt& r) {
  r.x = FE_ZERO;
  r.y = FE_ONE;
  r.z = FE_ZERO;
}
inline void pt_cmov(pt& r, const pt& a, u64 mask) {
  cmov256(r.x, a.x, mask);
  cmov256(r.y, a.y, mask);
  cmov256(r.z, a.z, mask);
}
inline void pt_cneg(pt& r, u64 mask) {
  fe ny;
  fe_neg(ny, r.y);
  cmov256(r.y, ny, mask);
}
// Algorithm 7 (addComplete, point_projective.go:24)
void pt_add(pt& r, const pt& p, const pt& q) {
  fe t0, t1, t2, t3, t4, x3, y3, z3;
  fe_mul(t0, p.x, q.x);
  fe_mul(t1, p.y, q.y);
  fe_mul(t2, p.z, q.z);
  fe_add(t3, p.x, p.y);
  fe_add(t4, q.x, q.y);
  fe_mul(t3, t3, t4);
  fe_add(t4, t0, t1);
  fe_sub(t3, t3, t4);
  fe_add(t4, p.y, p.z);
  fe_add(x3, q.y, q.z);
  fe_mul(t4, t4, x3);
  fe_add(x3, t1, t2);
  fe_sub(t4, t4, x3);
  fe_add(x3, p.x, p.z);
  fe_add(y3, q.x, q.z);
  fe_mul(x3, x3, y3);
  fe_add(y3, t0, t2);
  fe_sub(y3, x3, y3);
  fe_add(x3, t0, t0);
  fe_add(t0, x3, t0);
  fe_mul_small(t2, t2, 21);
  fe_add(z3, t1, t2);
  fe_sub(t1, t1, t2);
  fe_mul_small(y3, y3, 21);
  fe_mul(x3, t4, y3);
  fe_mul(t2, t3, t1);
  fe_sub(x3, t2, x3);
  fe_mul(y3, y3, t0);
  fe_mul(t1, t1, z3);
  fe_add(y3, t1, y3);
  fe_mul(t0, t0, t3);
  fe_mul(z3, z3, t4);
  fe_add(z3, z3, t0);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}
// Algorithm 8 (addMixed, point_projective.go:123): q = (qx, qy) affine, not the identity
void pt_add_mixed(pt& r, const pt& p, const fe& qx, const fe& qy) {
  fe t0, t1, t2, t3, t4, x3, y3, z3;
  fe_mul(t0, p.x, qx);
  fe_mul(t1, p.y, qy);
  fe_add(t3, qx, qy);
  fe_add(t4, p.x, p.y);
  fe_mul(t3, t3, t4);
  fe_add(t4, t0, t1);
  fe_sub(t3, t3, t4);
  fe_mul(t4, qy, p.z);
  fe_add(t4, t4, p.y);
  fe_mul(y3, qx, p.z);
  fe_add(y3, y3, p.x);
  fe_add(x3, t0, t0);
  fe_add(t0, x3, t0);
  fe_mul_small(t2, p.z, 21);
  fe_add(z3, t1, t2);
  fe_sub(t1, t1, t2);
  fe_mul_small(y3, y3, 21);
  fe_mul(x3, t4, y3);
  fe_mul(t2, t3, t1);
  fe_sub(x3, t2, x3);
  fe_mul(y3, y3, t0);
  fe_mul(t1, t1, z3);
  fe_add(y3, t1, y3);
  fe_mul(t0, t0, t3);
  fe_mul(z3, z3, t4);
  fe_add(z3, z3, t0);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}
// Algorithm 9 (doubleComplete, point_projective.go:208)
void pt_double(pt& r, const pt& p) {
  fe t0, t1, t2, x3, y3, z3;
  fe_sqr(t0, p.y);
  fe_add(z3, t0, t0);
  fe_add(z3, z3, z3);
  fe_add(z3, z3, z3);
  fe_mul(t1, p.y, p.z);
  fe_sqr(t2, p.z);
  fe_mul_small(t2, t2, 21);
  fe_mul(x3, t2, z3);
  fe_add(y3, t0, t2);
  fe_mul(z3, t1, z3);
  fe_add(t1, t2, t2);
  fe_add(t2, t1, t2);
  fe_sub(t0, t0, t2);
  fe_mul(y3, t0, y3);
  fe_add(y3, x3, y3);
  fe_mul(t1, p.x, p.y);
  fe_mul(x3, t0, t1);
  fe_add(x3, x3, x3);
  r.x = x3;
  r.y = y3;
  r.z = z3;
}

// 65-byte record <-> point.  Parsing handles PUBLIC data (the peer's point) and may branch.
bool pt_from_record(pt& p, const uint8_t* rec) {
  if (rec[0] == 0x00) {
    for (int i = 1; i < 65; ++i)
      if (rec[i]) return false;
    pt_identity(p);
    return true;
  }
  if (rec[0] != 0x04) return false;
  u256 t;
  from_be(p.x, rec + 1);
  from_be(p.y, rec + 33);
  if (!sub256(t, p.x, FP_P) || !sub256(t, p.y, FP_P)) return false;   // canonical coordinates (point_s11n.go:187-201)
  fe l, r3, seven = {{7, 0, 0, 0}};
  fe_sqr(l, p.y);
  fe_sqr(r3, p.x);
  fe_mul(r3, r3, p.x);
  fe_add(r3, r3, seven);
  fe_sub(l, l, r3);
  if (!is_zero256(l)) return false;
  p.z = FE_ONE;
  return true;
}
// affine output without branching on the (secret-derived) result: identity -> all-zero record
void pt_to_record(uint8_t* rec, const pt& p) {
  fe zi, x, y;
  fe_inv(zi, p.z);   // 0 -> 0
  fe_mul(x, p.x, zi);
  fe_mul(y, p.y, zi);
  u64 inf = is_zero256(p.z);
  uint8_t xb[32], yb[32];
  to_be(xb, x);
  to_be(yb, y);
  uint8_t keep = (uint8_t)(~inf & 0xFF);
  rec[0] = 0x04 & keep;
  for (int i = 0; i < 32; ++i) {
    rec[1 + i] = xb[i] & keep;
    rec[33 + i] = yb[i] & keep;
  }
}

// lookupProjectivePoint (point_mul_table_ref.go:11): idx in [0, 15], 0 = identity; full masked scan
void lookup_projective(pt& out, const pt tbl[15], u64 idx) {
  pt_identity(out);
  for (u64 i = 1; i < 16; ++i) pt_cmov(out, tbl[i - 1], mask_eq(idx, i));
}
// newProjectivePointMultTable (point_mul_table.go:51): [1P .. 15P]
void make_table(pt tbl[15], const pt& p) {
  tbl[0] = p;
  for (int i = 1; i < 15; i += 2) {
    pt_double(tbl[i], tbl[i / 2]);
    pt_add(tbl[i + 1], tbl[i], p);
  }
}

// Point.ScalarMult (point_mul_glv.go:257-303)
void ct_scalar_mult(pt& v, const sc& s, const pt& p) {
  pt pee = p, pee_prime = p;
  fe_mul(pee_prime.x, pee_prime.x, FE_BETA);   // mulBeta (:191)
  sc k1, k2;
  sc_split_glv(k1, k2, s);
  u64 neg1 = sc_gt_half_n(k1);
  sc_cneg(k1, k1, neg1);
  pt_cneg(pee, neg1);
  u64 neg2 = sc_gt_half_n(k2);
  sc_cneg(k2, k2, neg2);
  pt_cneg(pee_prime, neg2);
  pt t1[15], t2[15];
  make_table(t1, pee);
  make_table(t2, pee_prime);
  pt_identity(v);
  pt add;
  for (int i = 31; i >= 0; --i) {   // 32 four-bit windows of the 128-bit halves, top first
    if (i != 31) {
      pt_double(v, v);
      pt_double(v, v);
      pt_double(v, v);
      pt_double(v, v);
    }
    u64 w1 = (k1.v[i >> 4] >> ((i & 15) * 4)) & 15, w2 = (k2.v[i >> 4] >> ((i & 15) * 4)) & 15;
    lookup_projective(add, t1, w1);   // SelectAndAdd (point_mul_table.go:34)
    pt_add(v, v, add);
    lookup_projective(add, t2, w2);
    pt_add(v, v, add);
  }
}

// Point.MultiScalarMult for l != 1 (point_mul_multi.go:35-66): Straus with one 15-entry table per point,
// the doublings shared by all terms, the high nibble of each scalar byte before the low one, every
// table access a masked scan (projectivePointMultTable.SelectAndAdd, point_mul_table.go:34-41).
// `tbl` is caller-provided scratch of 15 * l points.  The sequence of field operations depends on l
// alone (l is public: it is the length of the caller's slices).
void ct_multi_scalar_mult(pt& v, size_t l, const sc* s, const pt* p, pt* tbl) {
  for (size_t j = 0; j < l; ++j) make_table(tbl + 15 * j, p[j]);
  pt_identity(v);
  pt add;
  for (int i = 0; i < 32; ++i) {   // scalar bytes, most significant first (Scalar.getBytes is big-endian)
    int limb = (31 - i) >> 3, sh = ((31 - i) & 7) * 8;
    for (int half = 0; half < 2; ++half) {
      if (i != 0 || half != 0) {
        pt_double(v, v);
        pt_double(v, v);
        pt_double(v, v);
        pt_double(v, v);
      }
      for (size_t j = 0; j < l; ++j) {
        u64 b = (s[j].v[limb] >> sh) & 0xff;
        lookup_projective(add, tbl + 15 * j, half ? (b & 15) : (b >> 4));
        pt_add(v, v, add);
      }
    }
  }
}

// generator tables for ScalarBaseMult: 64 tables of [1..15] * 16^i * G, affine (the reference's
// generatorHugeAffineTable / generatorOddAffineTable pair, point_mul_table.go:78-160), built once
// from public data
struct apt {
  fe x, y;
};
apt g_base_tbl[64][15];
std::once_flag g_base_once;
void build_base_tables() {
  pt base;
  base.x = FE_GX;
  base.y = FE_GY;
  base.z = FE_ONE;
  for (int i = 0; i < 64; ++i) {
    pt mult[15];
    make_table(mult, base);
    for (int j = 0; j < 15; ++j) {   // public data: plain inversions
      fe zi;
      fe_inv(zi, mult[j].z);
      fe_mul(g_base_tbl[i][j].x, mult[j].x, zi);
      fe_mul(g_base_tbl[i][j].y, mult[j].y, zi);
    }
    pt_double(base, mult[7]);   // 16 * base = 2 * (8 * base)
  }
}
// Point.ScalarBaseMult (point_mul_table.go:168-194) with affinePointMultTable.SelectAndAdd (:118)
void ct_scalar_base_mult(pt& v, const sc& s) {
  std::call_once(g_base_once, build_base_tables);
  pt_identity(v);
  for (int i = 0; i < 64; ++i) {
    u64 w = (s.v[i >> 4] >> ((i & 15) * 4)) & 15;
    apt a;
    a.x = FE_GX;   // any curve point: replaced by the scan unless w == 0, and then the sum is discarded
    a.y = FE_GY;
    for (u64 j = 1; j < 16; ++j) {   // lookupAffinePoint (point_mul_table_ref.go:18)
      u64 m = mask_eq(w, j);
      cmov256(a.x, g_base_tbl[i][j - 1].x, m);
      cmov256(a.y, g_base_tbl[i][j - 1].y, m);
    }
    pt tmp;
    pt_add_mixed(tmp, v, a.x, a.y);
    pt_cmov(v, tmp, mask_nonzero(w));   // uncheckedConditionalSelect(tmp, sum, isInfinity)
  }
}


// a^((p+1)/4) with the addition chain of the inversion's prefix (Element.Sqrt, internal/field/field_sqrt_ratio.go:14:
// p = 3 mod 4, so this is a square root whenever one exists); returns the mask "r^2 == a"
u64 fe_sqrt(fe& r, const fe& a) {
  fe x2, x3, x6, x9, x11, x22, x44, x88, x176, x220, x223, t;
  fe_sqr(t, a); fe_mul(x2, t, a);
  fe_sqr(t, x2); fe_mul(x3, t, a);
  fe_sqr_n(t, x3, 3); fe_mul(x6, t, x3);
  fe_sqr_n(t, x6, 3); fe_mul(x9, t, x3);
  fe_sqr_n(t, x9, 2); fe_mul(x11, t, x2);
  fe_sqr_n(t, x11, 11); fe_mul(x22, t, x11);
  fe_sqr_n(t, x22, 22); fe_mul(x44, t, x22);
  fe_sqr_n(t, x44, 44); fe_mul(x88, t, x44);
  fe_sqr_n(t, x88, 88); fe_mul(x176, t, x88);
  fe_sqr_n(t, x176, 44); fe_mul(x220, t, x44);
  fe_sqr_n(t, x220, 3); fe_mul(x223, t, x3);
  fe_sqr_n(t, x223, 23); fe_mul(t, t, x22);
  fe_sqr_n(t, t, 6); fe_mul(t, t, x2);
  fe_sqr_n(r, t, 2);
  fe chk;
  fe_sqr(chk, r);
  fe_sub(chk, chk, a);
  return is_zero256(chk);
}

// 65-byte record -> point WITHOUT branching on what the record holds (the single-operation calls: an operand may be an
// intermediate value of a secret computation, and whether it is the identity is then a secret too).  Returns the mask
// "the record is a Point the reference can hold": the identity record, or 0x04 || X || Y canonical and on the curve
// (SetUncompressedBytes, point_s11n.go:178-201).  The caller branches on THAT only (a malformed record is a usage error).
u64 pt_from_record_ct(pt& p, const uint8_t* rec) {
  u64 body = 0;
  for (int i = 1; i < 65; ++i) body |= rec[i];
  const u64 is_id = mask_eq(rec[0], 0x00) & ~mask_nonzero(body);
  u256 t;
  from_be(p.x, rec + 1);
  from_be(p.y, rec + 33);
  const u64 canon = mask_nonzero(sub256(t, p.x, FP_P)) & mask_nonzero(sub256(t, p.y, FP_P));
  fe zero = FE_ZERO;
  cmov256(p.x, zero, ~canon);        // (the arithmetic below assumes reduced operands; the verdict is already "no")
  cmov256(p.y, zero, ~canon);
  fe l, r3, seven = {{7, 0, 0, 0}};
  fe_sqr(l, p.y);
  fe_sqr(r3, p.x);
  fe_mul(r3, r3, p.x);
  fe_add(r3, r3, seven);
  fe_sub(l, l, r3);
  const u64 is_affine = mask_eq(rec[0], 0x04) & canon & is_zero256(l);
  p.z = FE_ONE;
  pt id;
  pt_identity(id);
  pt_cmov(p, id, ~is_affine);        // (a rejected record leaves the identity behind, never an off-curve point)
  return is_id | is_affine;
}
inline u64 ctrl_mask(uint64_t ctrl) { return mask_nonzero((u64)ctrl); }   // the reference's ctrl: 0 or "otherwise"
inline u64 sc_from_be_canonical(sc& r, const uint8_t* b) {               // mask: b < n (SetCanonicalBytes, scalar.go:135)
  u256 t;
  from_be(r, b);
  const u64 ok = mask_nonzero(sub256(t, r, SC_N));
  sc_reduce_once(r, 0);
  return ok;
}
inline u64 fe_from_be_canonical(fe& r, const uint8_t* b) {               // mask: b < p (NewElementFromCanonicalBytes, field.go:153)
  u256 t;
  from_be(r, b);
  const u64 ok = mask_nonzero(sub256(t, r, FP_P));
  fe zero = FE_ZERO;
  cmov256(r, zero, ~ok);
  return ok;
}
}  // namespace

extern "C" {

// out = a - b on 65-byte records (public data: the error points of the batch bisection, msm.hip)
__attribute__((visibility("hidden"))) int s2k_internal_point_sub65(const uint8_t* a, const uint8_t* b, uint8_t* out) {
  pt pa, pb, r;
  if (!pt_from_record(pa, a) || !pt_from_record(pb, b)) return 1;
  pt_cneg(pb, ~(u64)0);
  pt_add(r, pa, pb);
  pt_to_record(out, r);
  return 0;
}

// field multiplications executed by this thread since the last call (test instrumentation)
uint64_t s2k_ct_debug_fe_mul_count(void) {
  uint64_t c = g_fe_mul_count;
  g_fe_mul_count = 0;
  return c;
}

int s2k_ct_scalar_mult(const uint8_t k[32], const uint8_t point65[65], uint8_t out65[65]) {
  if (!k || !point65 || !out65) return S2K_ERR_ARG;
  pt p, v;
  if (!pt_from_record(p, point65)) return S2K_ERR_ARG;
  sc s;
  sc_from_be_reduce(s, k);
  ct_scalar_mult(v, s, p);
  pt_to_record(out65, v);
  return S2K_OK;
}

int s2k_ct_scalar_base_mult(const uint8_t k[32], uint8_t out65[65]) {
  if (!k || !out65) return S2K_ERR_ARG;
  sc s;
  sc_from_be_reduce(s, k);
  pt v;
  ct_scalar_base_mult(v, s);
  pt_to_record(out65, v);
  return S2K_OK;
}

// Point.MultiScalarMult (point_mul_multi.go:25-67), the constant-time form: n == 1 is Point.ScalarMult
// (:31-33), n == 0 leaves the identity (the loops of :35-66 add nothing to v.Identity()), otherwise Straus
// over per-point tables with masked scans.  The scalars are secret, the points and n are public.
int s2k_ct_multi_scalar_mult(size_t n, const uint8_t* k, const uint8_t* points65, uint8_t out65[65]) {
  if (!out65 || (n && (!k || !points65))) return S2K_ERR_ARG;
  if (n > ((size_t)1 << 24)) return S2K_ERR_ARG;   // 15 tables entries of 96 bytes per term: keep the scratch below 24 GiB
  pt v;
  if (n == 0) {
    pt_identity(v);
    pt_to_record(out65, v);
    return S2K_OK;
  }
  if (n == 1) return s2k_ct_scalar_mult(k, points65, out65);
  pt* p = (pt*)malloc(n * sizeof(pt));
  sc* s = (sc*)malloc(n * sizeof(sc));
  pt* tbl = (pt*)malloc(n * 15 * sizeof(pt));
  int rc = (p && s && tbl) ? S2K_OK : S2K_ERR_NOMEM;
  if (rc == S2K_OK)
    for (size_t j = 0; j < n; ++j)   // public data: may stop at the first malformed record
      if (!pt_from_record(p[j], points65 + 65 * j)) {
        rc = S2K_ERR_ARG;
        break;
      }
  if (rc == S2K_OK) {
    for (size_t j = 0; j < n; ++j) sc_from_be_reduce(s[j], k + 32 * j);
    ct_multi_scalar_mult(v, n, s, p, tbl);
    pt_to_record(out65, v);
  }
  if (s) {   // the only secret-derived scratch that outlives the call frame
    volatile u64* w = (volatile u64*)s;
    for (size_t i = 0; i < n * 4; ++i) w[i] = 0;
  }
  free(tbl);
  free(s);
  free(p);
  return rc;
}

// PrivateKey.ECDH (secec/secec.go:53-56): x(d * Q).  The public point must be a valid non-identity
// point (NewPublicKey, secec.go:188-216) and d in [1, n) (NewPrivateKey, secec.go:153-170), so the
// product is never the identity.
int s2k_ct_ecdh(const uint8_t priv32[32], const uint8_t pub65[65], uint8_t shared_x[32]) {
  if (!priv32 || !pub65 || !shared_x) return S2K_ERR_ARG;
  pt p, v;
  if (pub65[0] != 0x04 || !pt_from_record(p, pub65)) return S2K_ERR_ARG;
  u256 raw, t;
  from_be(raw, priv32);
  if (sub256(t, raw, SC_N) == 0) return S2K_ERR_ARG;   // d >= n: not a private key (the check reveals only that)
  sc d = raw;
  u64 dz = is_zero256(d);
  ct_scalar_mult(v, d, p);
  uint8_t rec[65];
  pt_to_record(rec, v);
  memcpy(shared_x, rec + 1, 32);
  return dz ? S2K_ERR_ARG : S2K_OK;
}

// The arithmetic of ECDSA signing (secec/ecdsa.go:335-390) for a caller-supplied nonce: R = k*G,
// r = x(R) mod n, s = k^-1 (e + r d) mod n, s normalised to the lower half (:385-387) with the
// recovery id adjusted.  Returns S2K_ERR_ARG when d or k is not in [1, n), or r == 0 or s == 0
// (the reference then draws another nonce, :352-376).  The nonce derivation (RFC 6979 / entropy
// mixing, ecdsa.go:284-333) stays with the caller.
int s2k_ct_ecdsa_sign_raw(const uint8_t priv32[32], const uint8_t digest32[32], const uint8_t nonce32[32], uint8_t r32[32],
                          uint8_t s32[32], uint8_t* recovery_id) {
  if (!priv32 || !digest32 || !nonce32 || !r32 || !s32) return S2K_ERR_ARG;
  u256 draw, kraw, t;
  from_be(draw, priv32);
  from_be(kraw, nonce32);
  u64 bad = ~mask_nonzero(sub256(t, draw, SC_N)) | ~mask_nonzero(sub256(t, kraw, SC_N)) | is_zero256(draw) | is_zero256(kraw);
  sc d = draw, k = kraw, e;
  sc_reduce_once(d, 0);
  sc_reduce_once(k, 0);
  sc_from_be_reduce(e, digest32);   // hashToScalar (ecdsa.go:477-486) on the leftmost 32 bytes
  pt R;
  ct_scalar_base_mult(R, k);
  uint8_t rec[65];
  pt_to_record(rec, R);
  u256 rx;
  from_be(rx, rec + 1);
  u64 overflow = ~mask_nonzero(sub256(t, rx, SC_N));   // x(R) >= n
  sc r = rx;
  sc_reduce_once(r, 0);
  sc kinv, s, rd;
  sc_inv(kinv, k);
  sc_mul(rd, r, d);
  sc_add(rd, rd, e);
  sc_mul(s, kinv, rd);
  u64 high = sc_gt_half_n(s);
  sc_cneg(s, s, high);
  bad |= is_zero256(r) | is_zero256(s);
  to_be(r32, r);
  to_be(s32, s);
  if (recovery_id) *recovery_id = (uint8_t)(((rec[64] & 1) ^ (high & 1)) | ((overflow & 1) << 1));
  return bad ? S2K_ERR_ARG : S2K_OK;
}

// ---- single operations of Point / Scalar / field.Element, constant time ("single-op entry points served by
// the CPU backend so the Go Point / Scalar methods have something to call").  The batched GPU forms (s2k_point_add_batch,
// s2k_fn_op_batch ...) are variable time, need a context and a device round trip: right for a million public values, wrong
// for one secret nonce.  Operands are 65-byte point records / 32-byte canonical big-endian values; a record or value the
// reference's type cannot hold is S2K_ERR_ARG (the reference's constructors return an error for those).

// v = p + q — Point.Add (point.go:62); addComplete (point_projective.go:24)
int s2k_ct_point_add(const uint8_t a65[65], const uint8_t b65[65], uint8_t out65[65]) {
  if (!a65 || !b65 || !out65) return S2K_ERR_ARG;
  pt a, b, r;
  if (~(pt_from_record_ct(a, a65) & pt_from_record_ct(b, b65))) return S2K_ERR_ARG;
  pt_add(r, a, b);
  pt_to_record(out65, r);
  return S2K_OK;
}
// v = p + p — Point.Double (point.go:73); doubleComplete (point_projective.go:208)
int s2k_ct_point_double(const uint8_t a65[65], uint8_t out65[65]) {
  if (!a65 || !out65) return S2K_ERR_ARG;
  pt a, r;
  if (~pt_from_record_ct(a, a65)) return S2K_ERR_ARG;
  pt_double(r, a);
  pt_to_record(out65, r);
  return S2K_OK;
}
// v = p - q — Point.Subtract (point.go:83)
int s2k_ct_point_subtract(const uint8_t a65[65], const uint8_t b65[65], uint8_t out65[65]) {
  if (!a65 || !b65 || !out65) return S2K_ERR_ARG;
  pt a, b, r;
  if (~(pt_from_record_ct(a, a65) & pt_from_record_ct(b, b65))) return S2K_ERR_ARG;
  pt_cneg(b, ~(u64)0);
  pt_add(r, a, b);
  pt_to_record(out65, r);
  return S2K_OK;
}
// v = p iff ctrl == 0, v = -p otherwise — Point.ConditionalNegate (point.go:102); ctrl = 1 is Point.Negate (point.go:89)
int s2k_ct_point_conditional_negate(const uint8_t a65[65], uint64_t ctrl, uint8_t out65[65]) {
  if (!a65 || !out65) return S2K_ERR_ARG;
  pt a;
  if (~pt_from_record_ct(a, a65)) return S2K_ERR_ARG;
  pt_cneg(a, ctrl_mask(ctrl));
  // (the operand is affine or the identity: the record is rebuilt without the inversion of pt_to_record)
  const u64 keep = ~is_zero256(a.z);
  uint8_t xb[32], yb[32];
  to_be(xb, a.x);
  to_be(yb, a.y);
  out65[0] = (uint8_t)(0x04 & keep);
  for (int i = 0; i < 32; ++i) {
    out65[1 + i] = (uint8_t)(xb[i] & keep);
    out65[33 + i] = (uint8_t)(yb[i] & keep);
  }
  return S2K_OK;
}
int s2k_ct_point_negate(const uint8_t a65[65], uint8_t out65[65]) { return s2k_ct_point_conditional_negate(a65, 1, out65); }
// v = a iff ctrl == 0, v = b otherwise — Point.ConditionalSelect (point.go:115)
int s2k_ct_point_conditional_select(const uint8_t a65[65], const uint8_t b65[65], uint64_t ctrl, uint8_t out65[65]) {
  if (!a65 || !b65 || !out65) return S2K_ERR_ARG;
  pt a, b;
  if (~(pt_from_record_ct(a, a65) & pt_from_record_ct(b, b65))) return S2K_ERR_ARG;
  const uint8_t m = (uint8_t)ctrl_mask(ctrl);
  for (int i = 0; i < 65; ++i) out65[i] = (uint8_t)((a65[i] & ~m) | (b65[i] & m));
  return S2K_OK;
}
// Point.Equal (point.go:133: X1 Z2 == X2 Z1 and Y1 Z2 == Y2 Z1), Point.IsIdentity (:148), Point.IsYOdd (:155; 0 for the
// identity, whose rescaled y is 0): *out = 1 or 0
int s2k_ct_point_equal(const uint8_t a65[65], const uint8_t b65[65], uint64_t* out) {
  if (!a65 || !b65 || !out) return S2K_ERR_ARG;
  pt a, b;
  if (~(pt_from_record_ct(a, a65) & pt_from_record_ct(b, b65))) return S2K_ERR_ARG;
  fe x1z2, x2z1, y1z2, y2z1, dx, dy;
  fe_mul(x1z2, a.x, b.z);
  fe_mul(x2z1, b.x, a.z);
  fe_mul(y1z2, a.y, b.z);
  fe_mul(y2z1, b.y, a.z);
  fe_sub(dx, x1z2, x2z1);
  fe_sub(dy, y1z2, y2z1);
  *out = (is_zero256(dx) & is_zero256(dy)) & 1;
  return S2K_OK;
}
int s2k_ct_point_is_identity(const uint8_t a65[65], uint64_t* out) {
  if (!a65 || !out) return S2K_ERR_ARG;
  pt a;
  if (~pt_from_record_ct(a, a65)) return S2K_ERR_ARG;
  *out = is_zero256(a.z) & 1;
  return S2K_OK;
}
int s2k_ct_point_is_y_odd(const uint8_t a65[65], uint64_t* out) {
  if (!a65 || !out) return S2K_ERR_ARG;
  pt a;
  if (~pt_from_record_ct(a, a65)) return S2K_ERR_ARG;
  *out = a.y.v[0] & 1 & ~is_zero256(a.z);
  return S2K_OK;
}

// Scalar.Add / Subtract / Negate / Multiply / Square (scalar.go:66-93) and Scalar.Invert (scalar_invert.go:11; 0 -> 0):
// op is one of S2K_OP_MUL / SQR / ADD / SUB / NEG / INV, b is read by the binary ones.  Operands canonical (< n).
int s2k_ct_scalar_op(int op, const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32]) {
  if (!a32 || !out32) return S2K_ERR_ARG;
  const bool binary = op == S2K_OP_MUL || op == S2K_OP_ADD || op == S2K_OP_SUB;
  if (!binary && op != S2K_OP_SQR && op != S2K_OP_NEG && op != S2K_OP_INV) return S2K_ERR_ARG;
  if (binary && !b32) return S2K_ERR_ARG;
  sc a, b = {{0, 0, 0, 0}}, r;
  u64 ok = sc_from_be_canonical(a, a32);
  if (binary) ok &= sc_from_be_canonical(b, b32);
  if (~ok) return S2K_ERR_ARG;
  switch (op) {                       // (op is public)
    case S2K_OP_MUL: sc_mul(r, a, b); break;
    case S2K_OP_SQR: sc_mul(r, a, a); break;
    case S2K_OP_ADD: sc_add(r, a, b); break;
    case S2K_OP_SUB: sc_neg(b, b); sc_add(r, a, b); break;
    case S2K_OP_NEG: sc_neg(r, a); break;
    default: sc_inv(r, a); break;
  }
  to_be(out32, r);
  return S2K_OK;
}
// Scalar.ConditionalSelect (scalar.go:176): out = a iff ctrl == 0, b otherwise; Scalar.ConditionalNegate (scalar.go:168) is
// select(a, -a, ctrl) and has its own entry so that a shim does not need two calls
int s2k_ct_scalar_conditional_select(const uint8_t a32[32], const uint8_t b32[32], uint64_t ctrl, uint8_t out32[32]) {
  if (!a32 || !b32 || !out32) return S2K_ERR_ARG;
  sc a, b;
  if (~(sc_from_be_canonical(a, a32) & sc_from_be_canonical(b, b32))) return S2K_ERR_ARG;
  cmov256(a, b, ctrl_mask(ctrl));
  to_be(out32, a);
  return S2K_OK;
}
int s2k_ct_scalar_conditional_negate(const uint8_t a32[32], uint64_t ctrl, uint8_t out32[32]) {
  if (!a32 || !out32) return S2K_ERR_ARG;
  sc a, r;
  if (~sc_from_be_canonical(a, a32)) return S2K_ERR_ARG;
  sc_cneg(r, a, ctrl_mask(ctrl));
  to_be(out32, r);
  return S2K_OK;
}
// Scalar.Equal (scalar.go:182), IsZero (:187), IsGreaterThanHalfN (:196): *out = 1 or 0.  what: 0 = IsZero(a),
// 1 = IsGreaterThanHalfN(a), 2 = Equal(a, b)
int s2k_ct_scalar_predicate(int what, const uint8_t a32[32], const uint8_t b32[32], uint64_t* out) {
  if (!a32 || !out || what < 0 || what > 2 || (what == 2 && !b32)) return S2K_ERR_ARG;
  sc a, b = {{0, 0, 0, 0}};
  u64 ok = sc_from_be_canonical(a, a32);
  if (what == 2) ok &= sc_from_be_canonical(b, b32);
  if (~ok) return S2K_ERR_ARG;
  u64 m;
  if (what == 0) m = is_zero256(a);
  else if (what == 1) m = sc_gt_half_n(a);
  else {
    u256 d;
    sub256(d, a, b);
    m = is_zero256(d);
  }
  *out = m & 1;
  return S2K_OK;
}
// Scalar.SetBytes (scalar.go:123): out = src mod n (one conditional subtraction), *did_reduce = 1 iff src >= n
int s2k_ct_scalar_set_bytes(const uint8_t src32[32], uint8_t out32[32], uint64_t* did_reduce) {
  if (!src32 || !out32) return S2K_ERR_ARG;
  sc a;
  const u64 ok = sc_from_be_canonical(a, src32);
  to_be(out32, a);
  if (did_reduce) *did_reduce = ~ok & 1;
  return S2K_OK;
}
// field.Element: Multiply / Square / Add / Subtract / Negate (internal/field/field.go:61-104), Invert (field_invert.go:11;
// 0 -> 0), Sqrt (field_sqrt_ratio.go:14; *flag = 1 iff a is a square, out = the root the chain gives, else out = 0).
// Operands canonical (< p); flag may be NULL except for S2K_OP_SQRT.
int s2k_ct_fe_op(int op, const uint8_t a32[32], const uint8_t b32[32], uint8_t out32[32], uint64_t* flag) {
  if (!a32 || !out32) return S2K_ERR_ARG;
  const bool binary = op == S2K_OP_MUL || op == S2K_OP_ADD || op == S2K_OP_SUB;
  if (!binary && op != S2K_OP_SQR && op != S2K_OP_NEG && op != S2K_OP_INV && op != S2K_OP_SQRT) return S2K_ERR_ARG;
  if ((binary && !b32) || (op == S2K_OP_SQRT && !flag)) return S2K_ERR_ARG;
  fe a, b = FE_ZERO, r;
  u64 ok = fe_from_be_canonical(a, a32);
  if (binary) ok &= fe_from_be_canonical(b, b32);
  if (~ok) return S2K_ERR_ARG;
  u64 f = ~(u64)0;
  switch (op) {
    case S2K_OP_MUL: fe_mul(r, a, b); break;
    case S2K_OP_SQR: fe_sqr(r, a); break;
    case S2K_OP_ADD: fe_add(r, a, b); break;
    case S2K_OP_SUB: fe_sub(r, a, b); break;
    case S2K_OP_NEG: fe_neg(r, a); break;
    case S2K_OP_INV: fe_inv(r, a); break;
    default: {
      f = fe_sqrt(r, a);
      fe zero = FE_ZERO;
      cmov256(r, zero, ~f);
    }
  }
  to_be(out32, r);
  if (flag) *flag = f & 1;
  return S2K_OK;
}

}  // extern "C"

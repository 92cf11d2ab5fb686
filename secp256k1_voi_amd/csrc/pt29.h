// pt29.h — complete projective group law (Renes–Costello–Batina, a = 0, b3 = 21) over the lazy
// 9x29 field.  Same algorithms as the reference's addComplete / addMixed / doubleComplete
// (point_projective.go:24,123,208) and as point.h, arranged for fe29.h's unit budget (a product
// needs the operands' units to multiply to <= 7.8, summed over fused terms): every output
// coordinate is a sum of two products taken through ONE reduction (fe29_mul_add_mul), and the
// differences that feed those products are formed inside the preceding reductions
// (fe29_mul_plus).  11 (mixed) / 12 products with 8 / 9 reductions.  Units in [brackets].
//
// Used where arbitrary points meet arbitrary points and no fallback is wanted: the bucket
// pipeline of the multi-scalar multiplication (msm.hip).  Invariant for a `pt29`: x, y, z [1].
#pragma once
#include "fe29.h"

namespace s2k {

struct pt29 {
  fe29 x, y, z;
};

S2K_DEV pt29 pt29_identity() {   // (0 : 1 : 0), point.go:42
  pt29 r;
  r.x = fe29_zero();
  r.y = fe29_one();
  r.z = fe29_zero();
  return r;
}

// a * k for a small constant k (k < 2^6), up to 7 units in, 1 unit out: 64-bit product per limb
// with the carry folded along, the top folded with 2^256 = 0x3D1 + 8 * 2^29.
// The products go through mad64s (inline asm) on purpose: written as `c += (uint64_t)a.n[8] * k`,
// hipcc 7.2 (inlined after fe29_mul_tail, where limb 8 is known to be 24 bits wide) first
// narrows the product to a 24-bit multiply, drops the `& F29_M8` that produced the limb, and then
// widens it back into v_mad_u64_u32 on the unmasked register, which corrupts the fold count
// (found with tools/pt29_selftest.hip).
S2K_DEV fe29 fe29_mul_small_norm(const fe29& a, uint32_t k) {
  fe29 r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    mad64s(c, a.n[i], k);
    r.n[i] = (uint32_t)c & F29_M;
    c >>= 29;
  }
  mad64s(c, a.n[8], k);
  r.n[8] = (uint32_t)c & F29_M8;
  uint32_t x = (uint32_t)(c >> 24);            // multiples of 2^256, < 2^15
  r.n[0] += x * 0x3D1u;                        // < 2^29 + 2^25
  r.n[1] += x << 3;
  return r;
}
// 3a for a of 1 unit, 1 unit out
S2K_DEV fe29 fe29_triple_norm(const fe29& a) { return fe29_normalize_weak(fe29_mul_int(a, 3)); }

// shared tail of the additions: t0 = X1X2, t1 = Y1Y2 [1]; t2 = 21 Z1Z2 [1]; t3 = X1Y2 + X2Y1,
// t4 = Y1Z2 + Y2Z1 [1]; y3 = 21 (X1Z2 + X2Z1) [1]
S2K_DEV pt29 pt29_add_tail(const fe29& t0, const fe29& t1, const fe29& t2, const fe29& t3, const fe29& t4,
                           const fe29& y3) {
  fe29 t0n = fe29_triple_norm(t0);                                 // [1]   3 X1X2
  fe29 z3 = fe29_add(t1, t2);                                      // [2]
  fe29 t1m = fe29_add(t1, fe29_negate(t2, 1));                     // [3]
  pt29 r;
  r.x = fe29_mul_add_mul(t3, t1m, fe29_negate(t4, 1), y3);         // [1]*[3] + [2]*[1] -> [1]
  r.y = fe29_mul_add_mul(t1m, z3, y3, t0n);                        // [3]*[2] + [1]*[1] -> [1]
  r.z = fe29_mul_add_mul(z3, t4, t0n, t3);                         // [2]*[1] + [1]*[1] -> [1]
  return r;
}

// Algorithm 8: p + (qx, qy), q affine [1] and not the identity; p anything.
S2K_DEV pt29 pt29_add_mixed(const pt29& p, const fe29& qx, const fe29& qy) {
  fe29 t0 = fe29_mul(p.x, qx);                                                          // [1]
  fe29 t1 = fe29_mul(p.y, qy);                                                          // [1]
  fe29 t3 = fe29_mul_plus(fe29_add(qx, qy), fe29_add(p.x, p.y), fe29_negate(fe29_add(t0, t1), 2));   // [2]*[2] + [3] -> [1]
  fe29 t4 = fe29_mul_plus(qy, p.z, p.y);                                                // [1]
  fe29 y3 = fe29_mul_small_norm(fe29_mul_plus(qx, p.z, p.x), 21);                       // [1]
  fe29 t2 = fe29_mul_small_norm(p.z, 21);                                               // [1]
  return pt29_add_tail(t0, t1, t2, t3, t4, y3);
}

// Algorithm 7: p + q, both projective, no exceptions.
S2K_DEV pt29 pt29_add(const pt29& p, const pt29& q) {
  fe29 t0 = fe29_mul(p.x, q.x);                                                         // [1]
  fe29 t1 = fe29_mul(p.y, q.y);                                                         // [1]
  fe29 t2 = fe29_mul(p.z, q.z);                                                         // [1]
  fe29 t3 = fe29_mul_plus(fe29_add(p.x, p.y), fe29_add(q.x, q.y), fe29_negate(fe29_add(t0, t1), 2));   // [1]
  fe29 t4 = fe29_mul_plus(fe29_add(p.y, p.z), fe29_add(q.y, q.z), fe29_negate(fe29_add(t1, t2), 2));   // [1]
  fe29 y3 = fe29_mul_plus(fe29_add(p.x, p.z), fe29_add(q.x, q.z), fe29_negate(fe29_add(t0, t2), 2));   // [1]
  return pt29_add_tail(t0, t1, fe29_mul_small_norm(t2, 21), t3, t4, fe29_mul_small_norm(y3, 21));
}

// Algorithm 9: 2p.
S2K_DEV pt29 pt29_double(const pt29& p) {
  fe29 t0 = fe29_sqr(p.y);                                                              // [1]
  fe29 z3 = fe29_mul_int(fe29_normalize_weak(fe29_mul_int(t0, 4)), 2);                  // [2]   8 Y^2
  fe29 t1 = fe29_mul(p.y, p.z);                                                         // [1]
  fe29 zz = fe29_sqr(p.z);                                                              // [1]
  fe29 t2 = fe29_mul_small_norm(zz, 21);                                                // [1]   b3 Z^2
  fe29 y3 = fe29_add(t0, t2);                                                           // [2]
  fe29 t0m = fe29_normalize_weak(fe29_add(t0, fe29_negate(fe29_mul_small_norm(zz, 63), 1)));   // [3] -> [1]   Y^2 - 3 b3 Z^2
  pt29 r;
  r.y = fe29_mul_add_mul(t2, z3, t0m, y3);                                              // [1]*[2] + [1]*[2] -> [1]
  r.z = fe29_mul(t1, z3);                                                               // [1]*[2] -> [1]
  r.x = fe29_mul(fe29_mul_int(t0m, 2), fe29_mul(p.x, p.y));                             // [2]*[1] -> [1]
  return r;
}

}  // namespace s2k

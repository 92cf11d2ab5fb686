// pt26.h — complete projective group law (Renes–Costello–Batina, a = 0, b3 = 21) over the
// lazy 10x26 field.  Same algorithms as the reference's addComplete / addMixed /
// doubleComplete (point_projective.go:24,123,208) and as point.h, with the magnitude of every
// intermediate in [brackets] (fe26.h: multiplication inputs must be <= 8).
//
// Used where arbitrary points meet arbitrary points and no fallback is wanted: the bucket
// pipeline of the multi-scalar multiplication (msm.h).  Invariant for a `pt26`: x, y, z [<= 3].
#pragma once
#include "fe26.h"

namespace s2k {

struct pt26 {
  fe26 x, y, z;
};

S2K_DEV pt26 pt26_identity() {   // (0 : 1 : 0), point.go:42
  pt26 r;
  r.x = fe26_zero();
  r.y = fe26_one();
  r.z = fe26_zero();
  return r;
}

// a * k for a small constant k (k < 2^6), any magnitude <= 8 in, magnitude 1 out:
// 64-bit product per limb with the carry folded along, top folded with 2^256 = 0x3D1 + 2^32.
S2K_DEV fe26 fe26_mul_small_norm(const fe26& a, uint32_t k) {
  fe26 r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    c += (uint64_t)a.n[i] * k;
    r.n[i] = (uint32_t)c & F26_M;
    c >>= 26;
  }
  c += (uint64_t)a.n[9] * k;
  r.n[9] = (uint32_t)c & (F26_M >> 4);
  uint32_t x = (uint32_t)(c >> 22);            // multiples of 2^256, < 2^14
  r.n[0] += x * 0x3D1u;                        // <= 2^26 + 2^24
  r.n[1] += x << 6;
  return r;
}

// Algorithm 8: p + (qx, qy), q affine [1] and not the identity; p anything.  11 M
S2K_DEV pt26 pt26_add_mixed(const pt26& p, const fe26& qx, const fe26& qy) {
  fe26 t0 = fe26_mul(p.x, qx);                                     // [1]
  fe26 t1 = fe26_mul(p.y, qy);                                     // [1]
  fe26 t3 = fe26_mul(fe26_add(qx, qy), fe26_add(p.x, p.y));        // [2]*[6] -> [1]
  t3 = fe26_add(t3, fe26_negate(fe26_add(t0, t1), 2));             // [4]
  fe26 t4 = fe26_add(fe26_mul(qy, p.z), p.y);                      // [4]
  fe26 y3 = fe26_add(fe26_mul(qx, p.z), p.x);                      // [4]
  t0 = fe26_mul_int(t0, 3);                                        // [3]
  fe26 t2 = fe26_mul_small_norm(p.z, 21);                          // [1]
  fe26 z3 = fe26_add(t1, t2);                                      // [2]
  t1 = fe26_add(t1, fe26_negate(t2, 1));                           // [3]
  y3 = fe26_mul_small_norm(y3, 21);                                // [1]
  fe26 x3 = fe26_mul(t4, y3);                                      // [1]
  t2 = fe26_mul(t3, t1);                                           // [1]
  pt26 r;
  r.x = fe26_add(t2, fe26_negate(x3, 1));                          // [3]
  r.y = fe26_add(fe26_mul(t1, z3), fe26_mul(y3, t0));              // [2]
  r.z = fe26_add(fe26_mul(z3, t4), fe26_mul(t0, t3));              // [2]
  return r;
}

// Algorithm 7: p + q, both projective, no exceptions.  12 M
S2K_DEV pt26 pt26_add(const pt26& p, const pt26& q) {
  fe26 t0 = fe26_mul(p.x, q.x);                                    // [1]
  fe26 t1 = fe26_mul(p.y, q.y);                                    // [1]
  fe26 t2 = fe26_mul(p.z, q.z);                                    // [1]
  fe26 t3 = fe26_mul(fe26_add(p.x, p.y), fe26_add(q.x, q.y));      // [6]*[6] -> [1]
  t3 = fe26_add(t3, fe26_negate(fe26_add(t0, t1), 2));             // [4]
  fe26 t4 = fe26_mul(fe26_add(p.y, p.z), fe26_add(q.y, q.z));      // [1]
  t4 = fe26_add(t4, fe26_negate(fe26_add(t1, t2), 2));             // [4]
  fe26 y3 = fe26_mul(fe26_add(p.x, p.z), fe26_add(q.x, q.z));      // [1]
  y3 = fe26_add(y3, fe26_negate(fe26_add(t0, t2), 2));             // [4]
  t0 = fe26_mul_int(t0, 3);                                        // [3]
  t2 = fe26_mul_small_norm(t2, 21);                                // [1]
  fe26 z3 = fe26_add(t1, t2);                                      // [2]
  t1 = fe26_add(t1, fe26_negate(t2, 1));                           // [3]
  y3 = fe26_mul_small_norm(y3, 21);                                // [1]
  fe26 x3 = fe26_mul(t4, y3);                                      // [1]
  t2 = fe26_mul(t3, t1);                                           // [1]
  pt26 r;
  r.x = fe26_add(t2, fe26_negate(x3, 1));                          // [3]
  r.y = fe26_add(fe26_mul(t1, z3), fe26_mul(y3, t0));              // [2]
  r.z = fe26_add(fe26_mul(z3, t4), fe26_mul(t0, t3));              // [2]
  return r;
}

// Algorithm 9: 2p.  6 M + 2 S
S2K_DEV pt26 pt26_double(const pt26& p) {
  fe26 t0 = fe26_sqr(p.y);                                         // [1]
  fe26 z3 = fe26_mul_int(t0, 8);                                   // [8]
  fe26 t1 = fe26_mul(p.y, p.z);                                    // [1]
  fe26 t2 = fe26_mul_small_norm(fe26_sqr(p.z), 21);                // [1]
  fe26 x3 = fe26_mul(t2, z3);                                      // [1]
  fe26 y3 = fe26_add(t0, t2);                                      // [2]
  pt26 r;
  r.z = fe26_mul(t1, z3);                                          // [1]
  t0 = fe26_add(t0, fe26_negate(fe26_mul_int(t2, 3), 3));          // [5]
  r.y = fe26_add(x3, fe26_mul(t0, y3));                            // [2]
  r.x = fe26_mul_int(fe26_mul(t0, fe26_mul(p.x, p.y)), 2);         // [2]
  return r;
}

}  // namespace s2k

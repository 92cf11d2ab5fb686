// jacobian29.h — fast-path group law: Jacobian coordinates (x = X/Z^2, y = Y/Z^3), a = 0, over
// the lazy 9x29 field of fe29.h.
//
// The reference uses complete projective formulas everywhere (point_projective.go).  On the
// GPU fast path the cheaper incomplete Jacobian formulas are used instead:
//   doubling   3 M + 4 S   (vs 6 M + 2 S + m3b)
//   mixed add  8 M + 3 S   (vs 11 M + 2 m3b)
// They are wrong in exactly the exceptional cases (an operand at infinity, P + P, P - P), and
// in each of those the result has Z = 0, which then stays 0 through every later doubling
// and addition (Z3 = Y*Z, Z3 = Z*H).  So a lane whose final Z is 0 either hit an exceptional
// case or really ended at infinity; such lanes are re-done by the complete kernel
// (k_verify_fallback, RCB formulas), everything else is exact.  Results are therefore
// identical to the reference's for every input.
//
// Units (fe29.h) of every value are given in [brackets]; a product needs the product of its
// operands' units (summed over fused terms) <= 7.8.  Sums that are squared next (H, I, X3) would
// break that, so they are formed inside the preceding product's reduction (fe29_mul_plus /
// fe29_sqr_plus: the addend's limbs join the column sums) and come out carry-propagated.
// Invariant for points held in `jpt29`: x [1], y [<= 2], z [1].
#pragma once
#include "fe29.h"

namespace s2k {

struct jpt29 {
  fe29 x, y, z;
};

// 2P.  L = 3/2 X^2, S = Y^2, T = -X*S, X3 = L^2 + 2T, Y3 = -(L*(X3 + T) + S^2), Z3 = Y*Z
S2K_DEV jpt29 jpt29_double(const jpt29& p) {
  jpt29 r;
  r.z = fe29_mul(p.y, p.z);                                      // [2]*[1] -> [1]
  fe29 s = fe29_sqr(p.y);                                        // [2]^2 -> [1]
  fe29 l = fe29_sqr(p.x);                                        // [1]
  l = fe29_half(fe29_mul_int(l, 3));                             // [3] -> [2]
  fe29 t = fe29_mul(fe29_negate(s, 1), p.x);                     // [2]*[1] -> [1]
  r.x = fe29_sqr_plus(l, fe29_add(t, t));                        // [2]^2 + [2] -> [1]
  t = fe29_add(t, r.x);                                          // [2]
  r.y = fe29_negate(fe29_mul_add_sqr(t, l, s), 1);               // [2]*[2] + [1]^2, one reduction [1] -> [2]
  return r;
}

// P + (bx, by), the addend affine [bx 1, by <= 2] on the same curve.  `h_out` receives
// H = bx*Z1^2 - X1 [1] (Z3 = Z1*H), used when a table is brought to a common Z.
S2K_DEV jpt29 jpt29_add_affine(const jpt29& p, const fe29& bx, const fe29& by, fe29* h_out = nullptr) {
  fe29 zz = fe29_sqr(p.z);                                       // [1]
  fe29 nx = fe29_negate(p.x, 1);                                 // [2]   -X1
  fe29 h = fe29_mul_plus(bx, zz, nx);                            // [1]*[1] + [2] -> [1]   U2 - X1
  fe29 ns = fe29_negate(fe29_mul(by, zz), 1);                    // [2]*[1] -> [1] -> [2]   -by Z1^2
  fe29 i = fe29_mul_plus(ns, p.z, p.y);                          // [2]*[1] + [2] -> [1]   Y1 - S2
  jpt29 r;
  r.z = fe29_mul(p.z, h);                                        // [1]
  fe29 h2 = fe29_sqr(h);                                         // [1]   H^2
  fe29 h3 = fe29_mul(h2, fe29_negate(h, 1));                     // [1]*[2] -> [1]   -H^3
  fe29 t = fe29_mul(nx, h2);                                     // [2]*[1] -> [1]   -X1 H^2
  r.x = fe29_sqr_plus(i, fe29_add(fe29_add(h3, t), t));          // [1]^2 + [3] -> [1]
  t = fe29_add(t, r.x);                                          // [2]
  r.y = fe29_mul_add_mul(t, i, h3, p.y);                         // [2]*[1] + [1]*[2], one reduction -> [1]
  if (h_out) *h_out = h;
  return r;
}

// P + Q, both Jacobian.  (X2, Y2) is an affine point of the curve isomorphic by Z2 (neither formula here
// contains the curve constant), on which P reads (X1 Z2^2, Y1 Z2^3, Z1): one mixed addition there, and
// the result's Z times Z2 is back on secp256k1.  12 M + 4 S.  Z1 = 0 or Z2 = 0 gives Z3 = 0.
S2K_DEV jpt29 jpt29_add(const jpt29& p, const jpt29& q) {
  fe29 zz = fe29_sqr(q.z);                                       // [1]
  fe29 zzz = fe29_mul(zz, q.z);                                  // [1]
  jpt29 a;
  a.x = fe29_mul(p.x, zz);                                       // [1]
  a.y = fe29_mul(p.y, zzz);                                      // [2]*[1] -> [1]
  a.z = p.z;
  jpt29 r = jpt29_add_affine(a, q.x, q.y);
  r.z = fe29_mul(r.z, q.z);
  return r;
}

}  // namespace s2k

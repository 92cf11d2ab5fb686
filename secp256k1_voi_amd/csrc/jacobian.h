// jacobian.h — fast-path group law: Jacobian coordinates (x = X/Z^2, y = Y/Z^3), a = 0,
// over the lazy 10x26 field of fe26.h.
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
// Operation order and magnitudes follow the well-known sequences of libsecp256k1's
// gej_double / gej_add_ge_var, whose only requirement is that every fe26_mul / fe26_sqr input
// has magnitude <= 8; the magnitude of each value is given in [brackets].
// Invariant for points held in `jpt26`: x [<= 4], y [<= 4], z [1].
#pragma once
#include "fe26.h"

namespace s2k {

struct jpt26 {
  fe26 x, y, z;
};

// 2P.  L = 3/2 X^2, S = Y^2, T = -X*S, X3 = L^2 + 2T, Y3 = -(L*(X3 + T) + S^2), Z3 = Y*Z
S2K_DEV jpt26 jpt26_double(const jpt26& p) {
  jpt26 r;
  r.z = fe26_mul(p.y, p.z);                    // [1]
  fe26 s = fe26_sqr(p.y);                      // [1]
  fe26 l = fe26_sqr(p.x);                      // [1]
  l = fe26_half(fe26_mul_int(l, 3));           // [3] -> [2]
  fe26 t = fe26_mul(fe26_negate(s, 1), p.x);   // [2]*[4] -> [1]
  r.x = fe26_add(fe26_add(fe26_sqr(l), t), t); // [3]
  t = fe26_add(t, r.x);                        // [4]
  r.y = fe26_negate(fe26_mul_add_sqr(t, l, s), 1);     // t*l + s^2 with one reduction [1] -> [2]
  return r;
}

// P + (bx, by), the addend affine [bx 1, by <= 2] on the same curve.  `h_out` receives
// H = bx*Z1^2 - X1 [6] (Z3 = Z1*H), used when a table is brought to a common Z.
S2K_DEV jpt26 jpt26_add_affine(const jpt26& p, const fe26& bx, const fe26& by, fe26* h_out = nullptr) {
  fe26 zz = fe26_sqr(p.z);                                  // [1]
  fe26 u2 = fe26_mul(bx, zz);                               // [1]
  fe26 s2 = fe26_mul(fe26_mul(by, zz), p.z);                // [1]
  fe26 h = fe26_add(fe26_negate(p.x, 4), u2);               // [6]   U2 - X1
  fe26 i = fe26_add(fe26_negate(s2, 1), p.y);               // [6]   Y1 - S2
  jpt26 r;
  r.z = fe26_mul(p.z, h);                                   // [1]
  fe26 h2 = fe26_negate(fe26_sqr(h), 1);                    // [2]   -H^2
  fe26 h3 = fe26_mul(h2, h);                                // [1]   -H^3
  fe26 t = fe26_mul(p.x, h2);                               // [1]   -X1 H^2
  r.x = fe26_add(fe26_add(fe26_add(fe26_sqr(i), h3), t), t);   // [4]
  t = fe26_add(t, r.x);                                     // [5]
  r.y = fe26_mul_add_mul(t, i, h3, p.y);                    // [5]*[6] + [1]*[4], one reduction -> [1]
  if (h_out) *h_out = h;
  return r;
}

}  // namespace s2k

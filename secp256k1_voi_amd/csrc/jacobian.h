// jacobian.h — fast-path group law: Jacobian coordinates (x = X/Z^2, y = Y/Z^3), a = 0.
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
#pragma once
#include "fe.h"
#include "point.h"

namespace s2k {

struct jpt {
  fe x, y, z;
};

// 2P.  L = 3/2 X^2, S = Y^2, T = X*S, X3 = L^2 - 2T, Y3 = L*(T - X3) - S^2, Z3 = Y*Z
S2K_DEV jpt jpt_double(const jpt& p) {
  fe s = fe_sqr(p.y);
  fe l = fe_sqr(p.x);
  l = fe_half(fe_add(fe_add(l, l), l));
  fe t = fe_mul(p.x, s);
  jpt r;
  r.z = fe_mul(p.y, p.z);
  r.x = fe_sub(fe_sqr(l), fe_add(t, t));
  r.y = fe_sub(fe_mul(l, fe_sub(t, r.x)), fe_sqr(s));
  return r;
}

// P + (x2, y2) with the addend affine on the same curve.  `h_out` receives H = x2*Z1^2 - X1
// (Z3 = Z1*H), used when a table is brought to a common Z.
S2K_DEV jpt jpt_add_affine(const jpt& p, const fe& x2, const fe& y2, fe* h_out = nullptr) {
  fe zz = fe_sqr(p.z);
  fe u2 = fe_mul(x2, zz);
  fe s2 = fe_mul(fe_mul(y2, p.z), zz);
  fe h = fe_sub(u2, p.x);
  fe rr = fe_sub(s2, p.y);
  fe hh = fe_sqr(h);
  fe hhh = fe_mul(h, hh);
  fe v = fe_mul(p.x, hh);
  jpt r;
  r.z = fe_mul(p.z, h);
  r.x = fe_sub(fe_sub(fe_sqr(rr), hhh), fe_add(v, v));
  r.y = fe_sub(fe_mul(rr, fe_sub(v, r.x)), fe_mul(p.y, hhh));
  if (h_out) *h_out = h;
  return r;
}

}  // namespace s2k

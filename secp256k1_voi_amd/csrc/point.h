// point.h — secp256k1 group law on gfx950, one point per lane.
//
// Complete projective formulas of Renes–Costello–Batina (eprint 2015/1060) for a = 0,
// b3 = 21, as used by the reference (point_projective.go:24 addComplete, :123 addMixed,
// :208 doubleComplete).  Homogeneous coordinates (x = X/Z, y = Y/Z), identity (0:1:0)
// (point.go:31-49).  These are exception-free, so all lanes run the same instruction
// stream whatever their data: the vartime "skip on zero digit" branches of the reference
// (point_mul_table.go:44-46) become adds of the identity or selects.
// Multiplication by b3 is a one-word multiply (fe_mul_small), not a full field multiply.
#pragma once
#include "fe.h"

namespace s2k {

struct pt {
  fe x, y, z;
};
struct apt {
  fe x, y;
};

__device__ static const uint32_t FE_BETA[8] = {0x719501eeu, 0xc1396c28u, 0x12f58995u, 0x9cf04975u,
                                               0xac3434e9u, 0x6e64479eu, 0x657c0710u, 0x7ae96a2bu};   // point_mul_glv.go:44
__device__ static const uint32_t FE_GX[8] = {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu,
                                             0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu};     // point.go:18
__device__ static const uint32_t FE_GY[8] = {0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u,
                                             0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u};     // point.go:20

S2K_DEV fe fe_from_limbs(const uint32_t* p) {
  fe r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = p[i];
  return r;
}
S2K_DEV pt pt_identity() {   // point.go:42
  pt r;
  r.x = fe_zero();
  r.y = fe_from_u32(1);
  r.z = fe_zero();
  return r;
}
S2K_DEV pt pt_from_affine(const apt& a) {
  pt r;
  r.x = a.x;
  r.y = a.y;
  r.z = fe_from_u32(1);
  return r;
}
S2K_DEV bool pt_is_identity(const pt& p) { return fe_is_zero(p.z); }   // point.go:148
S2K_DEV pt pt_select(bool pick_b, const pt& a, const pt& b) {
  pt r;
  r.x = fe_select(pick_b, a.x, b.x);
  r.y = fe_select(pick_b, a.y, b.y);
  r.z = fe_select(pick_b, a.z, b.z);
  return r;
}
S2K_DEV pt pt_cond_neg(const pt& p, bool neg) {   // point.go:89,99
  pt r = p;
  r.y = fe_select(neg, p.y, fe_neg(p.y));
  return r;
}

// y^2 == x^3 + 7  (xyOnCurve / maybeYY, point_s11n.go:298-307)
S2K_DEV fe fe_curve_rhs(const fe& x) { return fe_add(fe_mul(fe_sqr(x), x), fe_from_u32(7)); }
S2K_DEV bool apt_on_curve(const apt& a) { return fe_eq(fe_sqr(a.y), fe_curve_rhs(a.x)); }

// RCB'15 Algorithm 7 (point_projective.go:24-120): 12 M + 2 m3b + 19 a
S2K_DEV pt pt_add_complete(const pt& p, const pt& q) {
  fe t0 = fe_mul(p.x, q.x);
  fe t1 = fe_mul(p.y, q.y);
  fe t2 = fe_mul(p.z, q.z);
  fe t3 = fe_mul(fe_add(p.x, p.y), fe_add(q.x, q.y));
  t3 = fe_sub(t3, fe_add(t0, t1));
  fe t4 = fe_mul(fe_add(p.y, p.z), fe_add(q.y, q.z));
  t4 = fe_sub(t4, fe_add(t1, t2));
  fe y3 = fe_mul(fe_add(p.x, p.z), fe_add(q.x, q.z));
  y3 = fe_sub(y3, fe_add(t0, t2));
  fe x3 = fe_add(t0, t0);
  t0 = fe_add(x3, t0);
  t2 = fe_mul_small(t2, 21);
  fe z3 = fe_add(t1, t2);
  t1 = fe_sub(t1, t2);
  y3 = fe_mul_small(y3, 21);
  x3 = fe_mul(t4, y3);
  t2 = fe_mul(t3, t1);
  x3 = fe_sub(t2, x3);
  y3 = fe_mul(y3, t0);
  t1 = fe_mul(t1, z3);
  y3 = fe_add(t1, y3);
  t0 = fe_mul(t0, t3);
  z3 = fe_mul(z3, t4);
  z3 = fe_add(z3, t0);
  pt r;
  r.x = x3;
  r.y = y3;
  r.z = z3;
  return r;
}
// RCB'15 Algorithm 8 (point_projective.go:123-205): 11 M + 2 m3b + 13 a.
// The addend must not be the identity; p may be anything.
S2K_DEV pt pt_add_mixed(const pt& p, const apt& q) {
  fe t0 = fe_mul(p.x, q.x);
  fe t1 = fe_mul(p.y, q.y);
  fe t3 = fe_mul(fe_add(q.x, q.y), fe_add(p.x, p.y));
  t3 = fe_sub(t3, fe_add(t0, t1));
  fe t4 = fe_add(fe_mul(q.y, p.z), p.y);
  fe y3 = fe_add(fe_mul(q.x, p.z), p.x);
  fe x3 = fe_add(t0, t0);
  t0 = fe_add(x3, t0);
  fe t2 = fe_mul_small(p.z, 21);
  fe z3 = fe_add(t1, t2);
  t1 = fe_sub(t1, t2);
  y3 = fe_mul_small(y3, 21);
  x3 = fe_mul(t4, y3);
  t2 = fe_mul(t3, t1);
  x3 = fe_sub(t2, x3);
  y3 = fe_mul(y3, t0);
  t1 = fe_mul(t1, z3);
  y3 = fe_add(t1, y3);
  t0 = fe_mul(t0, t3);
  z3 = fe_mul(z3, t4);
  z3 = fe_add(z3, t0);
  pt r;
  r.x = x3;
  r.y = y3;
  r.z = z3;
  return r;
}
// RCB'15 Algorithm 9 (point_projective.go:208-273): 6 M + 2 S + 1 m3b + 9 a
S2K_DEV pt pt_double_complete(const pt& p) {
  fe t0 = fe_sqr(p.y);
  fe z3 = fe_add(t0, t0);
  z3 = fe_add(z3, z3);
  z3 = fe_add(z3, z3);
  fe t1 = fe_mul(p.y, p.z);
  fe t2 = fe_sqr(p.z);
  t2 = fe_mul_small(t2, 21);
  fe x3 = fe_mul(t2, z3);
  fe y3 = fe_add(t0, t2);
  z3 = fe_mul(t1, z3);
  t1 = fe_add(t2, t2);
  t2 = fe_add(t1, t2);
  t0 = fe_sub(t0, t2);
  y3 = fe_mul(t0, y3);
  y3 = fe_add(x3, y3);
  t1 = fe_mul(p.x, p.y);
  x3 = fe_mul(t0, t1);
  x3 = fe_add(x3, x3);
  pt r;
  r.x = x3;
  r.y = y3;
  r.z = z3;
  return r;
}

// rescale to Z = 1 (point_projective.go:278-302); returns false for the identity
S2K_DEV bool pt_to_affine(apt& out, const pt& p) {
  fe zi = fe_inv(p.z);
  out.x = fe_normalize(fe_mul(p.x, zi));
  out.y = fe_normalize(fe_mul(p.y, zi));
  return !pt_is_identity(p);
}

}  // namespace s2k

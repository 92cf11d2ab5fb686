// xyzz29.h — mixed addition in extended Jacobian ("XYZZ") coordinates, x = X/ZZ, y = Y/ZZZ with ZZ^3 = ZZZ^2, over the
// lazy 9x29 field: the bucket pass of the multi-scalar multiplication (msm.hip).
//
// The reference adds complete projective points everywhere (point_projective.go:123 addMixed: 11 products + 2 by b3);
// a bucket sum is hundreds of millions of additions of an AFFINE input point to an accumulator, and the incomplete
// XYZZ mixed addition (madd-2008-s) needs 8 products + 2 squarings in 9 reductions: 918 multiply-adds instead of 1072
// and none of the small multiplications, about 13 % fewer instructions.  It is wrong in exactly the exceptional cases -
// the accumulator and the input share their x (P + P, P - P) - and in each of those the result has ZZ = ZZZ = 0, which
// then stays 0 through every later addition (ZZ3 = ZZ1 * P^2).  A piece of a bucket whose final ZZ is 0 has either met an
// exceptional case or really sums to the identity; such pieces are re-done with the complete formulas
// (k_msm_accumulate_redo), everything else is exact - the same device as the verification ladder's sticky Z
// (jacobian29.h).  The identity has no XYZZ form here: an accumulator STARTS as its first point.
//
// Units (fe29.h) in [brackets]; a product needs the product of its operands' units (summed over fused terms) <= 7.8.
// Invariant of an xyzz29: x [1], y [<= 2], zz [1], zzz [1].
#pragma once
#include "fe29.h"
#include "jacobian29.h"
#include "pt29.h"

namespace s2k {

struct xyzz29 {
  fe29 x, y, zz, zzz;
};

// (bx, by) affine [bx 1, by <= 2] -> accumulator
S2K_DEV xyzz29 xyzz29_from_affine(const fe29& bx, const fe29& by) {
  xyzz29 r;
  r.x = bx;
  r.y = by;
  r.zz = fe29_one();
  r.zzz = fe29_one();
  return r;
}

// p + (bx, by), the addend affine [bx 1, by <= 2] and not the identity.  Same x as p: ZZ3 = ZZZ3 = 0 (see above).
S2K_DEV xyzz29 xyzz29_add_affine(const xyzz29& p, const fe29& bx, const fe29& by) {
  const fe29 nx = fe29_negate(p.x, 1);                                 // [2]   -X1
  const fe29 pp_ = fe29_mul_plus(bx, p.zz, nx);                        // [1]*[1] + [2] -> [1]   P = U2 - X1
  const fe29 rn = fe29_mul_plus(fe29_negate(by, 2), p.zzz, p.y);       // [3]*[1] + [2] -> [1]   -R = Y1 - S2
  const fe29 pp = fe29_sqr(pp_);                                       // [1]   P^2
  const fe29 pppn = fe29_mul(pp, fe29_negate(pp_, 1));                 // [1]*[2] -> [1]   -P^3
  const fe29 qn = fe29_mul(nx, pp);                                    // [2]*[1] -> [1]   -Q = -X1 P^2
  xyzz29 r;
  r.x = fe29_sqr_plus(rn, fe29_add(fe29_add(pppn, qn), qn));           // [1]^2 + [3] -> [1]   R^2 - P^3 - 2 Q
  const fe29 t = fe29_add(qn, r.x);                                    // [2]   X3 - Q
  r.y = fe29_mul_add_mul(t, rn, pppn, p.y);                            // [2]*[1] + [1]*[2] -> [1]   R (Q - X3) - Y1 P^3
  r.zz = fe29_mul(p.zz, pp);                                           // [1]
  r.zzz = fe29_mul(p.zzz, fe29_negate(pppn, 1));                       // [1]*[2] -> [1]
  return r;
}

// the same point in Jacobian coordinates (x = X'/Z'^2, y = Y'/Z'^3) with Z' = ZZ: (X ZZ, Y ZZZ, ZZ); ZZ = 0 gives Z' = 0
S2K_DEV jpt29 xyzz29_to_jacobian(const xyzz29& p) {
  jpt29 r;
  r.x = fe29_mul(p.x, p.zz);
  r.y = fe29_mul(p.y, p.zzz);     // [2]*[1]
  r.z = p.zz;
  return r;
}
// and back: (X, Y, Z^2, Z^3); Z = 0 gives ZZ = ZZZ = 0
S2K_DEV xyzz29 xyzz29_from_jacobian(const jpt29& p) {
  xyzz29 r;
  r.x = p.x;
  r.y = p.y;
  r.zz = fe29_sqr(p.z);
  r.zzz = fe29_mul(r.zz, p.z);
  return r;
}

// x = X/ZZ, y = Y/ZZZ as the projective point (X ZZZ : Y ZZ : ZZ ZZZ); ZZ = 0 gives Z = 0
S2K_DEV pt29 xyzz29_to_pt29(const xyzz29& p) {
  pt29 r;
  r.x = fe29_mul(p.x, p.zzz);
  r.y = fe29_mul(p.y, p.zz);      // [2]*[1]
  r.z = fe29_mul(p.zz, p.zzz);
  return r;
}

}  // namespace s2k

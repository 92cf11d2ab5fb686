// pt29q.h — the complete projective group law of pt29.h (Renes–Costello–Batina, a = 0, b3 = 21; the reference's
// addComplete / doubleComplete, point_projective.go:24,208) spread over the lanes of a QUAD.
//
// Where it is used the group law is a serial chain on a chip that is otherwise idle (bucket reduction, tree and Horner
// tail of the multi-scalar multiplication, msm.hip): a lone lane's addition is ~1800 dependent instructions, and a lone
// wave issues a multiply-add only every 8 cycles.  Here one point lives in four adjacent lanes, coordinate by
// coordinate - lane 0 of the quad holds X, lane 1 Y, lane 2 Z, lane 3 a copy of Z that takes the fourth product of a
// doubling - and every layer of independent products of the formulas is ONE product executed by the quad's lanes in
// parallel; operands travel between the lanes with DPP quad permutes (v_mov_b32_dpp quad_perm: each lane reads the
// same register of any lane of its quad).  An addition is 3 product rounds (a plain product, a product with an addend,
// a two-product sum) + ~330 linear / move instructions instead of 12 products, a doubling 2 rounds instead of 8:
// about 2.2x / 1.9x fewer dependent instructions.
//
// Invariant of a value of type fe29 "coordinate of a pt29q": 1 unit (fe29.h), like the coordinates of a pt29.
// Lane 3 computes along (no divergence); what it holds is never read by lanes 0..2 except where stated.
#pragma once
#include "pt29.h"

namespace s2k {

// quad_perm control word: lane i of every quad reads lane s_i
#define S2K_QP(s0, s1, s2, s3) ((s0) | ((s1) << 2) | ((s2) << 4) | ((s3) << 6))

template <int CTRL>
S2K_DEV fe29 fe29_qperm(const fe29& a) {
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.n[i], CTRL, 0xF, 0xF, true);
  return r;
}
S2K_DEV fe29 fe29_pick(bool pick_b, const fe29& a, const fe29& b) {   // lane-constant condition: one v_cndmask per limb
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.n[i] = pick_b ? b.n[i] : a.n[i];
  return r;
}
// a * k, k < 2^6 a per-lane value (fe29_mul_small_norm takes a uniform constant; same arithmetic)
S2K_DEV fe29 fe29_mul_small_lane(const fe29& a, uint32_t k) {
  fe29 r;
  uint64_t c = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a.n[i]), "v"(k) : "vcc");
    r.n[i] = (uint32_t)c & F29_M;
    c >>= 29;
  }
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a.n[8]), "v"(k) : "vcc");
  r.n[8] = (uint32_t)c & F29_M8;
  uint32_t x = (uint32_t)(c >> 24);
  r.n[0] += x * 0x3D1u;
  r.n[1] += x << 3;
  return r;
}

// this lane's coordinate of the identity (0 : 1 : 0); q = lane & 3
S2K_DEV fe29 pt29q_identity(uint32_t q) {
  fe29 r = fe29_zero();
  r.n[0] = q == 1 ? 1u : 0u;
  return r;
}
// this lane's coordinate of a point every lane holds in full
S2K_DEV fe29 pt29q_from(const pt29& p, uint32_t q) { return fe29_pick(q >= 2, fe29_pick(q == 1, p.x, p.y), p.z); }
// the full point in every lane
S2K_DEV pt29 pt29q_gather(const fe29& pc) {
  pt29 r;
  r.x = fe29_qperm<S2K_QP(0, 0, 0, 0)>(pc);
  r.y = fe29_qperm<S2K_QP(1, 1, 1, 1)>(pc);
  r.z = fe29_qperm<S2K_QP(2, 2, 2, 2)>(pc);
  return r;
}

// Algorithm 7 (pt29_add): p + q, no exceptions.  pc, qc: this lane's coordinates of p and q.
S2K_DEV fe29 pt29q_add(const fe29& pc, const fe29& qc, uint32_t q) {
  // round 1: t0 = X1 X2 | t1 = Y1 Y2 | t2 = Z1 Z2                                                      [1]
  const fe29 T = fe29_mul(pc, qc);
  // round 2: t3 = (X1 + Y1)(X2 + Y2) - t0 - t1 | t4 = (Y1 + Z1)(Y2 + Z2) - t1 - t2 | y3 = (Z1 + X1)(Z2 + X2) - t2 - t0
  constexpr int ROT = S2K_QP(1, 2, 0, 3);          // lane 0 <- 1, 1 <- 2, 2 <- 0
  const fe29 a = fe29_add(pc, fe29_qperm<ROT>(pc)), b = fe29_add(qc, fe29_qperm<ROT>(qc));              // [2]
  const fe29 e = fe29_negate(fe29_add(T, fe29_qperm<ROT>(T)), 2);                                       // [3]
  const fe29 U = fe29_mul_plus(a, b, e);                                                                // [2]*[2] + [3] -> [1]
  // scale: T' = 3 t0 | t1 | 21 t2;  U' = t3 | t4 | 21 y3                                                [1]
  const fe29 Ts = fe29_mul_small_lane(T, q == 0 ? 3u : (q == 1 ? 1u : 21u));
  const fe29 Us = fe29_mul_small_lane(U, q >= 2 ? 21u : 1u);
  const fe29 t1 = fe29_qperm<S2K_QP(1, 1, 1, 1)>(Ts), t2 = fe29_qperm<S2K_QP(2, 2, 2, 2)>(Ts);
  // round 3, lane 0: X3 = t3 (t1 - t2) - t4 y3 | lane 1: Z3 = t4 (t1 + t2) + t3 t0' | lane 2: Y3 = (t1 + t2)(t1 - t2) + y3 t0'
  const fe29 V = fe29_add(t1, fe29_cond_negate1(t2, q != 1));   // lanes 0, 2: t1 - t2 [3]; lane 1: t1 + t2 [2]
  const fe29 W = fe29_add(t1, t2);                               // [2]
  const fe29 A = fe29_pick(q >= 2, Us, W);                       // t3 | t4 | t1 + t2
  const fe29 C = fe29_qperm<S2K_QP(1, 0, 2, 3)>(Us);             // t4 | t3 | y3
  const fe29 Dr = fe29_pick(q >= 2, Ts, fe29_negate(Us, 1));     // lane 0: t0', lane 2: -y3 [2]
  const fe29 D = fe29_qperm<S2K_QP(2, 0, 0, 3)>(Dr);             // -y3 | t0' | t0'
  const fe29 R = fe29_mul_add_mul(A, V, C, D);                   // [1][3] + [1][2] | [1][2] + [1][1] | [2][3] + [1][1]  -> [1]
  return fe29_qperm<S2K_QP(0, 2, 1, 1)>(R);                      // X3 | Y3 | Z3 | Z3
}

// Algorithm 9 (pt29_double): 2p.
S2K_DEV fe29 pt29q_double(const fe29& pc, uint32_t q) {
  // round 1: X Y | Y^2 | Z^2 | Y Z                                                                      [1]
  const fe29 P = fe29_mul(fe29_qperm<S2K_QP(0, 1, 2, 1)>(pc), fe29_qperm<S2K_QP(1, 1, 2, 2)>(pc));
  const fe29 t0 = fe29_qperm<S2K_QP(1, 1, 1, 1)>(P), zz = fe29_qperm<S2K_QP(2, 2, 2, 2)>(P);
  const fe29 z3 = fe29_mul_int(fe29_normalize_weak(fe29_mul_int(t0, 4)), 2);                            // [2]   8 Y^2
  const fe29 t2 = fe29_mul_small_norm(zz, 21);                                                          // [1]   b3 Z^2
  const fe29 y3 = fe29_add(t0, t2);                                                                     // [2]
  const fe29 t0m = fe29_normalize_weak(fe29_add(t0, fe29_negate(fe29_mul_small_norm(zz, 63), 1)));      // [1]   Y^2 - 3 b3 Z^2
  // round 2: X3 = (X Y)(2 t0m) | t2 z3 | Z3 = z3 (Y Z) | t0m y3;   Y3 = lane 1 + lane 3
  const fe29 A = fe29_pick(q == 3, fe29_pick(q == 0, z3, P), y3);                                        // XY | z3 | z3 | y3
  const fe29 B = fe29_pick(q == 3, fe29_pick(q == 2, fe29_pick(q == 0, t2, fe29_mul_int(t0m, 2)), fe29_qperm<S2K_QP(3, 3, 3, 3)>(P)), t0m);
  const fe29 R = fe29_mul(A, B);                                                                        // [1][2] | [2][1] | [2][1] | [2][1] -> [1]
  const fe29 S = fe29_normalize_weak(fe29_add(R, fe29_qperm<S2K_QP(0, 3, 2, 3)>(R)));                    // lane 1: Y3 [2] -> [1]
  return fe29_qperm<S2K_QP(0, 1, 2, 2)>(fe29_pick(q == 1, R, S));                                        // X3 | Y3 | Z3 | Z3
}

}  // namespace s2k

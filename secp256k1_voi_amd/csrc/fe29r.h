// fe29r.h — GF(p) in the 9x29 lazy form of fe29.h with ONE LIMB PER LANE of a 16-lane DPP row, and pt29r: the complete
// projective group law (Renes–Costello–Batina, a = 0, b3 = 21; the reference's addComplete / doubleComplete,
// point_projective.go:24,208) with one field product per row, four products per wavefront.
//
// Why: the tail of the multi-scalar multiplication (msm.hip: the small levels of the bucket reduction, the Horner
// recurrence over the windows — 112 doublings that depend on each other) is a chain of group operations on a chip with
// nothing else to do.  What bounds it is the number of instructions ONE wave has to issue per operation (a lone wave
// issues a multiply-add every 8 cycles): 1070 for a doubling on one lane (pt29.h), 550 on the four lanes of a quad
// (pt29q.h), ~160 here.  A product is nine steps "broadcast limb i of a (row_newbcast:i), shift b by i lanes (row_shr:i),
// multiply-add": lane j collects column j of the schoolbook product; columns 9..16 are folded with 2^261 = R1 2^29 + R0 by
// two more multiply-adds on lane-shifted copies (row_shl:9, row_shl:8), carries move one lane up (row_shr:1).  No LDS,
// no memory.  Additions, negations, small multiples and selects are ONE instruction (an element is one register).
// The four products of a layer of the formulas sit in the four rows; operands reach the rows by v_cndmask on the row
// number, results are spread to all rows by v_permlane16_swap / v_permlane32_swap (gfx950).
//
// Invariants.  A value of type `fer` holds limb j of the element in lane j of every row it is valid in, lanes 9..15 of a
// row are ZERO.  "One unit": what a reduction leaves, limb 0 <= 2^29 + 2^26, limb 1 <= 2^29 + 2^19, limbs 2..7 <= 2^29 +
// 2^18, limb 8 <= 2^24 + 2^18; a product (sum of products, plus addend) needs the units of its operands multiplied and
// summed <= 7, as in fe29.h.  tests/fer_model.py is this file lane by lane in Python with every 32- / 64-bit width
// asserted (tests/test_fer_model.py: random, lazy and extremal inputs, the group law against the affine one);
// tests/test_gpu_round5.py runs the compiled functions through the C-ABI (S2K_HP_PT29R_*).
#pragma once
#include "fe29.h"
#include "pt29.h"

namespace s2k {

typedef uint32_t fer;

template <int CTRL>
S2K_DEV uint32_t fer_dpp(uint32_t v) {   // out-of-row sources read as 0 (bound_ctrl:1)
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
#define S2K_ROW_SHL(n) (0x100 + (n))
#define S2K_ROW_SHR(n) (0x110 + (n))
#define S2K_ROW_BCAST(n) (0x150 + (n))   // row_newbcast:n (gfx90a+): every lane of a row reads lane n of the row

// the per-lane constants, set up once per kernel from the lane id
struct fer_consts {
  uint32_t row;        // lane >> 4 (of the wave)
  uint32_t j;          // lane & 15
  uint32_t kR0, kR1, k16, k17, kFold2, kTop, wmask, shift, le8, lt8, p;
};
S2K_DEV fer_consts fer_setup(uint32_t lane_in_wave) {
  fer_consts k;
  k.row = (lane_in_wave >> 4) & 3u;
  k.j = lane_in_wave & 15u;
  const uint32_t j = k.j;
  k.kR0 = j <= 6 ? F29_R0 : 0u;                        // column j + 9 (<= 15) -> column j
  k.kR1 = (j >= 1 && j <= 7) ? F29_R1 : 0u;            // column j + 8 (9 .. 15) -> column j
  k.k16 = j == 7 ? F29_R0 : (j == 8 ? F29_R1 : 0u);    // column 16 (row-uniform) -> columns 7, 8
  k.k17 = j == 8 ? F29_R0 : (j == 9 ? F29_R1 : 0u);    // column 17 -> columns 8, 9 (column 9 is folded once more)
  k.kFold2 = j == 0 ? F29_R0 : (j == 1 ? F29_R1 : 0u);
  k.kTop = j == 0 ? 0x3D1u : (j == 1 ? 8u : 0u);       // 2^256 = 0x3D1 + 8 * 2^29
  k.wmask = j < 8 ? F29_M : (j == 8 ? F29_M8 : 0u);
  k.shift = j < 8 ? 29u : 24u;
  k.le8 = j <= 8 ? 0xFFFFFFFFu : 0u;
  k.lt8 = j < 8 ? 0xFFFFFFFFu : 0u;
  k.p = j == 0 ? F29_P0 : (j == 1 ? F29_P1 : (j < 8 ? F29_PM : (j == 8 ? F29_P8 : 0u)));
  return k;
}

S2K_DEV fer fer_add(fer a, fer b) { return a + b; }
// -a for a of at most w units; w + 1 units
S2K_DEV fer fer_negate(fer a, uint32_t w, const fer_consts& k) { return k.p * (w + 1u) - a; }
// row r of the result = row r of the r-th argument
S2K_DEV fer fer_sel(const fer_consts& k, fer r0, fer r1, fer r2, fer r3) {
  const fer lo = k.row == 0 ? r0 : r1, hi = k.row == 2 ? r2 : r3;
  return k.row < 2 ? lo : hi;
}
S2K_DEV fer fer_sel2(const fer_consts& k, fer r01, fer r23) { return k.row < 2 ? r01 : r23; }

// the four rows of v, each spread to all rows: v_permlane16_swap exchanges the odd rows of its first operand with the even
// rows of its second, v_permlane32_swap rows 2, 3 of the first with rows 0, 1 of the second
typedef unsigned int fer_u2 __attribute__((ext_vector_type(2)));
S2K_DEV void fer_pairs(fer v, fer& even, fer& odd) {            // even = (v0, v0, v2, v2), odd = (v1, v1, v3, v3)
  const fer_u2 s = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  even = s.x;
  odd = s.y;
}
S2K_DEV void fer_halves(fer v, fer& lo, fer& hi) {              // v = (a, a, b, b) -> lo = a everywhere, hi = b everywhere
  const fer_u2 s = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  lo = s.x;
  hi = s.y;
}
S2K_DEV void fer_bcast_rows(fer v, fer& r0, fer& r1, fer& r2, fer& r3) {
  fer e, o;
  fer_pairs(v, e, o);
  fer_halves(e, r0, r2);
  fer_halves(o, r1, r3);
}

// carry pass of 64-bit lane values (lanes 0..8 meaningful) to one unit: limbs 0..7 keep 29 bits, limb 8 keeps 24, every
// carry moves one lane up, what leaves limb 8 comes back through 2^256 = 0x3D1 + 8 * 2^29 (top carry < 2^22)
S2K_DEV fer fer_carry(uint64_t r, const fer_consts& k) {
  const uint32_t lo = (uint32_t)r & k.wmask;
  const uint32_t cc = (uint32_t)(r >> k.shift);
  const uint32_t x = fer_dpp<S2K_ROW_BCAST(8)>(cc);
  const uint32_t up = fer_dpp<S2K_ROW_SHR(1)>(cc & k.lt8);
  return (lo + up + x * k.kTop) & k.le8;
}
S2K_DEV fer fer_norm(fer a, const fer_consts& k) { return fer_carry((uint64_t)a, k); }
// a * m for a per-lane small m (< 2^7), one unit out
S2K_DEV fer fer_small_norm(fer a, uint32_t m, const fer_consts& k) {
  uint64_t r = 0;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(a), "v"(m) : "vcc");
  return fer_carry(r, k);
}

template <int I>
S2K_DEV void fer_mul_step(uint64_t& acc, fer a, fer b) {
  const uint32_t ai = fer_dpp<S2K_ROW_BCAST(I)>(a);
  const uint32_t bs = I == 0 ? b : fer_dpp<S2K_ROW_SHR(I == 0 ? 1 : I)>(b);
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(ai), "v"(bs) : "vcc");
}
// columns 0..15 of a * b into the lanes' acc, column 16 (a8 * b8) into h
S2K_DEV void fer_mul_acc(uint64_t& acc, uint64_t& h, fer a, fer b) {
  fer_mul_step<0>(acc, a, b);
  fer_mul_step<1>(acc, a, b);
  fer_mul_step<2>(acc, a, b);
  fer_mul_step<3>(acc, a, b);
  fer_mul_step<4>(acc, a, b);
  fer_mul_step<5>(acc, a, b);
  fer_mul_step<6>(acc, a, b);
  fer_mul_step<7>(acc, a, b);
  fer_mul_step<8>(acc, a, b);
  const uint32_t a8 = fer_dpp<S2K_ROW_BCAST(8)>(a), b8 = fer_dpp<S2K_ROW_BCAST(8)>(b);
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(h) : "v"(a8), "v"(b8) : "vcc");
}
// columns -> one unit (fer_model.py _tail)
S2K_DEV fer fer_mul_tail(uint64_t acc, uint64_t h, const fer_consts& k) {
  const uint32_t lo = (uint32_t)acc & F29_M;
  const uint32_t c1 = (uint32_t)(acc >> 29) & F29_M;
  const uint32_t c2 = (uint32_t)(acc >> 58);
  const uint32_t t = lo + fer_dpp<S2K_ROW_SHR(1)>(c1) + fer_dpp<S2K_ROW_SHR(2)>(c2);                     // < 2^30 + 64
  const uint32_t t16 = ((uint32_t)h & F29_M) + fer_dpp<S2K_ROW_BCAST(15)>(c1) + fer_dpp<S2K_ROW_BCAST(14)>(c2);
  const uint32_t t17 = (uint32_t)(h >> 29) + fer_dpp<S2K_ROW_BCAST(15)>(c2);                             // < 2^24
  // fold: column c >= 9 goes to column c - 9 times R0 and to column c - 8 times R1.  Columns 9..15 come from the lanes
  // (row_shl), columns 16 and 17 are row-uniform values times per-lane constants (R0 in lane 7 / 8, R1 in lane 8 / 9)
  const uint32_t hi9 = fer_dpp<S2K_ROW_SHL(9)>(t), hi8 = fer_dpp<S2K_ROW_SHL(8)>(t);
  uint64_t r = t & k.le8;
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(hi9), "v"(k.kR0) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(hi8), "v"(k.kR1) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(t16), "v"(k.k16) : "vcc");
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(t17), "v"(k.k17) : "vcc");
  const uint32_t r9 = fer_dpp<S2K_ROW_BCAST(9)>((uint32_t)r);        // column 9 = R1 * t17 (< 2^32): folded once more
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r) : "v"(r9), "v"(k.kFold2) : "vcc");
  return fer_carry(r, k);
}
S2K_DEV fer fer_mul(fer a, fer b, const fer_consts& k) {
  uint64_t acc = 0, h = 0;
  fer_mul_acc(acc, h, a, b);
  return fer_mul_tail(acc, h, k);
}
S2K_DEV fer fer_mul_plus(fer a, fer b, fer e, const fer_consts& k) {        // a * b + e
  uint64_t acc = e, h = 0;
  fer_mul_acc(acc, h, a, b);
  return fer_mul_tail(acc, h, k);
}
S2K_DEV fer fer_mul_add_mul(fer a, fer b, fer c, fer d, const fer_consts& k) {   // a * b + c * d
  uint64_t acc = 0, h = 0;
  fer_mul_acc(acc, h, a, b);
  fer_mul_acc(acc, h, c, d);
  return fer_mul_tail(acc, h, k);
}

// ---- points: X, Y, Z each in all four rows -------------------------------------------------------------------------
struct pt29r {
  fer x, y, z;     // x, z one unit; y up to two (what a doubling leaves; the addition normalises its inputs' y)
};
S2K_DEV pt29r pt29r_identity(const fer_consts& k) {
  pt29r r;
  r.x = 0;
  r.y = k.j == 0 ? 1u : 0u;
  r.z = 0;
  return r;
}
// from a point every lane holds in full / back (lane j of a row takes limb j)
S2K_DEV fer fer_from_fe29(const fe29& a, const fer_consts& k) {
  fer r = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) r = k.j == (uint32_t)i ? a.n[i] : r;
  return r;
}
S2K_DEV fe29 fer_to_fe29(fer a) {       // row-wise: every lane of a row gets the row's element
  fe29 r;
  r.n[0] = fer_dpp<S2K_ROW_BCAST(0)>(a);
  r.n[1] = fer_dpp<S2K_ROW_BCAST(1)>(a);
  r.n[2] = fer_dpp<S2K_ROW_BCAST(2)>(a);
  r.n[3] = fer_dpp<S2K_ROW_BCAST(3)>(a);
  r.n[4] = fer_dpp<S2K_ROW_BCAST(4)>(a);
  r.n[5] = fer_dpp<S2K_ROW_BCAST(5)>(a);
  r.n[6] = fer_dpp<S2K_ROW_BCAST(6)>(a);
  r.n[7] = fer_dpp<S2K_ROW_BCAST(7)>(a);
  r.n[8] = fer_dpp<S2K_ROW_BCAST(8)>(a);
  return r;
}
S2K_DEV pt29r pt29r_from(const pt29& p, const fer_consts& k) {
  pt29r r;
  r.x = fer_from_fe29(p.x, k);
  r.y = fer_from_fe29(p.y, k);
  r.z = fer_from_fe29(p.z, k);
  return r;
}
S2K_DEV pt29 pt29r_gather(const pt29r& p, const fer_consts& k) {
  pt29 r;
  r.x = fer_to_fe29(p.x);
  r.y = fer_to_fe29(fer_norm(p.y, k));
  r.z = fer_to_fe29(p.z);
  return r;
}
// a point stored in planes [27][stride] (msm.hip pt_store): lane j < 9 of every row loads limb j of each coordinate
S2K_DEV pt29r pt29r_load(const uint32_t* __restrict__ base, size_t stride, size_t slot, const fer_consts& k) {
  pt29r r;
  const bool on = k.j <= 8;
  const size_t o = (size_t)(on ? k.j : 0u) * stride + slot;
  r.x = on ? base[o] : 0u;
  r.y = on ? base[o + 9 * stride] : 0u;
  r.z = on ? base[o + 18 * stride] : 0u;
  return r;
}
S2K_DEV void pt29r_store(uint32_t* __restrict__ base, size_t stride, size_t slot, const pt29r& p, const fer_consts& k) {
  const fer yn = fer_norm(p.y, k);         // (lane exchanges: by all lanes, before the branch)
  if (k.row != 0 || k.j > 8) return;       // row 0 writes
  const size_t o = (size_t)k.j * stride + slot;
  base[o] = p.x;
  base[o + 9 * stride] = yn;
  base[o + 18 * stride] = p.z;
}

// Algorithm 9 (pt29_double): 2p.  Rows of round 1: X Y | Y^2 | Y Z | Z^2;  round 2: X3 = (XY)(2 t0m) | (21 ZZ)(8 YY) |
// Z3 = (8 YY)(YZ) | t0m y3, Y3 = row 1 + row 3 (two units), with t0m = YY - 63 ZZ, y3 = YY + 21 ZZ.
S2K_DEV pt29r pt29r_double(const pt29r& p, const fer_consts& k) {
  const fer P1 = fer_mul(fer_sel(k, p.x, p.y, p.y, p.z), fer_sel2(k, p.y, p.z), k);
  fer e, o, YY, ZZ;
  fer_pairs(P1, e, o);                       // e = (XY, XY, YZ, YZ), o = (YY, YY, ZZ, ZZ)
  fer_halves(o, YY, ZZ);
  const uint32_t m1 = k.row == 1 ? 21u : (k.row == 2 ? 4u : 63u), m2 = k.row == 1 ? 4u : (k.row == 3 ? 21u : 0u);
  const fer op1 = fer_small_norm(k.row == 2 ? YY : ZZ, m1, k);      // 63 ZZ | 21 ZZ | 4 YY | 63 ZZ
  const fer op2 = fer_small_norm(k.row == 1 ? YY : ZZ, m2, k);      // -     | 4 YY  | -    | 21 ZZ
  const fer t0m = fer_add(YY, fer_negate(op1, 1, k));               // rows 0, 3   [3]
  const fer y3 = fer_add(YY, op2);                                  // row 3       [2]
  const fer A2 = fer_sel(k, e, op1, op1, t0m) << (k.row == 2 ? 1u : 0u);
  const fer B2 = fer_sel(k, t0m, op2, e, y3) << (k.row < 2 ? 1u : 0u);
  const fer R = fer_mul(A2, B2, k);
  fer e2, o2, ya, yb;
  pt29r r;
  fer_pairs(R, e2, o2);
  fer_halves(e2, r.x, r.z);
  fer_halves(o2, ya, yb);
  r.y = fer_add(ya, yb);
  return r;
}

// Algorithm 7 (pt29_add): p + q, no exceptions.
S2K_DEV pt29r pt29r_add(const pt29r& p, const pt29r& q, const fer_consts& k) {
  const fer Y1 = fer_norm(p.y, k), Y2 = fer_norm(q.y, k);
  const fer T = fer_mul(fer_sel(k, p.x, Y1, p.z, p.z), fer_sel(k, q.x, Y2, q.z, q.z), k);        // t0 | t1 | t2 | -
  fer t0, t1, t2, tu;
  fer_bcast_rows(T, t0, t1, t2, tu);
  const fer e = fer_negate(fer_add(fer_sel(k, t0, t1, t0, t0), fer_sel(k, t1, t2, t2, t2)), 2, k);
  const fer A = fer_add(fer_sel(k, p.x, Y1, p.x, p.x), fer_sel(k, Y1, p.z, p.z, p.z));
  const fer B = fer_add(fer_sel(k, q.x, Y2, q.x, q.x), fer_sel(k, Y2, q.z, q.z, q.z));
  const fer U = fer_mul_plus(A, B, e, k);                                                         // t3 | t4 | y3' | -
  const fer Ts = fer_small_norm(T, k.row == 0 ? 3u : (k.row == 2 ? 21u : 1u), k);                 // 3 t0 | t1 | 21 t2
  const fer Us = fer_small_norm(U, k.row == 2 ? 21u : 1u, k);                                     // t3 | t4 | 21 y3'
  fer t0p, t1n, t2p, t3, t4, y3;
  fer_bcast_rows(Ts, t0p, t1n, t2p, tu);
  fer_bcast_rows(Us, t3, t4, y3, tu);
  const fer V = fer_add(t1n, fer_negate(t2p, 1, k));        // t1 - t2'   [3]
  const fer W = fer_add(t1n, t2p);                          // t1 + t2'   [2]
  // X3 = t3 V - t4 y3 | Z3 = t4 W + t3 t0' | Y3 = W V + y3 t0'
  const fer R = fer_mul_add_mul(fer_sel(k, t3, t4, W, W), fer_sel(k, V, W, V, V), fer_sel(k, fer_negate(t4, 1, k), t3, y3, y3),
                                fer_sel(k, y3, t0p, t0p, t0p), k);
  pt29r r;
  fer_bcast_rows(R, r.x, r.z, r.y, tu);
  return r;
}

}  // namespace s2k

// complete_path.h — device functions of the COMPLETE path: 8x32 field (fe.h), projective RCB
// formulas (point.h), the reference's algorithm shape.  Used by the fallback kernels of engine.hip
// (lanes the fast Jacobian kernel cannot decide, S2K_ECDSA_FORCE_COMPLETE) and by the Point /
// Scalar entry points of ops.hip.
#pragma once
#include "engine_internal.h"
#include "fe.h"
#include "point.h"
#include "sc.h"

// u*G for a plain scalar u (any 256-bit value): GT_WINDOWS table additions
S2K_DEV pt pt_base_mul(gt_view gt, const uint32_t u_in[8]) {
  uint32_t u[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) u[i] = u_in[i];
  pt acc = pt_from_affine(gt_load(gt, 0, gt_next_digit(u, gt.bits)));
#pragma unroll 1
  for (uint32_t w = 1; w < gt.windows; ++w) acc = pt_add_mixed(acc, gt_load(gt, w, gt_next_digit(u, gt.bits)));
  return acc;
}

// ---------------------------------------------------------------------------------------
// k*Q for a per-lane point on the COMPLETE path (8x32 field, RCB formulas; the fast path is
// k_verify_fast below): GLV split (point_mul_glv.go:203-254), then one fixed-window
// ladder over both 128-bit halves with signed odd digits:
//   k' = k | 1 = sum_{i=0..32} d_i 16^i,  d_i = 2*((k' >> (4i+1)) & 15) - 15 for i < 32, d_32 = 1
// every digit is odd and non-zero, so the per-lane table holds only the 8 odd multiples
// {1,3,..,15}*Q (the reference keeps 15 multiples and skips zero digits,
// point_mul_table.go:30-49) and every lane adds at every window.  The "+1" for even k is
// removed by one more complete addition of -+Q (or of the identity).
// beta*Q entries are derived on lookup by one multiplication (mulBeta, :191-200).
// Table storage: global scratch, word-major [entry*24 + coord*8 + limb][lane] so that a
// wave's access to one word of one entry is one contiguous 256-byte segment.
// ---------------------------------------------------------------------------------------
constexpr int QT_ENTRIES = 8;
constexpr int QT_WORDS = QT_ENTRIES * 24;

S2K_DEV void qt_store(uint32_t* __restrict__ qt, size_t stride, size_t lane, int entry, const pt& p) {
  uint32_t* base = qt + ((size_t)entry * 24) * stride + lane;
#pragma unroll
  for (int l = 0; l < 8; ++l) base[(size_t)l * stride] = p.x.v[l];
#pragma unroll
  for (int l = 0; l < 8; ++l) base[(size_t)(8 + l) * stride] = p.y.v[l];
#pragma unroll
  for (int l = 0; l < 8; ++l) base[(size_t)(16 + l) * stride] = p.z.v[l];
}
S2K_DEV pt qt_load(const uint32_t* __restrict__ qt, size_t stride, size_t lane, uint32_t entry) {
  const uint32_t* base = qt + ((size_t)entry * 24) * stride + lane;
  pt p;
#pragma unroll
  for (int l = 0; l < 8; ++l) p.x.v[l] = base[(size_t)l * stride];
#pragma unroll
  for (int l = 0; l < 8; ++l) p.y.v[l] = base[(size_t)(8 + l) * stride];
#pragma unroll
  for (int l = 0; l < 8; ++l) p.z.v[l] = base[(size_t)(16 + l) * stride];
  return p;
}

// top window first: returns (k' >> 125) & 15 ... by keeping k' left-aligned in 5 limbs
struct digit_stream {
  uint32_t w[5];   // k' << 31, so that bits (4i+1 .. 4i+4) of window 31 are the top nibble
};
S2K_DEV digit_stream ds_init(const sc& k_odd) {
  // k' < 2^129 occupies limbs 0..4 (limb 4 <= 1).  Window 31 is bits 125..128.
  // Left-align: shift so that bit 128 becomes bit 159 (top of limb 4): shift left by 31.
  digit_stream d;
  d.w[4] = (k_odd.v[4] << 31) | (k_odd.v[3] >> 1);
  d.w[3] = (k_odd.v[3] << 31) | (k_odd.v[2] >> 1);
  d.w[2] = (k_odd.v[2] << 31) | (k_odd.v[1] >> 1);
  d.w[1] = (k_odd.v[1] << 31) | (k_odd.v[0] >> 1);
  d.w[0] = (k_odd.v[0] << 31);
  return d;
}
S2K_DEV uint32_t ds_next(digit_stream& d) {
  uint32_t nib = d.w[4] >> 28;
  d.w[4] = (d.w[4] << 4) | (d.w[3] >> 28);
  d.w[3] = (d.w[3] << 4) | (d.w[2] >> 28);
  d.w[2] = (d.w[2] << 4) | (d.w[1] >> 28);
  d.w[1] = (d.w[1] << 4) | (d.w[0] >> 28);
  d.w[0] <<= 4;
  return nib;
}

S2K_DEV pt pt_mul_glv(const sc& k, const apt& q, uint32_t* __restrict__ qt, size_t stride, size_t lane) {
  sc k1, k2;
  bool neg1, neg2;
  sc_split_glv(k, k1, neg1, k2, neg2);
  bool even1 = (k1.v[0] & 1u) == 0, even2 = (k2.v[0] & 1u) == 0;
  k1.v[0] |= 1u;
  k2.v[0] |= 1u;

  // table of odd multiples: T[j] = (2j+1) Q
  pt q1 = pt_from_affine(q);
  pt q2 = pt_double_complete(q1);
  {
    pt cur = q1;
    qt_store(qt, stride, lane, 0, cur);
#pragma unroll 1
    for (int j = 1; j < QT_ENTRIES; ++j) {
      cur = pt_add_complete(cur, q2);
      qt_store(qt, stride, lane, j, cur);
    }
  }
  fe beta = fe_from_limbs(FE_BETA);

  digit_stream d1 = ds_init(k1), d2 = ds_init(k2);
  // top digits are +1: acc = s1*Q + s2*beta*Q
  pt acc = pt_cond_neg(q1, neg1);
  {
    pt qb = q1;
    qb.x = fe_mul(qb.x, beta);
    acc = pt_add_complete(acc, pt_cond_neg(qb, neg2));
  }
#pragma unroll 1
  for (int i = 31; i >= 0; --i) {
#pragma unroll 1
    for (int j = 0; j < 4; ++j) acc = pt_double_complete(acc);
    uint32_t w1 = ds_next(d1), w2 = ds_next(d2);
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      uint32_t w = t ? w2 : w1;
      bool neg = (t ? neg2 : neg1) != (w < 8u);
      uint32_t entry = (w < 8u) ? (7u - w) : (w - 8u);
      pt a = qt_load(qt, stride, lane, entry);
      if (t) a.x = fe_mul(a.x, beta);
      acc = pt_add_complete(acc, pt_cond_neg(a, neg));
    }
  }
  // remove the +1 of even halves: acc -= s*Q  <=>  add (-s)*Q
#pragma unroll 1
  for (int t = 0; t < 2; ++t) {
    pt a = q1;
    if (t) a.x = fe_mul(a.x, beta);
    a = pt_cond_neg(a, !(t ? neg2 : neg1));
    bool even = t ? even2 : even1;
    acc = pt_add_complete(acc, pt_select(even, pt_identity(), a));
  }
  return acc;
}

// ---------------------------------------------------------------------------------------
// ECDSA verification kernel, "complete" variant (every step exception-free).
// secec/ecdsa.go:392-470 per lane.
// ---------------------------------------------------------------------------------------
__device__ static const uint32_t FE_P_MINUS_N[8] = {0x2fc9baeeu, 0x402da172u, 0x50b75fc4u, 0x45512319u,
                                                    0x00000001u, 0, 0, 0};

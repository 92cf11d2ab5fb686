// engine.hip — gfx950 kernels and the C-ABI of include/secp256k1_voi_amd.h.
//
// One lane per signature / per point; every lane of a wave runs the same instruction
// stream (complete formulas, fixed windows), so there is no divergence on secret- or
// data-dependent branches and no idle lanes.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <sys/random.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "engine_internal.h"
#include "fe.h"
#include "fe29_inv.h"
#include "jacobian29.h"
#include "xyzz29.h"
#include "lane_tables.h"
#include "pt29.h"
#include "fe29r.h"
#include "pt29q.h"
#include "complete_path.h"
#include "point.h"
#include "sc.h"
#include "sc26.h"
#include "sha256.h"

using namespace s2k;

// ---------------------------------------------------------------------------------------
// Generator tables, resident in HBM:  GT_WINDOWS = ceil(256 / GT_BITS) tables of 2^GT_BITS affine
// points (64 MiB at 16 bits, 3 GiB at 22, 11 GiB at 24, 40 GiB at 26: the default),
//   T_0[d] = d * G - sum_{i>=1} B_i,   T_i[d] = (d + 1) * B_i,   B_i = 2^(GT_BITS i) * G,
// so that  u*G = sum_i T_i[(u >> GT_BITS i) & mask]  with GT_WINDOWS mixed additions, no
// doublings, no zero-digit special case (no entry is the identity) and no final correction.
// The reference's scalarBaseMultVartime (point_mul_table.go:197-211) is the same idea with
// 8-bit windows, sized for a CPU cache (510 KiB); HBM capacity buys wider windows here.  At 16
// bits the even-numbered windows of the reference's table blob are entries of these tables
// (tests/golden/gentable.json).
// Entry layout: 16 x u32 = X limbs (little-endian words) then Y limbs, 64-byte aligned.
// ---------------------------------------------------------------------------------------
// bases[i] = B_i = 2^(bits i) G (affine, 16 words) for i < windows; bases[windows] = -sum_{i>=1} B_i
__global__ void k_gen_gtable_bases(uint32_t* __restrict__ bases, uint32_t bits, uint32_t windows) {
  apt g;
  g.x = fe_from_limbs(FE_GX);
  g.y = fe_from_limbs(FE_GY);
  pt cur = pt_from_affine(g), sum = pt_identity();
#pragma unroll 1
  for (uint32_t i = 0; i <= windows; ++i) {
    apt a;
    if (i < windows) {
      pt_to_affine(a, cur);
      if (i >= 1) sum = pt_add_complete(sum, cur);
#pragma unroll 1
      for (uint32_t t = 0; t < bits; ++t) cur = pt_double_complete(cur);
    } else {
      pt_to_affine(a, pt_cond_neg(sum, true));
    }
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      bases[i * 16 + w] = a.x.v[w];
      bases[i * 16 + 8 + w] = a.y.v[w];
    }
  }
}
// one lane per entry: m * B_w by a double-and-add of bits + 1 steps (m = digit, or digit + 1 for w >= 1)
// (block0: the first block of this launch - the background build launches a table window by window)
__global__ void __launch_bounds__(256) k_gen_gtable(uint32_t* __restrict__ gt, const uint32_t* __restrict__ bases, uint32_t bits, uint32_t windows,
                                                    uint32_t block0) {
  size_t id = ((size_t)block0 + blockIdx.x) * 256 + threadIdx.x;
  uint32_t window = (uint32_t)(id >> bits), digit = (uint32_t)id & ((1u << bits) - 1u);
  apt b;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    b.x.v[w] = bases[window * 16 + w];
    b.y.v[w] = bases[window * 16 + 8 + w];
  }
  uint32_t m = window ? digit + 1 : digit;      // <= 2^bits
  pt acc = pt_identity();
#pragma unroll 1
  for (int bit = (int)bits; bit >= 0; --bit) {
    acc = pt_double_complete(acc);
    pt sum = pt_add_mixed(acc, b);
    acc = pt_select((m >> bit) & 1u, acc, sum);
  }
  if (window == 0) {
    apt c;
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      c.x.v[w] = bases[windows * 16 + w];
      c.y.v[w] = bases[windows * 16 + 8 + w];
    }
    acc = pt_add_mixed(acc, c);
  }
  apt a;
  pt_to_affine(a, acc);
  uint4* o = reinterpret_cast<uint4*>(gt + (id << 4));
  o[0] = make_uint4(a.x.v[0], a.x.v[1], a.x.v[2], a.x.v[3]);
  o[1] = make_uint4(a.x.v[4], a.x.v[5], a.x.v[6], a.x.v[7]);
  o[2] = make_uint4(a.y.v[0], a.y.v[1], a.y.v[2], a.y.v[3]);
  o[3] = make_uint4(a.y.v[4], a.y.v[5], a.y.v[6], a.y.v[7]);
}

// one signature, complete formulas only (secec/ecdsa.go:392-470)
S2K_DEV uint8_t verify_complete(size_t idx, const uint8_t* __restrict__ pub, const uint8_t* __restrict__ dig,
                                const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig, uint32_t flags,
                                gt_view gt, uint32_t* __restrict__ qt, size_t stride) {
  sc r, s;
  uint32_t e_raw[8];
  apt q;
  load_be32(r.v, rsig + idx * 32);
  load_be32(s.v, ssig + idx * 32);
  load_be32(e_raw, dig + idx * 32);
  load_be32(q.x.v, pub + idx * 64);
  load_be32(q.y.v, pub + idx * 64 + 32);

  // ParseCompactSignature range checks (s11n.go:129-144) + verify step 1 (ecdsa.go:400)
  bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s);
  if (flags & S2K_ECDSA_REJECT_MALLEABLE) ok = ok && !sc_is_gt_half_n(s);   // ecdsa.go:212
  // NewPublicKey: canonical coordinates on the curve (point_s11n.go:187-201)
  ok = ok && fe_is_canonical_raw(q.x.v) && fe_is_canonical_raw(q.y.v) && apt_on_curve(q);

  sc e = sc_reduce_once(e_raw);                    // hashToScalar (ecdsa.go:477-486)
  sc s_inv_m = sc_mont_inv(sc_to_mont(s));         // s^-1 * R
  sc u1 = sc_montmul(e, s_inv_m);                  // e / s   (plain)
  sc u2 = sc_montmul(r, s_inv_m);                  // r / s

  pt rg = pt_base_mul(gt, u1.v);
  pt rq = pt_mul_glv(u2, q, qt, stride, idx);
  pt R = pt_add_complete(rg, rq);                  // point_mul_glv.go:316

  ok = ok && !pt_is_identity(R);                   // ecdsa.go:450
  // x(R) mod n == r  <=>  X == r*Z  or  (r + n < p and X == (r + n)*Z)   (ecdsa.go:459-465)
  fe rf;
#pragma unroll
  for (int i = 0; i < 8; ++i) rf.v[i] = r.v[i];
  bool match = fe_eq(R.x, fe_mul(rf, R.z));
  if (u256_lt(r.v, FE_P_MINUS_N)) {
    fe r2;
    u256_add(r2.v, r.v, SC_N);
    match = match || fe_eq(R.x, fe_mul(r2, R.z));
  }
  return (ok && match) ? 1 : 0;
}

// every lane through the complete path (S2K_ECDSA_FORCE_COMPLETE; also the round-1 baseline kernel)
__global__ void __launch_bounds__(256)
k_ecdsa_verify(uint32_t n, const uint8_t* __restrict__ pub, const uint8_t* __restrict__ dig,
               const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig, uint32_t flags,
               uint8_t* __restrict__ out, gt_view gt, uint32_t* __restrict__ qt, size_t stride) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  out[idx] = verify_complete(idx, pub, dig, rsig, ssig, flags, gt, qt, stride);
}

// ---------------------------------------------------------------------------------------
// The worklist verifier: one signature with COMPLETE formulas on the 9x29 field (pt29.h: the
// Renes-Costello-Batina addition, mixed addition and doubling, point_projective.go:24,123,208).
// Same verdict as verify_complete for every input (tests run both), about a third of its cost:
// lanes whose Jacobian ladder hit an exceptional case are re-done at close to the fast path's own
// speed, so a batch crafted to send EVERY lane here (u1*G + u2*Q = infinity; anyone with a key
// pair can mint those) costs a small multiple of a normal batch instead of serialising on the
// 8x32 reference-shaped path.
// Ladder: odd GLV halves (sc_split_glv_odd), signed odd digits, projective table of the 8 odd
// multiples {1,3,..,15}*Q in planes [entry*27 + limb][lane] of the lane's table region; the
// lambda-half multiplies X by beta at the lookup.
// ---------------------------------------------------------------------------------------
S2K_DEV void p29_store(uint32_t* __restrict__ qt, size_t stride, size_t lane, int entry, const pt29& p) {
  uint32_t* b = qt + (size_t)(entry * 27) * stride + lane;
#pragma unroll
  for (int w = 0; w < 9; ++w) {
    b[(size_t)w * stride] = p.x.n[w];
    b[(size_t)(9 + w) * stride] = p.y.n[w];
    b[(size_t)(18 + w) * stride] = p.z.n[w];
  }
}
S2K_DEV pt29 p29_load(const uint32_t* __restrict__ qt, size_t stride, size_t lane, uint32_t entry) {
  const uint32_t* b = qt + (size_t)(entry * 27) * stride + lane;
  pt29 p;
#pragma unroll
  for (int w = 0; w < 9; ++w) {
    p.x.n[w] = b[(size_t)w * stride];
    p.y.n[w] = b[(size_t)(9 + w) * stride];
    p.z.n[w] = b[(size_t)(18 + w) * stride];
  }
  return p;
}
static_assert(QT_ENTRIES * 27 <= 8 * 8 * 4, "projective 9x29 table must fit in the lane's table region");

// u1*G + u2*Q (Q affine, on the curve) with complete formulas: the projective result (X : Y : Z)
S2K_DEV pt29 dsm_complete29(const sc& u1, const sc& u2, const fe29& qx, const fe29& qy, gt_view gt,
                            uint32_t* __restrict__ qt, size_t stride, size_t idx) {
  sc k1, k2;
  bool neg1, neg2;
  sc_split_glv_odd(u2, k1, neg1, k2, neg2);

  // table of odd multiples (projective): T[j] = (2j + 1) Q
  pt29 q1;
  q1.x = qx;
  q1.y = qy;
  q1.z = fe29_one();
  {
    pt29 q2 = pt29_double(q1), cur = q1;
    p29_store(qt, stride, idx, 0, cur);
#pragma unroll 1
    for (int j = 1; j < QT_ENTRIES; ++j) {
      cur = pt29_add(cur, q2);
      p29_store(qt, stride, idx, j, cur);
    }
  }
  const fe29 beta = fe29_from_words(FE_BETA);
  digit_stream d1 = ds_init(k1), d2 = ds_init(k2);
  // top digits are +1: acc = s1 * Q + s2 * lambda Q
  pt29 acc = q1;
  acc.y = fe29_cond_negate1(qy, neg1);
  acc.y = fe29_normalize_weak(acc.y);
  {
    fe29 bx = fe29_mul(qx, beta);
    acc = pt29_add_mixed(acc, bx, fe29_normalize_weak(fe29_cond_negate1(qy, neg2)));
  }
#pragma unroll 1
  for (int i = 31; i >= 0; --i) {
#pragma unroll 1
    for (int j = 0; j < 4; ++j) acc = pt29_double(acc);
    uint32_t w1 = ds_next(d1), w2 = ds_next(d2);
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
      uint32_t w = t ? w2 : w1;
      bool neg = (t ? neg2 : neg1) != (w < 8u);
      uint32_t entry = (w < 8u) ? (7u - w) : (w - 8u);
      pt29 a = p29_load(qt, stride, idx, entry);
      if (t) a.x = fe29_mul(a.x, beta);
      a.y = fe29_normalize_weak(fe29_cond_negate1(a.y, neg));
      acc = pt29_add(acc, a);
    }
  }
  // generator part (complete mixed additions; no table entry is the identity)
  {
    uint32_t u[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) u[w] = u1.v[w];
#pragma unroll 1
    for (uint32_t w = 0; w < gt.windows; ++w) {
      apt g = gt_load(gt, w, gt_next_digit(u, gt.bits));
      acc = pt29_add_mixed(acc, fe29_from_words(g.x.v), fe29_from_words(g.y.v));
    }
  }
  return acc;
}

// A worklist entry of the ECDSA flows is the signature's index, plus (batches below 2^30) a tag in the top bits:
// WL_DOUBLE_GEN = the fast ladder found u2 Q == u1 G, both finite, in its final addition: R = 2 u1 G;
// WL_DOUBLE_LAST = the plain ladder (generator part inside the kernel) found its sum equal to the last generator-table
// entry it was about to add: R = 2 T_last[top digit of u1].
constexpr uint32_t WL_TAG_LIMIT = 1u << 30, WL_INDEX_MASK = WL_TAG_LIMIT - 1u, WL_DOUBLE_GEN = 1u << 30, WL_DOUBLE_LAST = 2u << 30;

// (`key`: the 64 bytes X || Y of the signature's public key)
S2K_DEV uint8_t verify_complete29(size_t idx, const uint8_t* __restrict__ key, const uint8_t* __restrict__ dig,
                                  const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig, uint32_t flags,
                                  gt_view gt, uint32_t* __restrict__ qt, size_t stride, size_t lane,
                                  uint32_t tag = 0) {
  // idx: the signature; lane: the table column this thread may use (the worklist kernel passes the worklist
  // POSITION, not the signature: the keyed ladder queues signatures in key order, and columns picked by
  // signature would scatter a wave's table accesses over 64 cache lines: 47 ms instead of 20 for a batch
  // that is queued whole)
  sc r, s;
  uint32_t e_raw[8];
  apt q;
  load_be32(r.v, rsig + idx * 32);
  load_be32(s.v, ssig + idx * 32);
  load_be32(e_raw, dig + idx * 32);
  load_be32(q.x.v, key);
  load_be32(q.y.v, key + 32);
  bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s);
  if (flags & S2K_ECDSA_REJECT_MALLEABLE) ok = ok && !sc_is_gt_half_n(s);
  ok = ok && fe_is_canonical_raw(q.x.v) && fe_is_canonical_raw(q.y.v);
  if (!ok) {   // keep the arithmetic well defined; the verdict is already "invalid"
    q.x = fe_from_limbs(FE_GX);
    q.y = fe_from_limbs(FE_GY);
    s = sc_zero();
    s.v[0] = 1;
  }
  fe29 qx = fe29_from_words(q.x.v), qy = fe29_from_words(q.y.v);
  {
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    if (!fe29_eq(fe29_sqr(qy), rhs)) {   // not on the curve (point_s11n.go:298-307)
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
  }
  sc e = sc_reduce_once(e_raw);
  sc26 s_inv_m = sc26_mont_inv(sc26_to_mont(sc26_from_sc(s)));   // s^-1 * R (safegcd, modinv30.h)
  sc u1 = sc26_to_sc(sc26_mm(sc26_from_sc(e), s_inv_m)), u2 = sc26_to_sc(sc26_mm(sc26_from_sc(r), s_inv_m));
  pt29 acc;
  if (tag == WL_DOUBLE_GEN) {
    // the fast ladder has established u2 Q == u1 G (engine.hip, k_verify_fast: verdict): R = 2 u1 G, from the resident
    // tables with the complete formulas - no ladder
    acc.x = fe29_zero();
    acc.y = fe29_one();
    acc.z = fe29_zero();
    uint32_t u[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) u[w] = u1.v[w];
#pragma unroll 1
    for (uint32_t w = 0; w < gt.windows; ++w) {
      apt g = gt_load(gt, w, gt_next_digit(u, gt.bits));
      acc = pt29_add_mixed(acc, fe29_from_words(g.x.v), fe29_from_words(g.y.v));
    }
    acc = pt29_double(acc);
  } else if (tag == WL_DOUBLE_LAST) {
    uint32_t u[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) u[w] = u1.v[w];
    uint32_t digit = 0;
#pragma unroll 1
    for (uint32_t w = 0; w < gt.windows; ++w) digit = gt_next_digit(u, gt.bits);
    apt g = gt_load(gt, gt.windows - 1, digit);
    acc.x = fe29_from_words(g.x.v);
    acc.y = fe29_from_words(g.y.v);
    acc.z = fe29_one();
    acc = pt29_double(acc);
  } else {
    acc = dsm_complete29(u1, u2, qx, qy, gt, qt, stride, lane);
  }
  ok = ok && !fe29_is_zero(acc.z);                   // ecdsa.go:450
  // x(R) mod n == r  <=>  X == r*Z  or  (r + n < p and X == (r + n)*Z)   (ecdsa.go:459-465)
  bool match = fe29_eq(acc.x, fe29_mul(fe29_from_words(r.v), acc.z));
  if (u256_lt(r.v, FE_P_MINUS_N)) {
    uint32_t r2[8];
    u256_add(r2, r.v, SC_N);
    match = match || fe29_eq(acc.x, fe29_mul(fe29_from_words(r2), acc.z));
  }
  return (ok && match) ? 1 : 0;
}

// the lanes the fast kernel could not decide (final Z = 0): a short worklist, normally empty
__global__ void __launch_bounds__(256)
k_verify_fallback(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, const uint8_t* __restrict__ pub,
                  const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig,
                  uint32_t flags, uint8_t* __restrict__ out, gt_view gt, uint32_t* __restrict__ qt,
                  size_t stride) {
  uint32_t count = *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    const uint32_t e = wl[w];
    const size_t idx = e & WL_INDEX_MASK;
    out[idx] = verify_complete29(idx, pub + idx * 64, dig, rsig, ssig, flags, gt, qt, stride, w, e & ~WL_INDEX_MASK);
  }
}
// the same for a batch verified against a key set: the key of signature idx is keys[kidx[idx]]
__global__ void __launch_bounds__(256)
k_verify_fallback_keyset(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, const uint8_t* __restrict__ keys,
                         const uint32_t* __restrict__ kidx, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig,
                         const uint8_t* __restrict__ ssig, uint32_t flags, uint8_t* __restrict__ out, gt_view gt,
                         uint32_t* __restrict__ qt, size_t stride) {
  uint32_t count = *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    const uint32_t e = wl[w];
    const size_t idx = e & WL_INDEX_MASK;
    out[idx] = verify_complete29(idx, keys + (size_t)kidx[idx] * 64, dig, rsig, ssig, flags, gt, qt, stride, w, e & ~WL_INDEX_MASK);
  }
}

// ---------------------------------------------------------------------------------------
// Fast path, kernel 1: scalar preparation with batched inversion.
// Each thread owns PREP_M signatures (i = t, t+T, t+2T, ...): one inversion mod n (safegcd
// division steps, modinv30.h; the reference's Scalar.Invert is a Fermat chain,
// scalar_invert.go:11) is shared by PREP_M signatures through Montgomery's trick, 3 extra
// multiplications each, instead of one inversion per signature as in verify (ecdsa.go:428).
// PREP_M trades inversions against waves: measured at 2^20 (tools/ab_prep.sh) 2: 413, 3: 340,
// 4: 285, 6: 276, 8: 299, 16: 356, 32: 450 us (with the Fermat chain: 16: 466 us).
// Out per signature (word-major, coalesced):
//   u1 (8 words), |k1|, |k2| (4 words each; u2 = +-k1 +- k2*lambda), flag word.
// ---------------------------------------------------------------------------------------
#ifndef S2K_PREP_M
#define S2K_PREP_M 6
#endif
constexpr int PREP_M = S2K_PREP_M;
constexpr int PREP_WORDS = 17;
enum { PF_OK = 1, PF_NEG1 = 2, PF_NEG2 = 4, PF_K1_B128 = 8, PF_K2_B128 = 16 };   // bit 128 of the odd half-scalars

S2K_DEV void ws_store_sc26(uint32_t* __restrict__ base, size_t stride, size_t i, const sc26& v) {
#pragma unroll
  for (int w = 0; w < 10; ++w) base[(size_t)w * stride + i] = v.n[w];
}
S2K_DEV sc26 ws_load_sc26(const uint32_t* __restrict__ base, size_t stride, size_t i) {
  sc26 r;
#pragma unroll
  for (int w = 0; w < 10; ++w) r.n[w] = base[(size_t)w * stride + i];
  return r;
}

// recid != nullptr selects public-key recovery (RecoverPublicKey, ecdsa.go:244-282): the shared
// inversion is of r instead of s, and u1 = -e/r, u2 = s/r.
__global__ void __launch_bounds__(64)
k_scalar_prep(uint32_t n, uint32_t T, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig,
              const uint8_t* __restrict__ ssig, const uint8_t* __restrict__ recid, uint32_t flags,
              uint32_t* __restrict__ prep, uint32_t* __restrict__ pref, uint32_t* __restrict__ smont, size_t stride) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  // products run on the lazy 10x26 Montgomery form (sc26.h); pref/smont are 10-word planes
  sc26 one_m = sc26_from_limbs(SC26_ONE_M);
  sc26 acc = one_m;
#pragma unroll 1
  for (int j = 0; j < PREP_M; ++j) {
    size_t i = (size_t)t + (size_t)j * T;
    if (i >= n) break;
    sc s;
    load_be32(s.v, (recid ? rsig : ssig) + i * 32);   // the value whose inverse is needed
    bool ok_s = sc_is_canonical_raw(s.v) && !sc_is_zero(s);
    if (!ok_s) {   // keep the shared product invertible; the signature is rejected below
      s = sc_zero();
      s.v[0] = 1;
    }
    sc26 sm = sc26_to_mont(sc26_from_sc(s));
    ws_store_sc26(smont, stride, i, sm);
    acc = sc26_mm(acc, sm);
    ws_store_sc26(pref, stride, i, acc);
  }
  sc26 inv = sc26_mont_inv(acc);
#pragma unroll 1
  for (int j = PREP_M - 1; j >= 0; --j) {
    size_t i = (size_t)t + (size_t)j * T;
    if (i >= n) continue;
    sc26 sm = ws_load_sc26(smont, stride, i);
    sc26 prev = j > 0 ? ws_load_sc26(pref, stride, i - T) : one_m;
    sc26 s_inv_m = sc26_mm(inv, prev);             // s_i^-1 * R
    inv = sc26_mm(inv, sm);
    sc r, s;
    uint32_t e_raw[8];
    load_be32(r.v, rsig + i * 32);
    load_be32(s.v, ssig + i * 32);
    load_be32(e_raw, dig + i * 32);
    bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s);
    if (flags & S2K_ECDSA_REJECT_MALLEABLE) ok = ok && !sc_is_gt_half_n(s);
    sc e = sc_reduce_once(e_raw);
    sc u1, u2;
    uint32_t rid = 0;
    if (recid) {
      rid = recid[i];
      // RecoverPoint (point_s11n.go:245-282): id in [0,3]; bit 1 means x = r + n, which must stay < p
      ok = ok && rid < 4 && (!(rid & 2u) || u256_lt(r.v, FE_P_MINUS_N));
      u1 = sc26_to_sc(sc26_mm(sc26_from_sc(sc_neg(e)), s_inv_m));   // -e / r
      u2 = sc26_to_sc(sc26_mm(sc26_from_sc(s), s_inv_m));           //  s / r
    } else {
      u1 = sc26_to_sc(sc26_mm(sc26_from_sc(e), s_inv_m));
      u2 = sc26_to_sc(sc26_mm(sc26_from_sc(r), s_inv_m));
    }
    sc k1, k2;
    bool neg1, neg2;
    sc_split_glv_odd(u2, k1, neg1, k2, neg2);          // both halves odd, < 2^129
    uint32_t f = (ok ? PF_OK : 0) | (neg1 ? PF_NEG1 : 0) | (neg2 ? PF_NEG2 : 0) |
                 (k1.v[4] ? PF_K1_B128 : 0) | (k2.v[4] ? PF_K2_B128 : 0) | ((rid & 3u) << 8);
#pragma unroll
    for (int w = 0; w < 8; ++w) prep[(size_t)w * stride + i] = u1.v[w];
#pragma unroll
    for (int w = 0; w < 4; ++w) prep[(size_t)(8 + w) * stride + i] = k1.v[w];
#pragma unroll
    for (int w = 0; w < 4; ++w) prep[(size_t)(12 + w) * stride + i] = k2.v[w];
    prep[(size_t)16 * stride + i] = f;
  }
}

// Lanes of k_scalar_prep for n signatures: PREP_M signatures share a lane's inversion where that saves instructions that
// matter (2^20: a third of the kernel), but a call that cannot fill the chip anyway is better off with the latency of fewer
// signatures per lane - up to 2^16 lanes (a wave on every SIMD) are used before signatures start to share one (94 us for six in
// a row, 55 for one).
static inline uint32_t prep_lanes(size_t n) {
  const size_t shared = (n + PREP_M - 1) / PREP_M, wide = n < ((size_t)1 << 16) ? n : ((size_t)1 << 16);
  return (uint32_t)(shared > wide ? shared : wide);
}
// The same for ONE signature, its own inversion, the result in 17 words (u1 | k1 | k2 | flags: the planes' layout): the
// preparation wave of the wave-per-signature kernels (k_verify_row, k_recover_row).
S2K_DEV void scalar_prep_one(size_t i, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig,
                             const uint8_t* __restrict__ recid, uint32_t flags, uint32_t* __restrict__ out17) {
  sc x;
  load_be32(x.v, (recid ? rsig : ssig) + i * 32);      // the value whose inverse is needed
  if (!(sc_is_canonical_raw(x.v) && !sc_is_zero(x))) {   // (rejected below)
    x = sc_zero();
    x.v[0] = 1;
  }
  const sc26 s_inv_m = sc26_mont_inv(sc26_to_mont(sc26_from_sc(x)));     // x^-1 * R
  sc r, s;
  uint32_t e_raw[8];
  load_be32(r.v, rsig + i * 32);
  load_be32(s.v, ssig + i * 32);
  load_be32(e_raw, dig + i * 32);
  bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s);
  if (flags & S2K_ECDSA_REJECT_MALLEABLE) ok = ok && !sc_is_gt_half_n(s);
  const sc e = sc_reduce_once(e_raw);
  sc u1, u2;
  uint32_t rid = 0;
  if (recid) {
    rid = recid[i];
    ok = ok && rid < 4 && (!(rid & 2u) || u256_lt(r.v, FE_P_MINUS_N));     // RecoverPoint (point_s11n.go:245-282)
    u1 = sc26_to_sc(sc26_mm(sc26_from_sc(sc_neg(e)), s_inv_m));   // -e / r
    u2 = sc26_to_sc(sc26_mm(sc26_from_sc(s), s_inv_m));           //  s / r
  } else {
    u1 = sc26_to_sc(sc26_mm(sc26_from_sc(e), s_inv_m));
    u2 = sc26_to_sc(sc26_mm(sc26_from_sc(r), s_inv_m));
  }
  sc k1, k2;
  bool neg1, neg2;
  sc_split_glv_odd(u2, k1, neg1, k2, neg2);
#pragma unroll
  for (int w = 0; w < 8; ++w) out17[w] = u1.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) out17[8 + w] = k1.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) out17[12 + w] = k2.v[w];
  out17[16] = (ok ? PF_OK : 0) | (neg1 ? PF_NEG1 : 0) | (neg2 ? PF_NEG2 : 0) | (k1.v[4] ? PF_K1_B128 : 0) | (k2.v[4] ? PF_K2_B128 : 0) |
              ((rid & 3u) << 8);
}

// ---------------------------------------------------------------------------------------
// Fast path, kernel 2: per-lane table, ladder, generator part, final comparison.
// Per-lane table {1,3,..,15}*Q brought to one common Z (no inversion): build A_j = A_{j-1} + 2Q
// on the curve isomorphic by C = Z(2Q), remember H_j (Z_j = Z_{j-1} H_j), then scale entry j
// by (H_{j+1}...H_7)^{2,3}.  The ladder then only ever adds affine points (8 M + 3 S); its
// result has the true Z = Z_ladder * Z_7 * C.
// Table storage: tb_* (lane_tables.h).  Z_7 * C and the results handed to k_affine_finish go to the lane's
// "fin" elements.
// ---------------------------------------------------------------------------------------

// Waves per SIMD the joint-table key-set ladder is built for.  Measured (tools/gpu_joint_waves.sh, same box, two runs each):
// 3 waves (150 VGPRs, no spills) 1.743 / 1.746 ms, 4 waves (128 VGPRs, 21 spilled) 1.733 / 1.743 ms - the kernel is at the
// clock the chip gives it (1.89-2.02 GHz under 8 GB of table fetches per launch) either way; 3 stays.
#ifndef S2K_JOINT_WAVES
#define S2K_JOINT_WAVES 3
#endif
enum { MODE_ECDSA = 0, MODE_SCHNORR = 1, MODE_RECOVER = 2, MODE_POINT = 3, MODE_ECDSA_KEYED = 4, MODE_ECDSA_LEFT = 5,
       MODE_SCHNORR_KEYED = 6, MODE_SCHNORR_LEFT = 7, MODE_ECDSA_KEYSET = 8, MODE_ECDSA_KEYSET_JOINT = 9,
       MODE_ECDSA_KEYSET_JOINT5 = 10, MODE_ECDSA_KEYSET_JOINT6 = 11,
       // BIP-340 over a key set's tables, same four layouts: the ECDSA mode + 4
       MODE_SCHNORR_KEYSET = 12, MODE_SCHNORR_KEYSET_JOINT = 13, MODE_SCHNORR_KEYSET_JOINT5 = 14, MODE_SCHNORR_KEYSET_JOINT6 = 15 };
constexpr uint8_t VERDICT_PENDING = 2;   // k_verify_fast -> k_affine_finish
constexpr uint32_t KVF_FORCE_WORKLIST = 0x80000000u;   // top bit of k_verify_fast's first argument (batches are below 2^31)

// Digits of an odd half scalar k < 2^129 in the order the per-key ladder consumes them.  As in
// ds_init, digit i is nib_i = bits 4i+1 .. 4i+4 of k (signed value 2 nib_i - 15); digit i = 4c + j
// belongs to chunk c = 2^(16c) Q and round j, and the ladder runs round 3 first, chunks upwards: its
// nibble goes to place (3 - j) * 8 + c from the top of a 128-bit stream.
struct digit_stream4 {
  uint32_t w[4];
};
S2K_DEV digit_stream4 ds4_init_chunked(const sc& k_odd) {
  digit_stream4 d;
  d.w[0] = d.w[1] = d.w[2] = d.w[3] = 0;
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int bit = 4 * i + 1, limb = bit >> 5, sh = bit & 31;
    uint32_t nib = k_odd.v[limb] >> sh;
    if (sh > 28) nib |= k_odd.v[limb + 1] << (32 - sh);
    nib &= 15u;
    const int c = i >> 2, j = i & 3, pos = 124 - 4 * ((3 - j) * 8 + c);
    d.w[pos >> 5] |= nib << (pos & 31);
  }
  return d;
}
S2K_DEV uint32_t ds4_next(digit_stream4& d) {
  uint32_t nib = d.w[3] >> 28;
  d.w[3] = (d.w[3] << 4) | (d.w[2] >> 28);
  d.w[2] = (d.w[2] << 4) | (d.w[1] >> 28);
  d.w[1] = (d.w[1] << 4) | (d.w[0] >> 28);
  d.w[0] <<= 4;
  return nib;
}

// MODE_ECDSA:   pub = n x 64 (X||Y), rsig = n x 32 (r);      accept iff x(R) mod n == r
// MODE_SCHNORR: pub = n x 32 (x-only key, BIP-340), rsig = n x 64 signatures (r at offset 0);
//               P = lift_x(pub) (NewSchnorrPublicKey, schnorr.go:257-275), accept iff R != inf,
//               y(R) even and x(R) == r (verifySchnorrSignatureR, schnorr.go:451-478)
// MODE_RECOVER: pub unused, rsig = n x 32 (r); the point is R = RecoverPoint(r, id) with the id
//               from the prep flag word; out = ok bytes, out_pts = n x 65 records of
//               Q = (-e/r) G + (s/r) R (RecoverPublicKey, ecdsa.go:244-282)
// MODE_POINT:   pub = n x 64 (X||Y), rsig unused; out_pts = n x 65 records of u1*G + u2*P with
//               u1 and the split u2 taken from the prep planes as they are
//               (DoubleScalarMultBasepointVartime, point_mul_glv.go:307; k_hot_prep)
// MODE_ECDSA_KEYED / MODE_ECDSA_LEFT: MODE_ECDSA after the batch has been grouped by public key
//               (keyed.hip).  Lane idx < *kg.counters[...] handles signature perm[idx]: inputs, prep
//               planes and the verdict are the signature's, workspace columns are the lane's.
//               KEYED lanes take their points from the key's precomputed affine table (table ptab[idx],
//               no per-lane table, 12 doublings); LEFT lanes are the general path for the rest.
// MODE_SCHNORR_KEYED / MODE_SCHNORR_LEFT: the same for BIP-340 (tables of the lifted x-only keys); results go
//               to k_affine_finish in the SIGNATURE's fin column.
// Waves per SIMD the register allocator must leave room for.  Measured (2^20 signatures):
// unbounded (240 VGPRs, 2 waves) 11.08 ms; 3 waves (168 VGPRs, no spills) 10.89 ms; 4 waves
// (128 VGPRs, 96 spilled) 11.45 ms.
#ifndef S2K_FAST_WAVES
#define S2K_FAST_WAVES 3
#endif
// ---------------------------------------------------------------------------------------
// Grouped flow only: the generator part u1*G of every signature, on its own.  It needs nothing but
// the prepared u1, so it runs on the second stream, after the scalar preparation and beside the
// grouping and per-key table kernels (which are latency or memory bound and leave the multipliers
// idle); the ladders then finish with one Jacobian + Jacobian addition instead of GT_WINDOWS mixed ones.
// Out: (X, Y, Z) in three "fin"-format elements per SIGNATURE (gp planes).  A degenerate addition
// on the way (or u1 = 0) leaves Z = 0, which the ladder's final addition passes on to the worklist.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_generator_part(uint32_t first, uint32_t n, const uint32_t* __restrict__ prep, gt_view gt,
                 uint32_t* __restrict__ gp, size_t stride) {
  size_t idx = (size_t)first + (size_t)blockIdx.x * 256 + threadIdx.x;   // signatures [first, n)
  if (idx >= n) return;
  uint32_t u[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) u[w] = prep[(size_t)w * stride + idx];
  apt g = gt_load(gt, 0, gt_next_digit(u, gt.bits));
  xyzz29 xa = xyzz29_from_affine(fe29_from_words(g.x.v), fe29_from_words(g.y.v));   // XYZZ additions, as in the keyed ladder
  g = gt_load(gt, 1, gt_next_digit(u, gt.bits));
#pragma unroll 1
  for (uint32_t w = 1; w < gt.windows; ++w) {
    const fe29 gx = fe29_from_words(g.x.v), gy = fe29_from_words(g.y.v);
    if (w + 1 < gt.windows) g = gt_load(gt, w + 1, gt_next_digit(u, gt.bits));   // in flight during this addition
    xa = xyzz29_add_affine(xa, gx, gy);
  }
  // (two entries in flight instead of one halve the kernel's waiting - SQ_WAIT_ANY 87 M -> 46 M wave cycles - and change
  // neither its duration nor the step's: three waves per SIMD cover it; profiles/r06_wait_shares.txt)
  const jpt29 acc = xyzz29_to_jacobian(xa);
  fq_store(gp, stride, idx, 0, acc.x);
  fq_store(gp, stride, idx, 1, acc.y);
  fq_store(gp, stride, idx, 2, acc.z);
}

template <int MODE>
#ifdef S2K_FAST_MAX_WAVES   // experiment: cap the occupancy (leaves VGPRs for a kernel of another stream to run beside the ladder)
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(S2K_FAST_MAX_WAVES, S2K_FAST_MAX_WAVES)))
#else
// (The ladders over per-key tables hold no per-lane table state and come out at 126 VGPRs: four waves per SIMD.  That is
// the allocator's doing under a bound of three, and it is fragile: a rare branch that kept the accumulator alive across the
// final addition took 152, and asking for four waves outright spills - 26 VGPRs for <ECDSA_KEYED>, 28 for the key-set
// ladder, which stays at 142 VGPRs and three waves.  tests/test_counts_cpu.py watches the instruction counts; the
// resource usage is printed by tools/kernel_regs.sh.)
__global__ void __launch_bounds__(256, (MODE == MODE_ECDSA_KEYSET_JOINT || MODE == MODE_ECDSA_KEYSET_JOINT5 || MODE == MODE_ECDSA_KEYSET_JOINT6 ||
                                        MODE == MODE_SCHNORR_KEYSET_JOINT || MODE == MODE_SCHNORR_KEYSET_JOINT5 || MODE == MODE_SCHNORR_KEYSET_JOINT6)
                                           ? S2K_JOINT_WAVES : S2K_FAST_WAVES)
#endif
k_verify_fast(uint32_t n_and_flags, const uint8_t* __restrict__ pub, const uint8_t* __restrict__ rsig,
              const uint32_t* __restrict__ prep, uint32_t* __restrict__ qt, uint32_t* __restrict__ fin,
              gt_view gt, uint8_t* __restrict__ out, uint32_t* __restrict__ wl_count,
              uint32_t* __restrict__ wl, size_t stride, uint8_t* __restrict__ out_pts, uint64_t* __restrict__ clk,
              key_groups kg) {
  constexpr bool SKS = MODE >= MODE_SCHNORR_KEYSET && MODE <= MODE_SCHNORR_KEYSET_JOINT6;   // BIP-340 over a key set
  constexpr int KM = SKS ? MODE - 4 : MODE;                  // the key-set layout, as its ECDSA mode
  constexpr bool JOINT = KM == MODE_ECDSA_KEYSET_JOINT;      // ... over its joint tables: one addition per digit position
  constexpr int JW = KM == MODE_ECDSA_KEYSET_JOINT5 ? 5 : KM == MODE_ECDSA_KEYSET_JOINT6 ? 6 : 4;
  constexpr bool JOINTW = JW > 4;                            // ... over joint tables of 5- or 6-bit digits (kjw_geom: 26 / 22 positions)
  constexpr bool KEYSET = KM == MODE_ECDSA_KEYSET || JOINT || JOINTW;   // KEYED over a key set's 32-chunk tables: no doublings at all
  constexpr bool KEYED = MODE == MODE_ECDSA_KEYED || MODE == MODE_SCHNORR_KEYED || KEYSET;
  constexpr bool GROUPED = KEYED || MODE == MODE_ECDSA_LEFT || MODE == MODE_SCHNORR_LEFT;
  constexpr bool ECDSA = MODE == MODE_ECDSA || MODE == MODE_ECDSA_KEYED || MODE == MODE_ECDSA_LEFT || (KEYSET && !SKS);
  const bool force_wl = (n_and_flags & KVF_FORCE_WORKLIST) != 0;
  const uint32_t n = n_and_flags & ~KVF_FORCE_WORKLIST;
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;         // lane: workspace column
  size_t sig = idx;                                            // signature: input / prep / verdict column
  uint32_t lanes = n;
  if constexpr (GROUPED) {
    uint32_t lo = 0, hi = kg.counters[KEYED ? KG_NKEYED : KG_NLEFT];
    if (KEYED && kg.nparts > 1) {                              // two-part flow: this launch takes one side of the split
      const uint32_t sp = kg.counters[KG_SPLIT_LANE];
      if (kg.part) lo = sp;
      else hi = sp;
    }
    idx += lo;
    if (idx >= hi) return;
    lanes = hi - lo;
    sig = (KEYED ? kg.perm : kg.left)[idx];
  } else {
    if (idx >= n) return;
  }
  // effective shader clock of this launch (bench.py roofline): wave 0 of workgroup 0 stamps the
  // shader cycle counter and the constant-rate wall clock when it starts and when it ends
  // (and wave 0 of the LAST workgroup, which runs in the launch's final round: the clock sags under power
  // over the 8 ms of a launch)
  const bool stamp = clk != nullptr && (blockIdx.x == 0 || blockIdx.x == (lanes - 1) / 256) && threadIdx.x < 64;
  uint64_t t0c = 0, t0w = 0;
  if (stamp) {
    t0c = __builtin_readcyclecounter();
    t0w = __builtin_amdgcn_s_memrealtime();
  }
  uint32_t pf = prep[(size_t)16 * stride + sig];
  bool ok;
  fe29 qx, qy;
  if constexpr (KEYED) {
    ok = (pf & PF_OK) && kg.tinfo[kg.ptab[idx]];      // the key was validated when its table was built
  } else if constexpr (MODE == MODE_RECOVER) {
    uint32_t xw[8];
    load_be32(xw, rsig + idx * 32);
    ok = (pf & PF_OK) != 0;
    if (ok && (pf & 0x200u)) u256_add(xw, xw, SC_N);   // x = r + n (< p, checked by the prep kernel)
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xw[i] = FE_GX[i];
    }
    qx = fe29_from_words(xw);
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    bool has = fe29_sqrt(qy, rhs);                      // SetCompressedBytes (point_s11n.go:152-169)
    if (!has) {
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
    qy = fe29_normalize(qy);
    bool want_odd = (pf & 0x100u) != 0;
    qy = fe29_select(((qy.n[0] & 1u) != 0) != want_odd, qy, fe29_normalize_weak(fe29_negate(qy, 1)));
  } else if constexpr (ECDSA || MODE == MODE_POINT) {
    apt q;
    load_be32(q.x.v, pub + sig * 64);
    load_be32(q.y.v, pub + sig * 64 + 32);
    ok = (pf & PF_OK) && fe_is_canonical_raw(q.x.v) && fe_is_canonical_raw(q.y.v);
    if (!ok) {   // keep the arithmetic on the curve; the verdict is already "invalid"
      q.x = fe_from_limbs(FE_GX);
      q.y = fe_from_limbs(FE_GY);
    }
    qx = fe29_from_words(q.x.v);
    qy = fe29_from_words(q.y.v);
    // y^2 == x^3 + 7 (point_s11n.go:298-307)
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    bool on = fe29_eq(fe29_sqr(qy), rhs);
    if (!on) {
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
  } else {
    uint32_t xw[8];
    load_be32(xw, pub + sig * 32);
    ok = (pf & PF_OK) && fe_is_canonical_raw(xw);
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xw[i] = FE_GX[i];
    }
    qx = fe29_from_words(xw);
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;                                    // [1] (+7 on limb 0)
    bool has = fe29_sqrt(qy, rhs);
    if (!has) {   // not an x-coordinate of the curve
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
    qy = fe29_normalize(qy);
    qy = fe29_select((qy.n[0] & 1u) != 0, qy, fe29_normalize_weak(fe29_negate(qy, 1)));   // even y
  }
  bool neg1 = pf & PF_NEG1, neg2 = pf & PF_NEG2;
  if constexpr (SKS) {
    // BIP-340 multiplies P = lift_x(x), the point with EVEN y; the set holds the key as its caller gave it (X || Y): an odd Y
    // means P = -Q, i.e. both half scalars change sign (`pub`: the set's key array)
    const bool y_odd = (pub[(size_t)kg.ptab[idx] * 64 + 63] & 1u) != 0;
    neg1 = neg1 != y_odd;
    neg2 = neg2 != y_odd;
  }

  // ---- table ----
  if constexpr (!KEYED) {
    jpt29 a0;
    a0.x = qx;
    a0.y = qy;
    a0.z = fe29_one();
    jpt29 d = jpt29_double(a0);                      // x [1] y [2] z [1]
    fe29 c2 = fe29_sqr(d.z);
    fe29 c3 = fe29_mul(c2, d.z);
    const fe29 dx = d.x, dy = d.y;                   // the affine addend on the isomorphic curve
    jpt29 cur;
    cur.x = fe29_mul(qx, c2);
    cur.y = fe29_mul(qy, c3);
    cur.z = fe29_one();
    tb_store(qt, stride, idx, 0, TB_X, cur.x);
    tb_store(qt, stride, idx, 0, TB_Y, cur.y);
#pragma unroll 1
    for (int j = 1; j < QT_ENTRIES; ++j) {
      fe29 h;
      cur = jpt29_add_affine(cur, dx, dy, &h);
      tb_store(qt, stride, idx, j, TB_X, cur.x);
      tb_store(qt, stride, idx, j, TB_Y, cur.y);
      tb_store(qt, stride, idx, j, TB_BX, h);          // H_j, replaced by beta * x_j below
    }
    fq_store(fin, stride, idx, TB_ZC_ELEM, fe29_mul(cur.z, d.z));   // Z_7 * C
    fe29 prev_x = cur.x;
    const fe29 beta = fe29_from_words(FE_BETA);
    fe29 rr = fe29_one();
#pragma unroll 1
    for (int j = QT_ENTRIES - 2; j >= 0; --j) {
      rr = fe29_mul(rr, tb_load(qt, stride, idx, j + 1, TB_BX));
      tb_store(qt, stride, idx, j + 1, TB_BX, fe29_mul(prev_x, beta));
      fe29 r2 = fe29_sqr(rr);
      fe29 r3 = fe29_mul(r2, rr);
      prev_x = fe29_mul(tb_load(qt, stride, idx, j, TB_X), r2);
      fe29 y = fe29_mul(tb_load(qt, stride, idx, j, TB_Y), r3);
      tb_store(qt, stride, idx, j, TB_X, prev_x);
      tb_store(qt, stride, idx, j, TB_Y, y);
    }
    tb_store(qt, stride, idx, 0, TB_BX, fe29_mul(prev_x, beta));
  }

  // ---- ladder over |k1|, |k2| ----
  sc k1 = sc_zero(), k2 = sc_zero();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    k1.v[w] = prep[(size_t)(8 + w) * stride + sig];
    k2.v[w] = prep[(size_t)(12 + w) * stride + sig];
  }
  k1.v[4] = (pf & PF_K1_B128) ? 1u : 0u;   // odd by construction (sc_split_glv_odd): the signed
  k2.v[4] = (pf & PF_K2_B128) ? 1u : 0u;   // odd-digit recoding is exact, no final correction
  jpt29 acc;
  if constexpr (KEYED) {
    // k = 16^32 + sum_i d_i 16^i with d_i = 2 nib_i - 15, i = 4c + j: round j (3 down to 0) adds
    // d_(4c+j) * 2^(16c) Q for the eight chunks c, with four doublings between rounds; the leading
    // 16^32 Q = 16^3 * 2^116 Q goes in first (for both halves at once: one table point, see below).
    using G = kt_geom<KEYSET ? KS_CHUNKS : KT_CHUNKS>;
    const uint4* kt = kg.ktab + (size_t)kg.ptab[idx] * (G::SLOTS * 8);
    const fe29 kw = ke_load(kt + (size_t)(G::SCR + G::W_SLOT / 3) * 8, G::W_SLOT % 3);   // wanted at the very end: asked for early
    // The 64 additions of this ladder run in XYZZ coordinates (xyzz29.h: 8 M + 2 S in 9 reductions against 8 M + 3 S in
    // 10 for the Jacobian mixed addition: 1 441 instead of 1 578 instructions); the 12 doublings stay Jacobian (7 products
    // against 9), so the accumulator changes form at the three round borders (3 M + 1 S each way together).  The
    // exceptional-case rule carries over: equal x makes ZZ = 0, ZZ = 0 is sticky through additions, conversions and
    // doublings, and ends as Z = 0 -> worklist.
    xyzz29 xa;
    {
      // +-L +- phi(L): the table holds L + phi(L) and L - phi(L) (keyed.hip), the other two are their negatives
      // (wide joint tables: L = 2^(W POS) Q, the pair behind the positions of the key's joint table)
      fe29 lx, ly;
      if constexpr (JOINTW)
        jw_load(kg.jtab + (size_t)kg.ptab[idx] * kjw_geom<JW>::KEY_QUADS + (kjw_geom<JW>::LEAD + (neg1 == neg2 ? 0 : 1)) * kjw_geom<JW>::EQ, lx, ly);
      else
        ke_load_xy(kt + (size_t)(neg1 == neg2 ? G::LEAD : G::LEAD + 1) * 8, false, lx, ly);
      xa = xyzz29_from_affine(lx, fe29_cond_negate1(ly, neg1));
    }
    if constexpr (KEYSET) {
      // Key sets: one chunk per digit, {1,3,..,15} * 16^i Q for i < 32 and L = 16^32 Q: k = 16^32 + sum d_i 16^i is 64
      // table additions, in any order - bottom up, nibble i = bits 4i+1 .. 4i+4 of the odd half scalar (k >> 1, 128 bits).
      // JOINT tables hold, per position, E_a + s phi(E_b) for all pairs of odd multiples and both signs: the two halves'
      // digits d1 = +-(2a+1), d2 = +-(2b+1) are ONE addition of +-(E_a + s phi(E_b)), s = the product of their signs.
      uint32_t a[4], b[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        a[w] = (k1.v[w] >> 1) | (k1.v[w + 1] << 31);
        b[w] = (k2.v[w] >> 1) | (k2.v[w + 1] << 31);
      }
      [[maybe_unused]] const uint4* jt = JOINT ? kg.jtab + (size_t)kg.ptab[idx] * KJ_KEY_QUADS
                                        : JOINTW ? kg.jtab + (size_t)kg.ptab[idx] * kjw_geom<JW>::KEY_QUADS : nullptr;
      if constexpr (JOINTW) {
        // W-bit digits: w_i = bits W i .. W i + W - 1 of (k - 1) / 2, d_i = 2 w_i - (2^W - 1): below 2^(W-1) negative with
        // magnitude 2 (NE - 1 - w) + 1, else positive with magnitude 2 (w - NE) + 1
        using J = kjw_geom<JW>;
        constexpr uint32_t WM = (1u << JW) - 1u, NE = (uint32_t)J::NE;
        // The entry of position c + 1 is asked for BEFORE the addition of position c (measured both ways, profiles/r06_keyset_wait.json): a joint
        // table is 0.8 / 2.75 MiB per key and a lookup is a 64-byte read from HBM more often than not - issued behind the
        // addition, its latency stood between every two additions of a wave.
        auto next_entry = [&](int c, bool& n1_out) -> uint32_t {
          const uint32_t w1 = a[0] & WM, w2 = b[0] & WM;
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            a[w] = (a[w] >> JW) | (a[w + 1] << (32 - JW));
            b[w] = (b[w] >> JW) | (b[w + 1] << (32 - JW));
          }
          a[3] >>= JW;
          b[3] >>= JW;
          const bool n1 = neg1 != (w1 < NE), n2 = neg2 != (w2 < NE);
          const uint32_t ea = (w1 < NE) ? (NE - 1u - w1) : (w1 - NE), eb = (w2 < NE) ? (NE - 1u - w2) : (w2 - NE);
          n1_out = n1;
          return (((uint32_t)c * NE + ea) * NE + eb) * 2u + (n1 != n2 ? 1u : 0u);
        };
        bool n1_cur;
        jw_raw raw = jw_fetch(jt + (size_t)next_entry(0, n1_cur) * J::EQ);
#pragma unroll 1
        for (int c = 0; c < J::POS; ++c) {
          const jw_raw cur = raw;
          const bool n1 = n1_cur;
          if (c + 1 < J::POS) raw = jw_fetch(jt + (size_t)next_entry(c + 1, n1_cur) * J::EQ);
          fe29 x, y;
          jw_point(cur, x, y);
          xa = xyzz29_add_affine(xa, x, fe29_cond_negate1(y, n1));
        }
      } else if constexpr (JOINT) {
        // the 4-bit joint ladder with the same lead: the 80-byte entry of position c + 1 (18 limbs) asked for before the addition
        // of position c
        auto next_entry4 = [&](int c, bool& n1_out) -> uint32_t {
          const uint32_t w1 = a[0] & 15u, w2 = b[0] & 15u;
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            a[w] = (a[w] >> 4) | (a[w + 1] << 28);
            b[w] = (b[w] >> 4) | (b[w + 1] << 28);
          }
          a[3] >>= 4;
          b[3] >>= 4;
          const bool n1 = neg1 != (w1 < 8u), n2 = neg2 != (w2 < 8u);
          const uint32_t ea = (w1 < 8u) ? (7u - w1) : (w1 - 8u), eb = (w2 < 8u) ? (7u - w2) : (w2 - 8u);
          n1_out = n1;
          return (((uint32_t)c * 8u + ea) * 8u + eb) * 2u + (n1 != n2 ? 1u : 0u);
        };
        bool n1_nxt;
        fe29 nx, ny;
        je_load(jt + (size_t)next_entry4(0, n1_nxt) * KJ_ENTRY_QUADS, nx, ny);
#pragma unroll 1
        for (int c = 0; c < KS_CHUNKS; ++c) {
          const fe29 x = nx, y = ny;
          const bool n1 = n1_nxt;
          if (c + 1 < KS_CHUNKS) je_load(jt + (size_t)next_entry4(c + 1, n1_nxt) * KJ_ENTRY_QUADS, nx, ny);
          xa = xyzz29_add_affine(xa, x, fe29_cond_negate1(y, n1));
        }
      } else {
#pragma unroll 1
        for (int c = 0; c < KS_CHUNKS; ++c) {
          const uint32_t w1 = a[0] & 15u, w2 = b[0] & 15u;
#pragma unroll
          for (int w = 0; w < 3; ++w) {
            a[w] = (a[w] >> 4) | (a[w + 1] << 28);
            b[w] = (b[w] >> 4) | (b[w + 1] << 28);
          }
          a[3] >>= 4;
          b[3] >>= 4;
#pragma unroll 1
          for (int t = 0; t < 2; ++t) {
            uint32_t w = t ? w2 : w1;
            bool neg = (t ? neg2 : neg1) != (w < 8u);
            uint32_t entry = (w < 8u) ? (7u - w) : (w - 8u);
            fe29 x, y;
            ke_load_xy(kt + (size_t)(c * 8 + entry) * 8, t != 0, x, y);
            xa = xyzz29_add_affine(xa, x, fe29_cond_negate1(y, neg));
          }
        }
      }
    } else {
      digit_stream4 d1 = ds4_init_chunked(k1), d2 = ds4_init_chunked(k2);
#pragma unroll 1
      for (int round = 0; round < 4; ++round) {
        if (round) {
          jpt29 j4 = xyzz29_to_jacobian(xa);
#pragma unroll 1
          for (int j = 0; j < 4; ++j) j4 = jpt29_double(j4);
          xa = xyzz29_from_jacobian(j4);
        }
#pragma unroll 1
        for (int c = 0; c < KT_CHUNKS; ++c) {
          uint32_t w1 = ds4_next(d1), w2 = ds4_next(d2);
#pragma unroll 1
          for (int t = 0; t < 2; ++t) {
            uint32_t w = t ? w2 : w1;
            bool neg = (t ? neg2 : neg1) != (w < 8u);
            uint32_t entry = (w < 8u) ? (7u - w) : (w - 8u);
            fe29 x, y;
            ke_load_xy(kt + (size_t)(c * 8 + entry) * 8, t != 0, x, y);
            xa = xyzz29_add_affine(xa, x, fe29_cond_negate1(y, neg));
          }
        }
      }
      // (the next entry asked for before the current addition - what the key-set ladders do - costs this ladder its fourth wave
      // per SIMD: 145-147 VGPRs against 126.  Measured in both forms, four alternating pairs: the step 4.89 against 4.91 ms,
      // the kernel 58.2 M against 57.8 M cycles under the counters - nothing, and not kept; profiles/r06_wait_shares.txt)
    }
    acc = xyzz29_to_jacobian(xa);
    // the table's points are affine on the curve isomorphic by W (keyed.hip): back on secp256k1 itself
    acc.z = fe29_mul(acc.z, kw);
  } else {
    digit_stream d1 = ds_init(k1), d2 = ds_init(k2);
    {
      fe29 t0x = tb_load(qt, stride, idx, 0, TB_X), t0y = tb_load(qt, stride, idx, 0, TB_Y), t0bx = tb_load(qt, stride, idx, 0, TB_BX);
      acc.x = t0x;
      acc.y = fe29_cond_negate1(t0y, neg1);
      acc.z = fe29_one();
      acc = jpt29_add_affine(acc, t0bx, fe29_cond_negate1(t0y, neg2));
    }
#pragma unroll 1
    for (int i = 31; i >= 0; --i) {
#pragma unroll 1
      for (int j = 0; j < 4; ++j) acc = jpt29_double(acc);
      uint32_t w1 = ds_next(d1), w2 = ds_next(d2);
#pragma unroll 1
      for (int t = 0; t < 2; ++t) {
        uint32_t w = t ? w2 : w1;
        bool neg = (t ? neg2 : neg1) != (w < 8u);
        uint32_t entry = (w < 8u) ? (7u - w) : (w - 8u);
        fe29 x, y;
        tb_load_xy(qt, stride, idx, entry, t != 0, x, y);
        acc = jpt29_add_affine(acc, x, fe29_cond_negate1(y, neg));
      }
    }
    acc.z = fe29_mul(acc.z, fq_load(fin, stride, idx, TB_ZC_ELEM));   // times Z_7 * C: back on secp256k1 itself
  }

  // ---- generator part: u1*G from the resident tables ----
  // (final_only: Z was not 0 BEFORE the last addition and the addend is finite - then a zero after it arose in that addition,
  // and the verdict code below decides it instead of queueing the lane for the complete-formula kernel)
  bool final_only = false;
  if constexpr (GROUPED) {
    jpt29 q;                                            // computed by k_generator_part
    q.x = fq_load(kg.gp, stride, sig, 0);
    q.y = fq_load(kg.gp, stride, sig, 1);
    q.z = fq_load(kg.gp, stride, sig, 2);
    if constexpr (ECDSA) final_only = !fe29_is_zero(acc.z) && !fe29_is_zero(q.z);
    acc = jpt29_add(acc, q);
  } else {
    uint32_t u[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) u[w] = prep[(size_t)w * stride + sig];
#pragma unroll 1
    for (uint32_t w = 0; w + 1 < gt.windows; ++w) {
      apt g = gt_load(gt, w, gt_next_digit(u, gt.bits));
      acc = jpt29_add_affine(acc, fe29_from_words(g.x.v), fe29_from_words(g.y.v));
    }
    {   // the last addition apart: whether Z was 0 before it is what the verdict code needs to know
      apt g = gt_load(gt, gt.windows - 1, gt_next_digit(u, gt.bits));
      if constexpr (ECDSA) final_only = !fe29_is_zero(acc.z);
      acc = jpt29_add_affine(acc, fe29_from_words(g.x.v), fe29_from_words(g.y.v));
    }
  }

  // ---- verdict ----
  uint8_t verdict = 0;
  // (MODE_RECOVER / MODE_POINT: the 65-byte records start as zeros - one fill of the output array by the caller; the loop of
  // 65 single-byte stores per lane that used to stand here cost the recovery ladder a tenth of its time)
  // Z = 0: infinity, or an exceptional case of the incomplete formulas somewhere on the way.  Where the zero first appears
  // in the FINAL addition of two finite points (Z3 = Z1 Z2 H = 0: equal x) the answer is at hand.  With H = 0 the mixed
  // addition leaves X3 = I^2, I = Y1 - S2 (jacobian29.h), so X3 != 0 says OPPOSITE points: R is the identity and verify
  // rejects it (ecdsa.go:450).  That is what anyone who owns a key gets with r = -e/d in every lane, whatever the digests
  // (the signature need not be valid), and it used to send the whole batch through the complete-formula kernel: 3.4 x a
  // normal step; now the lane is done.  X3 = 0 says EQUAL points (r = e/d): R is twice the last addend - left to the
  // worklist kernel with a tag (WL_DOUBLE_GEN: 2 u1 G in the grouped flows, WL_DOUBLE_LAST: twice the last generator-table
  // entry in the plain one; a twelfth of a full verification or less), since the keyed ladder has no registers to spare.
  // Zeros from INSIDE the ladder or the generator part (and u1 = 0: a digest that is 0 mod n) go to the worklist as before.
  bool undecided = ok && fe29_is_zero(acc.z), is_identity = false;
  uint32_t wl_tag = 0;
  if constexpr (ECDSA) {
    if (force_wl && ok) {                                // S2K_ECDSA_FORCE_WORKLIST (diagnostic)
      undecided = true;
      final_only = false;
    }
  }
  if constexpr (ECDSA) {
    if (undecided && final_only) {
      if (!fe29_is_zero(acc.x)) {
        undecided = false;
        is_identity = true;                              // P - P
      } else if (n < WL_TAG_LIMIT) {
        wl_tag = GROUPED ? WL_DOUBLE_GEN : WL_DOUBLE_LAST;   // P + P: R = 2 u1 G, or twice the last table entry
      }
    }
  }
  if (ok) {
    if (undecided) {
      // infinity or an exceptional case inside the ladder: the complete kernel decides
      uint32_t pos = atomicAdd(wl_count, 1u);
      wl[pos] = (uint32_t)sig | wl_tag;
    } else if (is_identity) {
      verdict = 0;                                      // R = infinity (ecdsa.go:450)
    } else if constexpr (!ECDSA) {
      // affine epilogue (key bytes / even-y test) needs 1/Z: leave (X, Y, Z) in the SIGNATURE's fin
      // column (its own lane's when not grouped) and let k_affine_finish share one inversion between 16
      fq_store(fin, stride, sig, 0, acc.x);
      fq_store(fin, stride, sig, 1, acc.y);
      fq_store(fin, stride, sig, 2, acc.z);
      verdict = VERDICT_PENDING;
    } else {
      // x(R) mod n == r  (ecdsa.go:450-465)
      uint32_t rw[8];
      load_be32(rw, rsig + sig * 32);
      fe29 zz = fe29_sqr(acc.z);
      bool match = fe29_eq(acc.x, fe29_mul(fe29_from_words(rw), zz));
      if (u256_lt(rw, FE_P_MINUS_N)) {
        uint32_t r2[8];
        u256_add(r2, rw, SC_N);
        match = match || fe29_eq(acc.x, fe29_mul(fe29_from_words(r2), zz));
      }
      verdict = match ? 1 : 0;
    }
  }
  out[sig] = verdict;
  if (stamp && threadIdx.x == 0) {
    uint64_t* c = clk + (blockIdx.x == 0 ? 0 : 4);
    c[0] = t0c;
    c[1] = __builtin_readcyclecounter();
    c[2] = t0w;
    c[3] = __builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------------------
// Small batches: ONE WAVE PER SIGNATURE (round 5).  A call of up to a few thousand signatures - the reference's own
// shape is a loop of single Verify calls, BASELINE config 1 verifies 1024 - leaves the chip empty whatever the kernel: what
// it costs is the latency of one lane's ladder, 261 k dependent instructions, 0.7 ms for 64 signatures as for 8192
// (bench.py batch_sweep).  Here the ladder runs in the row arithmetic of fe29r.h (a field element = one register, one limb
// per lane of a 16-lane row, the four products of a formula layer in the four rows of the wave): the same signed-odd-digit
// GLV ladder as k_verify_fast - table {1,3,..,15} Q, 32 windows of four doublings and two additions, the generator part from
// the resident tables - on the COMPLETE projective formulas (pt29r_add / pt29r_double: no exceptional case, so no worklist
// and no second kernel), ~70 k instructions of one wave per signature.  Digits, table indices and generator-table addresses
// are wave-uniform.  Verdicts are those of the other paths (same preparation kernel, same accept rule ecdsa.go:392-470).
// ---------------------------------------------------------------------------------------
// limb j (lane j of every row) of a 256-bit value given as eight little-endian words every lane holds
S2K_DEV fer fer_from_words(const uint32_t w[8], const fer_consts& k) {
  const uint32_t bit = 29u * k.j, idx = bit >> 5, sh = bit & 31u;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    lo = idx == (uint32_t)i ? w[i] : lo;
    hi = idx + 1 == (uint32_t)i ? w[i] : hi;
  }
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return k.j <= 8 ? ((uint32_t)(v >> sh) & F29_M) : 0u;       // (limb 8: bits 232..255, 24 of them)
}
// value == 0 (mod p), the same answer in every lane
S2K_DEV bool fer_is_zero(fer a, const fer_consts& k) { return fe29_is_zero(fer_to_fe29(fer_norm(a, k))); }

// canonical value odd?  (the same answer in every lane of a row)
S2K_DEV bool fer_is_odd(fer a, const fer_consts& k) { return (fe29_normalize(fer_to_fe29(fer_norm(a, k))).n[0] & 1u) != 0; }
S2K_DEV fer fer_sqr_n(fer a, int n, const fer_consts& k) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = fer_mul(a, a, k);
  return a;
}
// a^((p+1)/4): the chain of fe29_sqrt (Sqrt, internal/field/field_sqrt_ratio.go:14), row by row - four roots per wave
S2K_DEV fer fer_sqrt_chain(fer a, const fer_consts& k) {
  const fer x2 = fer_mul(fer_mul(a, a, k), a, k);
  const fer x3 = fer_mul(fer_mul(x2, x2, k), a, k);
  const fer x6 = fer_mul(fer_sqr_n(x3, 3, k), x3, k);
  const fer x9 = fer_mul(fer_sqr_n(x6, 3, k), x3, k);
  const fer x11 = fer_mul(fer_sqr_n(x9, 2, k), x2, k);
  const fer x22 = fer_mul(fer_sqr_n(x11, 11, k), x11, k);
  const fer x44 = fer_mul(fer_sqr_n(x22, 22, k), x22, k);
  const fer x88 = fer_mul(fer_sqr_n(x44, 44, k), x44, k);
  const fer x176 = fer_mul(fer_sqr_n(x88, 88, k), x88, k);
  const fer x220 = fer_mul(fer_sqr_n(x176, 44, k), x44, k);
  const fer x223 = fer_mul(fer_sqr_n(x220, 3, k), x3, k);
  fer t = fer_mul(fer_sqr_n(x223, 23, k), x22, k);
  t = fer_mul(fer_sqr_n(t, 6, k), x2, k);
  return fer_sqr_n(t, 2, k);
}
// x^3 + 7
S2K_DEV fer fer_curve_rhs(fer x, const fer_consts& k) {
  const fer seven = k.j == 0 ? 7u : 0u;
  return fer_mul_plus(fer_mul(x, x, k), x, seven, k);
}

// The wave-per-signature kernels.  A block is four signature waves and one PREPARATION wave: lanes 0..3 of the
// fifth wave run the scalar arithmetic of the block's four signatures (s^-1, u1, u2, the odd GLV split: one lane's chain of
// 55 us) into LDS while the signature waves do what does not depend on it - key checks, square roots, the key's table -, then
// the block meets at one barrier.  One launch per call.
//
// row_table: the table of the key in LDS, one 256-byte line per coordinate of an entry (lane l of the wave at word l) - x, y,
// z and lambda's x (beta * x) of (2j + 1) Q, j < 8; an entry is three loads at a wave-uniform offset.
S2K_DEV void row_table(const pt29r& Q1, uint32_t (*tab)[8][64], const fer_consts& k, uint32_t lane) {
  const fer beta = fer_from_words(FE_BETA, k);
  const pt29r D = pt29r_double(Q1, k);
  pt29r cur = Q1;
#pragma unroll 1
  for (int j = 0; j < 8; ++j) {
    if (j) cur = pt29r_add(cur, D, k);
    tab[0][j][lane] = cur.x;
    tab[1][j][lane] = fer_norm(cur.y, k);
    tab[2][j][lane] = cur.z;
    tab[3][j][lane] = fer_mul(cur.x, beta, k);
  }
}
// u * G + k1 * Q + k2 * lambda(Q) for ONE signature on the whole wave, Q's table in `tab`, u | k1 | k2 | flags in the 17 words
// `p` (LDS): 32 x (4 doublings + 2 additions) on the complete formulas, then the generator part from the resident tables (no
// entry is the identity, no digit is special).  The result is projective.
S2K_DEV pt29r row_ladder(uint32_t (*tab)[8][64], const uint32_t* p, gt_view gt, const fer_consts& k, uint32_t lane) {
  const fer one = k.j == 0 ? 1u : 0u;
  const uint32_t pf = p[16];
  // |k1|, |k2| odd, < 2^129 (sc_split_glv_odd)
  sc k1 = sc_zero(), k2 = sc_zero();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    k1.v[w] = p[8 + w];
    k2.v[w] = p[12 + w];
  }
  k1.v[4] = (pf & PF_K1_B128) ? 1u : 0u;
  k2.v[4] = (pf & PF_K2_B128) ? 1u : 0u;
  const bool neg1 = pf & PF_NEG1, neg2 = pf & PF_NEG2;
  digit_stream d1 = ds_init(k1), d2 = ds_init(k2);
  auto entry_of = [&](uint32_t w, bool lam, bool sneg) -> pt29r {   // the entry of digit w (signed odd: 2w - 15), or its image
    const bool neg = sneg != (w < 8u);                              // under the endomorphism, negated with the scalar's sign
    const uint32_t e = (uint32_t)__builtin_amdgcn_readfirstlane((int)((w < 8u) ? (7u - w) : (w - 8u)));
    pt29r a;
    a.x = tab[lam ? 3 : 0][e][lane];
    a.y = tab[1][e][lane];
    a.z = tab[2][e][lane];
    if (neg) a.y = fer_negate(a.y, 1, k);                           // [2]: the addition normalises y
    return a;
  };
  pt29r acc = pt29r_add(entry_of(8u, false, neg1), entry_of(8u, true, neg2), k);   // the top digits are +1 (w = 8)
#pragma unroll 1
  for (int i = 31; i >= 0; --i) {
    const uint32_t w1 = ds_next(d1), w2 = ds_next(d2);
    const pt29r a1 = entry_of(w1, false, neg1), a2 = entry_of(w2, true, neg2);   // (in flight during the doublings)
#pragma unroll 1
    for (int j = 0; j < 4; ++j) acc = pt29r_double(acc, k);
    acc = pt29r_add(acc, a1, k);
    acc = pt29r_add(acc, a2, k);
  }
  uint32_t u[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) u[w] = p[w];
  apt g = gt_load(gt, 0, gt_next_digit(u, gt.bits));
#pragma unroll 1
  for (uint32_t w = 0; w < gt.windows; ++w) {
    pt29r q;
    q.x = fer_from_words(g.x.v, k);
    q.y = fer_from_words(g.y.v, k);
    q.z = one;
    if (w + 1 < gt.windows) g = gt_load(gt, w + 1, gt_next_digit(u, gt.bits));   // in flight during the addition
    acc = pt29r_add(acc, q, k);
  }
  return acc;
}
// a^(p-2), the chain of fe29_inv (Invert, internal/field/field_invert.go:11), in row products
S2K_DEV fer fer_inv_chain(fer a, const fer_consts& k) {
  const fer x2 = fer_mul(fer_mul(a, a, k), a, k);
  const fer x3 = fer_mul(fer_mul(x2, x2, k), a, k);
  const fer x6 = fer_mul(fer_sqr_n(x3, 3, k), x3, k);
  const fer x9 = fer_mul(fer_sqr_n(x6, 3, k), x3, k);
  const fer x11 = fer_mul(fer_sqr_n(x9, 2, k), x2, k);
  const fer x22 = fer_mul(fer_sqr_n(x11, 11, k), x11, k);
  const fer x44 = fer_mul(fer_sqr_n(x22, 22, k), x22, k);
  const fer x88 = fer_mul(fer_sqr_n(x44, 44, k), x44, k);
  const fer x176 = fer_mul(fer_sqr_n(x88, 88, k), x88, k);
  const fer x220 = fer_mul(fer_sqr_n(x176, 44, k), x44, k);
  const fer x223 = fer_mul(fer_sqr_n(x220, 3, k), x3, k);
  fer t = fer_mul(fer_sqr_n(x223, 23, k), x22, k);
  t = fer_mul(fer_sqr_n(t, 5, k), a, k);
  t = fer_mul(fer_sqr_n(t, 3, k), x2, k);
  return fer_mul(fer_sqr_n(t, 2, k), a, k);
}

// PublicKey.Verify (secec/ecdsa.go:171, 392-470) with one WAVEFRONT per signature: the small-batch ladder
__global__ void __launch_bounds__(320)
k_verify_row(uint32_t n, const uint8_t* __restrict__ pub, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig,
             const uint8_t* __restrict__ ssig, uint32_t flags, gt_view gt, uint8_t* __restrict__ out) {
  __shared__ uint32_t tab_all[4][4][8][64];
  __shared__ uint32_t prep_s[4][20];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (wave == 4) {                                              // the preparation wave
    const uint32_t i = blockIdx.x * 4 + lane;
    if (lane < 4 && i < n) scalar_prep_one(i, dig, rsig, ssig, nullptr, flags, prep_s[lane]);
    __syncthreads();
    return;
  }
  const uint32_t sig = blockIdx.x * 4 + wave;
  const fer_consts k = fer_setup(lane);
  bool ok = sig < n;
  if (ok) {
    uint32_t xw[8], yw[8];
    load_be32(xw, pub + (size_t)sig * 64);
    load_be32(yw, pub + (size_t)sig * 64 + 32);
    ok = fe_is_canonical_raw(xw) && fe_is_canonical_raw(yw);
    pt29r Q1;
    Q1.x = fer_from_words(xw, k);
    Q1.y = fer_from_words(yw, k);
    Q1.z = k.j == 0 ? 1u : 0u;
    // y^2 == x^3 + 7 (point_s11n.go:298-307)
    ok = ok && fer_is_zero(fer_add(fer_mul(Q1.y, Q1.y, k), fer_negate(fer_curve_rhs(Q1.x, k), 1, k)), k);
    if (ok) row_table(Q1, tab_all[wave], k, lane);              // (wave-uniform)
  }
  __syncthreads();
  if (sig >= n) return;
  const uint32_t* p = prep_s[wave];
  if (!(ok && (p[16] & PF_OK))) {
    if (lane == 0) out[sig] = 0;
    return;
  }
  const pt29r acc = row_ladder(tab_all[wave], p, gt, k, lane);
  // ---- verdict: R != infinity and x(R) mod n == r (ecdsa.go:450-465), x(R) = X / Z ----
  uint8_t verdict = 0;
  if (!fer_is_zero(acc.z, k)) {
    uint32_t rw[8];
    load_be32(rw, rsig + (size_t)sig * 32);
    bool match = fer_is_zero(fer_add(acc.x, fer_negate(fer_mul(fer_from_words(rw, k), acc.z, k), 1, k)), k);
    if (u256_lt(rw, FE_P_MINUS_N)) {
      uint32_t r2[8];
      u256_add(r2, rw, SC_N);
      match = match || fer_is_zero(fer_add(acc.x, fer_negate(fer_mul(fer_from_words(r2, k), acc.z, k), 1, k)), k);
    }
    verdict = match ? 1 : 0;
  }
  if (lane == 0) out[sig] = verdict;
}

// FOUR LANES per signature (pt29q.h: a point is X | Y | Z | Z on the lanes of a quad, a layer of the complete formulas one lane
// product): the ladders for calls between the wave-per-signature kernels and the lane-per-signature ones - a
// doubling is 550 dependent instructions where a lane needs 1 070, so a lone wave is through its 16 signatures in half the
// time, at twice the instructions per signature.  Behind the preparation kernels (one lane per signature).
// quad_double_mult: u G + k1 Q + k2 lambda(Q) for the quad's signature, Q = (qx, qy) affine and on the curve; u | k1 | k2 | flags
// from the preparation's planes.  The key's table - (2j + 1) Q, j < 8, projective - lives in LDS, tab[entry][limb][lane]: the
// lanes of a quad hold x | y | z | beta * x of the entry (lane 3's copy of z is not read by an addition; lambda's image reads its
// x from there).  Returns this lane's coordinate of the result.
S2K_DEV fe29 quad_double_mult(const fe29& qx, const fe29& qy, uint32_t (*tab)[9][64], const uint32_t* __restrict__ prep, size_t stride,
                              uint32_t sig, uint32_t pf, gt_view gt, uint32_t lane) {
  const uint32_t q = lane & 3u;
  const fe29 beta = fe29_from_words(FE_BETA), one = fe29_one();
  {
    fe29 cur = fe29_pick(q >= 2, fe29_pick(q == 1, qx, qy), one);                  // x | y | 1 | 1
    const fe29 D = pt29q_double(cur, q);
#pragma unroll 1
    for (int j = 0; j < 8; ++j) {
      if (j) cur = pt29q_add(cur, D, q);
      const fe29 bx = fe29_mul(fe29_qperm<S2K_QP(0, 0, 0, 0)>(cur), beta);
      const fe29 st = fe29_pick(q == 3, cur, bx);
#pragma unroll
      for (int i = 0; i < 9; ++i) tab[j][i][lane] = st.n[i];
    }
  }
  // |k1|, |k2| odd, < 2^129
  sc k1 = sc_zero(), k2 = sc_zero();
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    k1.v[w] = prep[(size_t)(8 + w) * stride + sig];
    k2.v[w] = prep[(size_t)(12 + w) * stride + sig];
  }
  k1.v[4] = (pf & PF_K1_B128) ? 1u : 0u;
  k2.v[4] = (pf & PF_K2_B128) ? 1u : 0u;
  const bool neg1 = pf & PF_NEG1, neg2 = pf & PF_NEG2;
  digit_stream d1 = ds_init(k1), d2 = ds_init(k2);
  const uint32_t col3 = lane | 3u;
  auto entry_of = [&](uint32_t w, bool lam, bool sneg) -> fe29 {   // this lane's coordinate of the entry of digit w (2w - 15)
    const bool neg = sneg != (w < 8u);
    const uint32_t e = (w < 8u) ? (7u - w) : (w - 8u);
    const uint32_t col = (lam && q == 0) ? col3 : lane;            // lambda's image: x from the quad's fourth slot
    fe29 a;
#pragma unroll
    for (int i = 0; i < 9; ++i) a.n[i] = tab[e][i][col];
    return fe29_normalize_weak(fe29_cond_negate1(a, neg && q == 1));
  };
  fe29 acc = pt29q_add(entry_of(8u, false, neg1), entry_of(8u, true, neg2), q);     // the top digits are +1 (w = 8)
#pragma unroll 1
  for (int i = 31; i >= 0; --i) {
    const uint32_t w1 = ds_next(d1), w2 = ds_next(d2);
#pragma unroll 1
    for (int j = 0; j < 4; ++j) acc = pt29q_double(acc, q);
    acc = pt29q_add(acc, entry_of(w1, false, neg1), q);
    acc = pt29q_add(acc, entry_of(w2, true, neg2), q);
  }
  uint32_t u[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) u[w] = prep[(size_t)w * stride + sig];
  apt g = gt_load(gt, 0, gt_next_digit(u, gt.bits));
#pragma unroll 1
  for (uint32_t w = 0; w < gt.windows; ++w) {
    const fe29 qc = fe29_pick(q >= 2, fe29_pick(q == 1, fe29_from_words(g.x.v), fe29_from_words(g.y.v)), one);
    if (w + 1 < gt.windows) g = gt_load(gt, w + 1, gt_next_digit(u, gt.bits));
    acc = pt29q_add(acc, qc, q);
  }
  return acc;
}

// PublicKey.Verify (secec/ecdsa.go:171, 392-470), four lanes per signature; behind k_scalar_prep
__global__ void __launch_bounds__(256)
k_verify_quad(uint32_t n, const uint8_t* __restrict__ pub, const uint8_t* __restrict__ rsig, const uint32_t* __restrict__ prep,
              gt_view gt, uint8_t* __restrict__ out, size_t stride) {
  __shared__ uint32_t tab_all[4][8][9][64];
  const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
  const uint32_t sig0 = (blockIdx.x * 256 + threadIdx.x) >> 2;
  const bool live = sig0 < n;
  const uint32_t sig = live ? sig0 : n - 1;                     // (idle quads redo the last signature: no divergence, no store)
  const uint32_t pf = prep[(size_t)16 * stride + sig];
  uint32_t xw[8], yw[8];
  load_be32(xw, pub + (size_t)sig * 64);
  load_be32(yw, pub + (size_t)sig * 64 + 32);
  bool ok = (pf & PF_OK) && fe_is_canonical_raw(xw) && fe_is_canonical_raw(yw);
  fe29 qx = fe29_from_words(xw), qy = fe29_from_words(yw);
  {   // y^2 == x^3 + 7 (point_s11n.go:298-307); an invalid key is replaced by G, its verdict is 0
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    if (!fe29_eq(fe29_sqr(qy), rhs)) ok = false;
    if (!ok) {
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
  }
  const fe29 acc = quad_double_mult(qx, qy, tab_all[threadIdx.x >> 6], prep, stride, sig, pf, gt, lane);
  // ---- verdict: R != infinity and x(R) mod n == r (ecdsa.go:450-465), x(R) = X / Z ----
  const pt29 R = pt29q_gather(acc);
  uint8_t verdict = 0;
  if (ok && !fe29_is_zero(R.z)) {
    uint32_t rw[8];
    load_be32(rw, rsig + (size_t)sig * 32);
    bool match = fe29_eq(R.x, fe29_mul(fe29_from_words(rw), R.z));
    if (u256_lt(rw, FE_P_MINUS_N)) {
      uint32_t r2[8];
      u256_add(r2, rw, SC_N);
      match = match || fe29_eq(R.x, fe29_mul(fe29_from_words(r2), R.z));
    }
    verdict = match ? 1 : 0;
  }
  if (live && q == 0) out[sig] = verdict;
}

// SchnorrPublicKey.Verify (schnorr.go:221-253), four lanes per signature; behind k_schnorr_prep.  The two lifts - of the key and of
// r, the point R would have to be - run side by side on the lanes of the quad (lanes 0, 2: the key; lanes 1, 3: r), one chain of
// lane products; valid iff s G - e P is finite and equals (r, even y), compared projectively.
__global__ void __launch_bounds__(256)
k_schnorr_quad(uint32_t n, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig64, const uint32_t* __restrict__ prep,
               gt_view gt, uint8_t* __restrict__ out, size_t stride) {
  __shared__ uint32_t tab_all[4][8][9][64];
  const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
  const uint32_t sig0 = (blockIdx.x * 256 + threadIdx.x) >> 2;
  const bool live = sig0 < n;
  const uint32_t sig = live ? sig0 : n - 1;
  const uint32_t pf = prep[(size_t)16 * stride + sig];
  uint32_t xw[8], rw[8];
  load_be32(xw, pk + (size_t)sig * 32);
  load_be32(rw, sig64 + (size_t)sig * 64);
  bool ok = (pf & PF_OK) && fe_is_canonical_raw(xw);            // (PF_OK: r < p, s < n)
  fe29 xP = fe29_from_words(xw);
  const fe29 xR = fe29_from_words(rw);
  const fe29 xin = fe29_pick((q & 1u) != 0, xP, xR);
  fe29 c = fe29_mul(fe29_sqr(xin), xin);
  c.n[0] += 7;
  fe29 y;
  const bool has = fe29_sqrt(y, c);
  y = fe29_normalize(y);
  y = fe29_pick((y.n[0] & 1u) != 0, y, fe29_normalize_weak(fe29_negate(y, 1)));       // the even root
  const int hasP = __builtin_amdgcn_mov_dpp(has ? 1 : 0, S2K_QP(0, 0, 0, 0), 0xF, 0xF, true);
  const int hasR = __builtin_amdgcn_mov_dpp(has ? 1 : 0, S2K_QP(1, 1, 1, 1), 0xF, 0xF, true);
  ok = ok && hasP && hasR;
  fe29 yP = fe29_qperm<S2K_QP(0, 0, 0, 0)>(y);
  const fe29 yR = fe29_qperm<S2K_QP(1, 1, 1, 1)>(y);
  if (!ok) {                                                    // keep the arithmetic on the curve; the verdict is 0
    xP = fe29_from_words(FE_GX);
    yP = fe29_from_words(FE_GY);
  }
  const fe29 acc = quad_double_mult(xP, yP, tab_all[threadIdx.x >> 6], prep, stride, sig, pf, gt, lane);
  const pt29 R = pt29q_gather(acc);
  uint8_t verdict = 0;
  if (ok && !fe29_is_zero(R.z))
    verdict = (fe29_eq(R.x, fe29_mul(xR, R.z)) && fe29_eq(R.y, fe29_mul(yR, R.z))) ? 1 : 0;
  if (live && q == 0) out[sig] = verdict;
}

// RecoverPublicKey (ecdsa.go:244-282), four lanes per item; between k_scalar_prep (u1 = -e/r, u2 = s/r, the id in the flags) and
// k_affine_finish<MODE_RECOVER>, which gets the Jacobian triple (X Z, Y Z^2, Z) it expects.  Items without a key: ok = 0 (their
// record was zeroed before).
__global__ void __launch_bounds__(256)
k_recover_quad(uint32_t n, const uint8_t* __restrict__ rsig, const uint32_t* __restrict__ prep, gt_view gt, uint32_t* __restrict__ fin,
               uint8_t* __restrict__ out, size_t stride) {
  __shared__ uint32_t tab_all[4][8][9][64];
  const uint32_t lane = threadIdx.x & 63u, q = lane & 3u;
  const uint32_t sig0 = (blockIdx.x * 256 + threadIdx.x) >> 2;
  const bool live = sig0 < n;
  const uint32_t sig = live ? sig0 : n - 1;
  const uint32_t pf = prep[(size_t)16 * stride + sig];
  const uint32_t rid = (pf >> 8) & 3u;
  uint32_t xw[8];
  load_be32(xw, rsig + (size_t)sig * 32);
  bool ok = (pf & PF_OK) != 0;                                  // (r, s in range, id < 4, r + n < p where the id asks for it)
  if (ok && (rid & 2u)) u256_add(xw, xw, SC_N);
  fe29 qx = fe29_from_words(xw);
  fe29 c = fe29_mul(fe29_sqr(qx), qx);
  c.n[0] += 7;
  fe29 qy;
  ok = fe29_sqrt(qy, c) && ok;
  qy = fe29_normalize(qy);
  qy = fe29_pick(((qy.n[0] & 1u) != 0) != ((rid & 1u) != 0), qy, fe29_normalize_weak(fe29_negate(qy, 1)));
  if (!ok) {
    qx = fe29_from_words(FE_GX);
    qy = fe29_from_words(FE_GY);
  }
  const fe29 acc = quad_double_mult(qx, qy, tab_all[threadIdx.x >> 6], prep, stride, sig, pf, gt, lane);
  const pt29 R = pt29q_gather(acc);
  const bool finite = !fe29_is_zero(R.z);                       // identity: NewPublicKeyFromPoint fails (secec.go:206-209)
  if (live && q == 0) {
    if (ok && finite) {
      const fe29 zz = fe29_sqr(R.z);
      fq_store(fin, stride, sig, 0, fe29_mul(R.x, R.z));
      fq_store(fin, stride, sig, 1, fe29_mul(R.y, zz));
      fq_store(fin, stride, sig, 2, R.z);
      out[sig] = VERDICT_PENDING;
    } else {
      out[sig] = 0;
    }
  }
}

// ---------------------------------------------------------------------------------------
// Affine epilogue of the BIP-340 and recovery paths.  Both need x/Z^2, y/Z^3 of the ladder's
// result (the reference inverts per point: XBytes / IsYOdd, point_s11n.go:119-134); here one
// thread takes FIN_M pending lanes (strided, so loads coalesce), multiplies their Z together,
// inverts once and walks back (Montgomery's trick): 1 inversion per 16 results.
//   MODE_SCHNORR: valid iff y even and x == r (verifySchnorrSignatureR, schnorr.go:451-478)
//   MODE_RECOVER: writes the 65-byte key record (RecoverPublicKey, ecdsa.go:244-282)
// Lanes that are not pending (already rejected, or queued for the complete kernel) ride along
// with Z = 1.  Scratch: element 3 of the lane's fin region takes the prefix product.
// ---------------------------------------------------------------------------------------
constexpr int FIN_M = 16;
template <int MODE>
__global__ void __launch_bounds__(64)
k_affine_finish(uint32_t n, uint32_t T, const uint8_t* __restrict__ rsig, uint32_t* __restrict__ fin,
                uint8_t* __restrict__ out, size_t stride, uint8_t* __restrict__ out_pts) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= T) return;
  fe29 acc = fe29_one();
#pragma unroll 1
  for (int j = 0; j < FIN_M; ++j) {
    size_t i = (size_t)t + (size_t)j * T;
    if (i >= n) break;
    if (out[i] == VERDICT_PENDING) acc = fe29_mul(acc, fq_load(fin, stride, i, 2));
    fq_store(fin, stride, i, 3, acc);
  }
  fe29 inv = fe29_inv_gcd(acc);   // safegcd mod p (fe29_inv.h): a quarter of the Fermat chain's instructions, and this kernel is latency bound
#pragma unroll 1
  for (int j = FIN_M - 1; j >= 0; --j) {
    size_t i = (size_t)t + (size_t)j * T;
    if (i >= n || out[i] != VERDICT_PENDING) continue;
    fe29 prev = j > 0 ? fq_load(fin, stride, i - T, 3) : fe29_one();
    fe29 zi = fe29_mul(inv, prev);                     // 1 / Z_i
    inv = fe29_mul(inv, fq_load(fin, stride, i, 2));
    fe29 zi2 = fe29_sqr(zi);
    fe29 x = fe29_mul(fq_load(fin, stride, i, 0), zi2);
    fe29 y = fe29_normalize(fe29_mul(fe29_mul(fq_load(fin, stride, i, 1), zi2), zi));
    if constexpr (MODE == MODE_RECOVER) {
      uint32_t xw[8], yw[8];
      fe29_to_words(xw, fe29_normalize(x));
      fe29_to_words(yw, y);
      uint8_t* rec = out_pts + i * 65;
      rec[0] = 0x04;
      store_be32_unaligned(rec + 1, xw);
      store_be32_unaligned(rec + 33, yw);
      out[i] = 1;
    } else {
      uint32_t rw[8];
      load_be32(rw, rsig + i * 64);
      out[i] = ((y.n[0] & 1u) == 0 && fe29_eq(x, fe29_from_words(rw))) ? 1 : 0;
    }
  }
}

// ---------------------------------------------------------------------------------------
// BIP-340: scalar preparation (no inversion needed: R = s*G + (-e)*P) and complete fallback
// ---------------------------------------------------------------------------------------
S2K_DEV void schnorr_msg(const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs, uint32_t msg_len,
                         size_t i, const uint8_t*& m, uint32_t& len) {
  if (offs) {
    m = msgs + offs[i];
    len = (uint32_t)(offs[i + 1] - offs[i]);
  } else {
    m = msgs + i * (size_t)msg_len;
    len = msg_len;
  }
}
// parseSchnorrSignature (schnorr.go:420-449): r < p, s < n, e = H(r || P || m) mod n.
// Returns ok; s and e as plain scalars.
S2K_DEV bool schnorr_parse(size_t i, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig,
                           const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs, uint32_t msg_len,
                           sc& s_out, sc& e_out) {
  uint32_t r_le[8], pk_le[8];
  load_be32(r_le, sig + i * 64);
  load_be32(s_out.v, sig + i * 64 + 32);
  load_be32(pk_le, pk + i * 32);
  bool ok = fe_is_canonical_raw(r_le) && sc_is_canonical_raw(s_out.v);
  uint32_t r_be[8], pk_be[8], dg[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    r_be[w] = r_le[7 - w];
    pk_be[w] = pk_le[7 - w];
  }
  const uint8_t* m;
  uint32_t len;
  schnorr_msg(msgs, offs, msg_len, i, m, len);
  bip340_challenge(dg, r_be, pk_be, m, len);
  uint32_t e_raw[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) e_raw[w] = dg[7 - w];
  e_out = sc_reduce_once(e_raw);
  if (!ok) s_out = sc_zero();
  return ok;
}

__global__ void __launch_bounds__(256)
k_schnorr_prep(uint32_t n, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig,
               const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs, uint32_t msg_len,
               uint32_t* __restrict__ prep, size_t stride) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  sc s, e;
  bool ok = schnorr_parse(i, pk, sig, msgs, offs, msg_len, s, e);
  sc u2 = sc_neg(e);                                   // schnorr.go:244
  sc k1, k2;
  bool neg1, neg2;
  sc_split_glv_odd(u2, k1, neg1, k2, neg2);
  uint32_t f = (ok ? PF_OK : 0) | (neg1 ? PF_NEG1 : 0) | (neg2 ? PF_NEG2 : 0) |
               (k1.v[4] ? PF_K1_B128 : 0) | (k2.v[4] ? PF_K2_B128 : 0);
#pragma unroll
  for (int w = 0; w < 8; ++w) prep[(size_t)w * stride + i] = s.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) prep[(size_t)(8 + w) * stride + i] = k1.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) prep[(size_t)(12 + w) * stride + i] = k2.v[w];
  prep[(size_t)16 * stride + i] = f;
}

// SchnorrPublicKey.Verify (schnorr.go:221-253) with one WAVEFRONT per signature (small batches; the block's fifth
// wave hashes the challenges and splits -e meanwhile).  lift_x of the key and of r - the point R would have to be - run as
// ONE chain of row products (rows 0, 1: the key; rows 2, 3: r): the signature is valid iff s G - e P is the point (r, even
// y), compared projectively, so there is no inversion.
// FUSED: five waves as described; the hashing makes that instance a 248-register kernel (two waves per SIMD), enough for up
// to 1024 signatures (1280 waves).  !FUSED (larger calls): four waves behind k_schnorr_prep, whose planes they read - 62 registers.
template <bool FUSED>
__global__ void __launch_bounds__(FUSED ? 320 : 256)
k_schnorr_row(uint32_t n, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig64, const uint8_t* __restrict__ msgs,
              const uint64_t* __restrict__ offs, uint32_t msg_len, const uint32_t* __restrict__ prep, size_t stride, gt_view gt,
              uint8_t* __restrict__ out) {
  __shared__ uint32_t tab_all[4][4][8][64];
  __shared__ uint32_t prep_s[4][20];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if constexpr (!FUSED) {
    const uint32_t i = blockIdx.x * 4 + wave;
    if (lane < 17 && i < n) prep_s[wave][lane] = prep[(size_t)lane * stride + i];
  }
  if (FUSED && wave == 4) {                                     // the preparation wave (k_schnorr_prep for four signatures)
    const uint32_t i = blockIdx.x * 4 + lane;
    if (lane < 4 && i < n) {
      sc s, e;
      const bool ok = schnorr_parse(i, pk, sig64, msgs, offs, msg_len, s, e);     // (r < p, s < n)
      sc k1, k2;
      bool neg1, neg2;
      sc_split_glv_odd(sc_neg(e), k1, neg1, k2, neg2);          // schnorr.go:244
      uint32_t* o = prep_s[lane];
#pragma unroll
      for (int w = 0; w < 8; ++w) o[w] = s.v[w];
#pragma unroll
      for (int w = 0; w < 4; ++w) o[8 + w] = k1.v[w];
#pragma unroll
      for (int w = 0; w < 4; ++w) o[12 + w] = k2.v[w];
      o[16] = (ok ? PF_OK : 0) | (neg1 ? PF_NEG1 : 0) | (neg2 ? PF_NEG2 : 0) | (k1.v[4] ? PF_K1_B128 : 0) | (k2.v[4] ? PF_K2_B128 : 0);
    }
    __syncthreads();
    return;
  }
  const uint32_t sig = blockIdx.x * 4 + wave;
  const fer_consts k = fer_setup(lane);
  bool ok = sig < n;
  fer xR = 0, yR = 0;
  if (ok) {
    uint32_t xw[8], rw[8];
    load_be32(xw, pk + (size_t)sig * 32);
    load_be32(rw, sig64 + (size_t)sig * 64);
    ok = fe_is_canonical_raw(xw);
    const fer xP = fer_from_words(xw, k);
    xR = fer_from_words(rw, k);
    const fer c = fer_curve_rhs(fer_sel2(k, xP, xR), k);
    const fer y = fer_sqrt_chain(c, k);
    const bool is_root = fer_is_zero(fer_add(fer_mul(y, y, k), fer_negate(c, 1, k)), k);     // (row by row)
    const bool odd = fer_is_odd(y, k);
    const fer y_even = fer_norm(odd ? fer_negate(y, 1, k) : y, k);
    const uint64_t roots = __builtin_amdgcn_ballot_w64(is_root);
    ok = ok && (roots & 1u) && ((roots >> 32) & 1u);            // both lifts exist (lane 0: the key, lane 32: r)
    pt29r Q1;
    Q1.x = xP;
    fer_halves(y_even, Q1.y, yR);
    Q1.z = k.j == 0 ? 1u : 0u;
    if (ok) row_table(Q1, tab_all[wave], k, lane);              // (wave-uniform)
  }
  __syncthreads();
  if (sig >= n) return;
  const uint32_t* p = prep_s[wave];
  if (!(ok && (p[16] & PF_OK))) {
    if (lane == 0) out[sig] = 0;
    return;
  }
  const pt29r acc = row_ladder(tab_all[wave], p, gt, k, lane);
  // R = s G - e P is finite, x(R) = r and y(R) is even: X = r Z and Y = y_r Z with Z != 0
  uint8_t verdict = 0;
  if (!fer_is_zero(acc.z, k)) {
    const bool mx = fer_is_zero(fer_add(acc.x, fer_negate(fer_mul(xR, acc.z, k), 1, k)), k);
    const bool my = fer_is_zero(fer_add(fer_norm(acc.y, k), fer_negate(fer_mul(yR, acc.z, k), 1, k)), k);
    verdict = (mx && my) ? 1 : 0;
  }
  if (lane == 0) out[sig] = verdict;
}

// SchnorrPublicKey.Verify (schnorr.go:221-253) with complete formulas
S2K_DEV uint8_t schnorr_verify_complete(size_t idx, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig,
                                        const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs,
                                        uint32_t msg_len, gt_view gt, uint32_t* __restrict__ qt,
                                        size_t stride) {
  sc s, e;
  bool ok = schnorr_parse(idx, pk, sig, msgs, offs, msg_len, s, e);
  apt P;
  load_be32(P.x.v, pk + idx * 32);
  ok = ok && fe_is_canonical_raw(P.x.v);
  fe y;
  bool has = fe_sqrt(y, fe_curve_rhs(P.x));
  ok = ok && has;
  if (!ok) {
    P.x = fe_from_limbs(FE_GX);
    y = fe_from_limbs(FE_GY);
  }
  y = fe_normalize(y);
  P.y = fe_normalize(fe_select((y.v[0] & 1u) != 0, y, fe_neg(y)));
  pt R = pt_add_complete(pt_base_mul(gt, s.v), pt_mul_glv(sc_neg(e), P, qt, stride, idx));
  apt a;
  bool finite = pt_to_affine(a, R);
  uint32_t r_le[8];
  load_be32(r_le, sig + idx * 64);
  return (ok && finite && (a.y.v[0] & 1u) == 0 && u256_eq(a.x.v, r_le)) ? 1 : 0;
}

__global__ void __launch_bounds__(256)
k_schnorr_fallback(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                   const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig, const uint8_t* __restrict__ msgs,
                   const uint64_t* __restrict__ offs, uint32_t msg_len, uint8_t* __restrict__ out,
                   gt_view gt, uint32_t* __restrict__ qt, size_t stride) {
  // all_n != 0: diagnostic mode, every signature through the complete path
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
    out[idx] = schnorr_verify_complete(idx, pk, sig, msgs, offs, msg_len, gt, qt, stride);
  }
}

// RecoverPublicKey (ecdsa.go:244-282) with one WAVEFRONT per item (small batches; the block's fifth wave computes
// u1 = -e/r, u2 = s/r meanwhile): R = (r or r + n, the root of the asked parity) by a chain of row products, Q = u1 G + u2 R
// on the complete formulas, 1/Z by the inversion chain in row products, the 65-byte record written by the wave.  Items
// without a key get ok = 0 and the zero record.
__global__ void __launch_bounds__(320)
k_recover_row(uint32_t n, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig,
              const uint8_t* __restrict__ recid, gt_view gt, uint8_t* __restrict__ out, uint8_t* __restrict__ out_pts) {
  __shared__ uint32_t tab_all[4][4][8][64];
  __shared__ uint32_t prep_s[4][20];
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (wave == 4) {                                              // the preparation wave
    const uint32_t i = blockIdx.x * 4 + lane;
    if (lane < 4 && i < n) scalar_prep_one(i, dig, rsig, ssig, recid, 0u, prep_s[lane]);
    __syncthreads();
    return;
  }
  const uint32_t sig = blockIdx.x * 4 + wave;
  const fer_consts k = fer_setup(lane);
  bool ok = sig < n;
  if (ok) {
    const uint32_t rid = recid[sig];
    uint32_t xw[8];
    load_be32(xw, rsig + (size_t)sig * 32);
    if (rid & 2u) u256_add(xw, xw, SC_N);                       // (that r + n < p holds is the preparation wave's check)
    pt29r Q1;
    Q1.x = fer_from_words(xw, k);
    Q1.z = k.j == 0 ? 1u : 0u;
    const fer c = fer_curve_rhs(Q1.x, k);
    const fer y = fer_sqrt_chain(c, k);
    ok = fer_is_zero(fer_add(fer_mul(y, y, k), fer_negate(c, 1, k)), k);
    Q1.y = fer_norm(fer_is_odd(y, k) != ((rid & 1u) != 0) ? fer_negate(y, 1, k) : y, k);
    if (ok) row_table(Q1, tab_all[wave], k, lane);              // (wave-uniform)
  }
  __syncthreads();
  if (sig >= n) return;
  uint8_t* rec = out_pts + (size_t)sig * 65;
  auto no_key = [&]() {                                         // ok = 0 and the zero record (RecoverPublicKey's error)
    rec[lane] = 0;
    if (lane == 0) {
      rec[64] = 0;
      out[sig] = 0;
    }
  };
  const uint32_t* p = prep_s[wave];
  if (!(ok && (p[16] & PF_OK))) {                               // (r, s in range, id < 4, r + n < p where the id asks for it)
    no_key();
    return;
  }
  const pt29r acc = row_ladder(tab_all[wave], p, gt, k, lane);
  if (fer_is_zero(acc.z, k)) {                                  // identity: NewPublicKeyFromPoint fails (secec.go:206-209)
    no_key();
    return;
  }
  const fer zi = fer_inv_chain(acc.z, k);
  uint32_t xw[8], yw[8];
  fe29_to_words(xw, fe29_normalize(fer_to_fe29(fer_mul(acc.x, zi, k))));
  fe29_to_words(yw, fe29_normalize(fer_to_fe29(fer_mul(fer_norm(acc.y, k), zi, k))));
  if (lane == 0) {
    rec[0] = 0x04;
    store_be32_unaligned(rec + 1, xw);
    store_be32_unaligned(rec + 33, yw);
    out[sig] = 1;
  }
}

// RecoverPublicKey (ecdsa.go:244-282) for one item with complete formulas
S2K_DEV uint8_t recover_complete(size_t idx, const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig,
                                 const uint8_t* __restrict__ ssig, const uint8_t* __restrict__ recid,
                                 uint8_t* __restrict__ out_pts, gt_view gt,
                                 uint32_t* __restrict__ qt, size_t stride) {
  sc r, s;
  uint32_t e_raw[8];
  load_be32(r.v, rsig + idx * 32);
  load_be32(s.v, ssig + idx * 32);
  load_be32(e_raw, dig + idx * 32);
  uint32_t rid = recid[idx];
  bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s) && rid < 4 &&
            (!(rid & 2u) || u256_lt(r.v, FE_P_MINUS_N));
  apt R;
#pragma unroll
  for (int i = 0; i < 8; ++i) R.x.v[i] = r.v[i];
  if (ok && (rid & 2u)) u256_add(R.x.v, R.x.v, SC_N);
  fe y;
  bool has = fe_sqrt(y, fe_curve_rhs(R.x));
  ok = ok && has;
  if (!ok) {
    R.x = fe_from_limbs(FE_GX);
    y = fe_from_limbs(FE_GY);
  }
  y = fe_normalize(y);
  R.y = fe_normalize(fe_select(((y.v[0] & 1u) != 0) != ((rid & 1u) != 0), y, fe_neg(y)));
  sc e = sc_reduce_once(e_raw);
  sc r_inv_m = sc_mont_inv(sc_to_mont(r));
  sc u1 = sc_montmul(sc_neg(e), r_inv_m), u2 = sc_montmul(s, r_inv_m);
  pt Q = pt_add_complete(pt_base_mul(gt, u1.v), pt_mul_glv(u2, R, qt, stride, idx));
  uint8_t* rec = out_pts + idx * 65;
  apt a;
  bool finite = pt_to_affine(a, Q);              // identity: NewPublicKeyFromPoint fails (secec.go:206-209)
  if (!(ok && finite)) {
    for (int i = 0; i < 65; ++i) rec[i] = 0;
    return 0;
  }
  rec[0] = 0x04;
  store_be32_unaligned(rec + 1, a.x.v);
  store_be32_unaligned(rec + 33, a.y.v);
  return 1;
}

__global__ void __launch_bounds__(256)
k_recover_fallback(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                   const uint8_t* __restrict__ dig, const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig,
                   const uint8_t* __restrict__ recid, uint8_t* __restrict__ ok_out, uint8_t* __restrict__ out_pts,
                   gt_view gt, uint32_t* __restrict__ qt, size_t stride) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
    ok_out[idx] = recover_complete(idx, dig, rsig, ssig, recid, out_pts, gt, qt, stride);
  }
}

// x/Z, y/Z of a projective 9x29 point (canonical words); false for the identity
S2K_DEV bool p29_to_affine_words(uint32_t xw[8], uint32_t yw[8], const pt29& p) {
  if (fe29_is_zero(p.z)) return false;
  fe29 zi = fe29_inv_gcd(fe29_normalize_weak(p.z));
  fe29_to_words(xw, fe29_normalize(fe29_mul(p.x, zi)));
  fe29_to_words(yw, fe29_normalize(fe29_mul(p.y, zi)));
  return true;
}
// lift_x on the 9x29 field: y with the wanted parity for a canonical x (words), or false
S2K_DEV bool lift_x29(fe29& x, fe29& y, const uint32_t xw[8], bool want_odd) {
  x = fe29_from_words(xw);
  fe29 rhs = fe29_mul(fe29_sqr(x), x);
  rhs.n[0] += 7;
  if (!fe29_sqrt(y, rhs)) return false;
  y = fe29_normalize(y);
  y = fe29_select(((y.n[0] & 1u) != 0) != want_odd, y, fe29_normalize_weak(fe29_negate(y, 1)));
  return true;
}

// The worklists of the BIP-340 and recovery paths, same verifier core (9x29 complete formulas) as
// the ECDSA worklist; the reference-shaped 8x32 versions above stay for S2K_ECDSA_FORCE_COMPLETE.
__global__ void __launch_bounds__(256)
k_schnorr_worklist(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, const uint8_t* __restrict__ pk,
                   const uint8_t* __restrict__ sig, const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs,
                   uint32_t msg_len, uint8_t* __restrict__ out, gt_view gt, uint32_t* __restrict__ qt,
                   size_t stride) {
  const uint32_t count = *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    const size_t idx = wl[w];
    sc s, e;
    bool ok = schnorr_parse(idx, pk, sig, msgs, offs, msg_len, s, e);
    uint32_t xw[8];
    load_be32(xw, pk + idx * 32);
    ok = ok && fe_is_canonical_raw(xw);
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xw[i] = FE_GX[i];
    }
    fe29 px, py;
    if (!lift_x29(px, py, xw, false)) {   // not an x-coordinate of the curve (NewSchnorrPublicKey, schnorr.go:257-275)
      ok = false;
      px = fe29_from_words(FE_GX);
      py = fe29_from_words(FE_GY);
    }
    pt29 R = dsm_complete29(s, sc_neg(e), px, py, gt, qt, stride, w);   // s*G - e*P (schnorr.go:244); table column = worklist position
    uint32_t rx[8], ry[8], r_le[8];
    bool finite = p29_to_affine_words(rx, ry, R);
    load_be32(r_le, sig + idx * 64);
    out[idx] = (ok && finite && (ry[0] & 1u) == 0 && u256_eq(rx, r_le)) ? 1 : 0;   // verifySchnorrSignatureR (:451-478)
  }
}

__global__ void __launch_bounds__(256)
k_recover_worklist(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, const uint8_t* __restrict__ dig,
                   const uint8_t* __restrict__ rsig, const uint8_t* __restrict__ ssig, const uint8_t* __restrict__ recid,
                   uint8_t* __restrict__ ok_out, uint8_t* __restrict__ out_pts, gt_view gt,
                   uint32_t* __restrict__ qt, size_t stride) {
  const uint32_t count = *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    const size_t idx = wl[w];
    sc r, s;
    uint32_t e_raw[8], xw[8];
    load_be32(r.v, rsig + idx * 32);
    load_be32(s.v, ssig + idx * 32);
    load_be32(e_raw, dig + idx * 32);
    const uint32_t rid = recid[idx];
    bool ok = sc_is_canonical_raw(r.v) && !sc_is_zero(r) && sc_is_canonical_raw(s.v) && !sc_is_zero(s) && rid < 4 &&
              (!(rid & 2u) || u256_lt(r.v, FE_P_MINUS_N));   // RecoverPoint (point_s11n.go:245-282)
#pragma unroll
    for (int i = 0; i < 8; ++i) xw[i] = r.v[i];
    if (ok && (rid & 2u)) u256_add(xw, xw, SC_N);
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xw[i] = FE_GX[i];
      r = sc_zero();
      r.v[0] = 1;
    }
    fe29 px, py;
    if (!lift_x29(px, py, xw, (rid & 1u) != 0)) {
      ok = false;
      px = fe29_from_words(FE_GX);
      py = fe29_from_words(FE_GY);
    }
    sc e = sc_reduce_once(e_raw);
    sc26 r_inv_m = sc26_mont_inv(sc26_to_mont(sc26_from_sc(r)));
    sc u1 = sc26_to_sc(sc26_mm(sc26_from_sc(sc_neg(e)), r_inv_m)), u2 = sc26_to_sc(sc26_mm(sc26_from_sc(s), r_inv_m));
    pt29 Q = dsm_complete29(u1, u2, px, py, gt, qt, stride, idx);   // (-e/r) G + (s/r) R (ecdsa.go:244-282)
    uint32_t qxw[8], qyw[8];
    bool finite = p29_to_affine_words(qxw, qyw, Q);   // identity: NewPublicKeyFromPoint fails (secec.go:206-209)
    uint8_t* rec = out_pts + idx * 65;
    if (!(ok && finite)) {
      for (int i = 0; i < 65; ++i) rec[i] = 0;
      ok_out[idx] = 0;
    } else {
      rec[0] = 0x04;
      store_be32_unaligned(rec + 1, qxw);
      store_be32_unaligned(rec + 33, qyw);
      ok_out[idx] = 1;
    }
  }
}

// ---------------------------------------------------------------------------------------
// u1*G + u2*P for arbitrary (u1, u2, P) through the verification ladder (S2K_IMPL_FAST of
// s2k_double_scalar_mult_basepoint_batch_ex): the same table, signed-digit ladder and generator
// additions as a verification, with the affine result written as a 65-byte record.
//   k_hot_prep          65-byte records -> 64-byte affine points + prep planes (u1, odd GLV
//                       halves of u2); identity / malformed records go straight to the worklist
//   k_verify_fast<MODE_POINT>, k_affine_finish<MODE_RECOVER>
//   k_point_fallback    worklist lanes by the complete path (pt_base_mul + pt_mul_glv)
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_hot_prep(uint32_t n, const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
           uint8_t* __restrict__ pub64, uint32_t* __restrict__ prep, size_t stride, uint32_t* __restrict__ wl_count,
           uint32_t* __restrict__ wl, uint32_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* rec = pts65 + i * 65;
  apt a;
  load_be32_unaligned(a.x.v, rec + 1);
  load_be32_unaligned(a.y.v, rec + 33);
  bool finite = rec[0] == 0x04 && fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v) && apt_on_curve(a);
  if (!finite && rec[0] != 0x00) atomicOr(status, 1u);   // not a Point the reference could have built
  if (!finite) {                                          // identity: the complete kernel returns u1*G
    a.x = fe_from_limbs(FE_GX);
    a.y = fe_from_limbs(FE_GY);
    uint32_t pos = atomicAdd(wl_count, 1u);
    wl[pos] = (uint32_t)i;
  }
  store_be32(pub64 + i * 64, a.x.v);
  store_be32(pub64 + i * 64 + 32, a.y.v);
  uint32_t raw[8];
  sc k1, k2;
  bool neg1, neg2;
  load_be32(raw, u1 + i * 32);        // zeros for a plain scalar multiplication
  sc v1 = sc_reduce_once(raw);
  load_be32(raw, u2 + i * 32);
  sc_split_glv_odd(sc_reduce_once(raw), k1, neg1, k2, neg2);
  uint32_t f = (finite ? PF_OK : 0) | (neg1 ? PF_NEG1 : 0) | (neg2 ? PF_NEG2 : 0) | (k1.v[4] ? PF_K1_B128 : 0) |
               (k2.v[4] ? PF_K2_B128 : 0);
#pragma unroll
  for (int w = 0; w < 8; ++w) prep[(size_t)w * stride + i] = v1.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) prep[(size_t)(8 + w) * stride + i] = k1.v[w];
#pragma unroll
  for (int w = 0; w < 4; ++w) prep[(size_t)(12 + w) * stride + i] = k2.v[w];
  prep[(size_t)16 * stride + i] = f;
}

// ALL: every item (S2K_IMPL_COMPLETE); otherwise the worklist.  u1 is always present (the host
// passes zeros for a plain scalar multiplication): the conditional form of this kernel
// (`if (u1)` around the generator part, `all_n ? w : wl[w]` for the index) was miscompiled by
// hipcc 7.2 -- per-lane garbage for half of the inputs, reproduced standalone in tools/dbg/ --
// so the kernel keeps to the shape of k_recover_fallback, which is not.
template <bool ALL>
__global__ void __launch_bounds__(256)
k_point_fallback(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count;
  if constexpr (ALL) count = n; else count = *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx;
    if constexpr (ALL) idx = w; else idx = wl[w];
    const uint8_t* rec = pts65 + idx * 65;
    apt a;
    load_be32_unaligned(a.x.v, rec + 1);
    load_be32_unaligned(a.y.v, rec + 33);
    bool finite = rec[0] == 0x04 && fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v) && apt_on_curve(a);
    if (!finite && rec[0] != 0x00) atomicOr(status, 1u);
    if (!finite) {   // keep the arithmetic on the curve; the term is masked below
      a.x = fe_from_limbs(FE_GX);
      a.y = fe_from_limbs(FE_GY);
    }
    uint32_t raw1[8], raw2[8];
    load_be32(raw1, u1 + idx * 32);
    load_be32(raw2, u2 + idx * 32);
    sc v1 = sc_reduce_once(raw1), v2 = sc_reduce_once(raw2);
    pt rq = pt_mul_glv(v2, a, qt, stride, idx);
    rq = pt_select(!finite, rq, pt_identity());
    pt res = pt_add_complete(pt_base_mul(gt, v1.v), rq);   // point_mul_glv.go:316
    uint8_t* o = out65 + idx * 65;
    apt r;
    if (!pt_to_affine(r, res)) {
      for (int j = 0; j < 65; ++j) o[j] = 0;
    } else {
      o[0] = 0x04;
      store_be32_unaligned(o + 1, r.x.v);
      store_be32_unaligned(o + 33, r.y.v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// per-stage timing (s2k_ctx_profile): six events per call, around scalar preparation | grouping and
// per-key tables | ladder (keyed, or the general one when grouping is off) | general ladder over the
// ungrouped rest | complete-formula worklist
constexpr int PROF_EV = 6;
static inline void prof_mark(s2k_ctx* ctx, hipStream_t st, int slot) {
  if (!ctx->prof_on || ctx->prof_used + PROF_EV > ctx->prof_cap) return;
  (void)hipEventRecord(ctx->prof_ev[ctx->prof_used + slot], st);
  if (slot == PROF_EV - 1) ctx->prof_used += PROF_EV;
}

// Front end of the grouped flows (ECDSA and BIP-340 verification): everything up to the ladders.
// Caller's stream: grouping by key (k_key_insert / k_key_alloc / k_key_place), then the per-key tables.
// Second stream: the scalar preparation (`launch_prep`), then the generator part u1*G in two pieces.  These
// have nothing to do with the keys and the grouping / table kernels are short of work for the multipliers
// on their own (the doubling chain of the tables is one lane per KEY, the scaling pass is memory bound).
// The first piece of the generator part runs beside the chain, the second once k_key_odd (which does keep
// the multipliers busy) is through.  Returns with the caller's stream waiting for the second.
// Stage times (s2k_ctx_profile_read_stages): [0] grouping, [1] tables and whatever is left of the second
// stream's work.
// S2K_KEYS_ADAPTIVE (engine_internal.h: kga_*): takes in the notes that have arrived, decides whether this call looks for
// repeated keys, and if it does, where its k_key_counts leaves its note.  No device call that waits.
static bool kg_adaptive_decide(s2k_ctx* ctx) {
  s2k_ctx* own = ctx->parent ? ctx->parent : ctx;
  if (!own->kga_note) {
    void* p = nullptr;
    if (hipHostMalloc(&p, KG_ADAPT_SLOTS * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return true;                                     // no note, no learning: S2K_KEYS_AUTO
    }
    memset(p, 0, KG_ADAPT_SLOTS * sizeof(unsigned long long));
    own->kga_note = (unsigned long long*)p;
  }
  for (;;) {                                           // notes in the order of their numbers; a slot overwritten by a later call: that one
    unsigned long long next = 0;
    for (uint32_t i = 0; i < KG_ADAPT_SLOTS; ++i) {
      const unsigned long long v = __atomic_load_n(&own->kga_note[i], __ATOMIC_ACQUIRE);
      const uint32_t sq = (uint32_t)(v >> 32);
      if (v && (int32_t)(sq - own->kga_seen) > 0 && (!next || (int32_t)(sq - (uint32_t)(next >> 32)) < 0)) next = v;
    }
    if (!next) break;
    own->kga_seen = (uint32_t)(next >> 32);
    ++own->kga_observed;
    if ((uint32_t)next == 0) {
      ++own->kga_miss_streak;
    } else {
      own->kga_miss_streak = 0;
      own->kga_skip_left = 0;
      own->kga_skipping = false;
    }
  }
  if (own->kga_miss_streak >= KG_ADAPT_MISSES) {
    if (!own->kga_skipping) {
      own->kga_skipping = true;
      own->kga_skip_left = KG_ADAPT_SKIP;
    }
    if (own->kga_skip_left) {
      --own->kga_skip_left;
      ++own->kga_skipped;
      return false;
    }
    own->kga_skip_left = KG_ADAPT_SKIP;                // this call looks again; the next ones do not, unless it finds something
    ++own->kga_probes;
  } else {
    own->kga_skipping = false;
  }
  if (++own->kga_seq == 0) own->kga_seq = 1;           // (0 marks an empty slot)
  ctx->kg_note_seq = own->kga_seq;
  ctx->kg_note_dst = own->kga_note + own->kga_seq % KG_ADAPT_SLOTS;
  return true;
}

template <class PrepFn>
static int grouped_front_forked(s2k_ctx* ctx, hipStream_t st, size_t n, const uint8_t* d_keys, int key_bytes, uint32_t* prep,
                                uint32_t* gp, size_t stride, PrepFn launch_prep, key_groups* kg, bool gp_in_prep) {
  // One stream (a child context of submit / wait): the grouping and the tables first - they need the keys only, which
  // arrive first -, then the preparation and the generator part; other tickets' kernels fill the machine meanwhile.
  const bool one_stream = ctx->s_aux == st;
  // gp_in_prep: the caller's launch_prep has enqueued the generator part as well (piecewise, as its inputs arrive)
  const uint32_t n_first = (gp_in_prep || one_stream) ? (gp_in_prep ? (uint32_t)n : 0u)
                                                      : (uint32_t)((n * (size_t)ctx->gp_first_percent / 100) & ~(size_t)255);
  if (!one_stream) {
    launch_prep(ctx->s_aux);
    HIP_TRY(ctx, hipGetLastError());
    if (n_first && !gp_in_prep) {
      k_generator_part<<<blocks_for(n_first), 256, 0, ctx->s_aux>>>(0u, n_first, prep, ctx->gt_call, gp, stride);
      HIP_TRY(ctx, hipGetLastError());
    }
  }
  int rc = s2k_internal_key_group(ctx, n, d_keys, key_bytes, st, kg);
  if (rc) return rc;
  kg->gp = gp;
  kg->part = 0;
  kg->nparts = ctx->kg_parts;
  prof_mark(ctx, st, 1);
  rc = s2k_internal_key_chains(ctx, d_keys, st, kg);
  if (rc) return rc;
  // Two-part flow (S2K_KEYED_PARTS=2, off by default): the tables of the first half of the keys here; those
  // of the second half on a third stream, started when these are done, i.e. beside the first half's ladder.
  // Measured: the tables' share of the step shrinks by 0.3 ms and the ladders grow by 0.36 (2^16 keys: 5.27-5.30
  // against 5.22 ms; 2^17 keys: 6.33 against 6.12-6.21) - the table kernels' traffic costs the ladder more
  // than their idle multipliers give it.
  rc = s2k_internal_key_tables(ctx, st, kg, 0, kg->nparts, ctx->ev_mid);
  if (rc) return rc;
  if (kg->nparts > 1) {
    HIP_TRY(ctx, hipEventRecord(ctx->ev_part0, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux2, ctx->ev_part0, 0));
    rc = s2k_internal_key_tables(ctx, ctx->s_aux2, kg, 1, kg->nparts, nullptr);
    if (rc) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_part1, ctx->s_aux2));
  }
  if (one_stream) {
    launch_prep(st);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_mid, 0));
  }
  if (n_first < n) {
    k_generator_part<<<blocks_for(n - n_first), 256, 0, ctx->s_aux>>>(n_first, (uint32_t)n, prep, ctx->gt_call, gp, stride);
    HIP_TRY(ctx, hipGetLastError());
  }
  return S2K_OK;
}
template <class PrepFn>
static int grouped_front(s2k_ctx* ctx, hipStream_t st, size_t n, const uint8_t* d_keys, int key_bytes, uint32_t* prep,
                         uint32_t* gp, size_t stride, PrepFn launch_prep, key_groups* kg, bool gp_in_prep = false) {
  int rc = ctx_aux_streams(ctx);
  if (rc) return rc;
  // the buffers the grouping needs are reserved BEFORE the fork: growing one is a device-wide synchronisation
  // (hipFree), which must not meet work of this call in flight on the second stream
  rc = s2k_internal_key_reserve(ctx, n, key_bytes);
  if (rc) return rc;
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_fork, 0));
  rc = grouped_front_forked(ctx, st, n, d_keys, key_bytes, prep, gp, stride, launch_prep, kg, gp_in_prep);
  // error or not, the caller's stream waits for the second one again, and the context's next call for this one: an
  // error return leaves nothing of this call in flight behind the context's back
  ctx_aux_join(ctx, st);
  if (kg->nparts > 1 && rc == S2K_OK) { /* (ev_part1 is waited for by the caller, before the second ladder) */ }
  if (rc) {
    if (ctx->s_aux2) {
      (void)hipEventRecord(ctx->ev_part1, ctx->s_aux2);
      (void)hipStreamWaitEvent(st, ctx->ev_part1, 0);
    }
    (void)ctx_leave(ctx, st);
  }
  return rc;
}

s2k_phase_locks& s2k_internal_phase(int device) {
  static s2k_phase_locks locks[64];
  return locks[device & 63];
}

extern "C" {

const char* s2k_version(void) { return "secp256k1_voi_amd 0.4 (gfx950)"; }
#define S2K_STR2(x) #x
#define S2K_STR(x) S2K_STR2(x)
#ifndef S2K_BUILD_FLAGS
#define S2K_BUILD_FLAGS ""
#endif
const char* s2k_build_config(void) {
  return "GT_BITS=" S2K_STR(S2K_GT_BITS) " GT_BITS_FIRST=" S2K_STR(S2K_GT_BITS_FIRST) " PREP_M=" S2K_STR(S2K_PREP_M)
         " QT_PACK=" S2K_STR(S2K_QT_PACK) " FAST_WAVES=" S2K_STR(S2K_FAST_WAVES) " STRIDE_PAD=" S2K_STR(S2K_STRIDE_PAD)
         " ROW_MAX=" S2K_STR(S2K_ROW_MAX_DEFAULT) " QUAD_MAX=" S2K_STR(S2K_QUAD_MAX_DEFAULT)
         " flags=[" S2K_BUILD_FLAGS "]";
}
const char* s2k_last_error(const s2k_ctx* ctx) { return ctx ? ctx->err : g_err; }
int s2k_device_count(void) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return count;
}
// "0000:05:00.0" of a device (hipDeviceGetPCIBusId): what topology.cpp looks up in sysfs
int s2k_device_pci_bus_id(int device, char* out, size_t len) {
  if (!out || len < 13) return S2K_ERR_ARG;
  out[0] = 0;
  if (hipDeviceGetPCIBusId(out, (int)len, device) != hipSuccess) {
    (void)hipGetLastError();
    out[0] = 0;
    return S2K_ERR_NO_DEVICE;
  }
  return S2K_OK;
}

// workspace (32-bit words per lane, lane stride = n rounded up to 64):
//   [0,256)    per-lane point table of the fast path: 8 entries x 8 quads (tb_*); the complete path
//              uses the first 192 words for its projective table
//   [256,304)  four scratch field elements ("fin": X, Y, Z for k_affine_finish; Z_7*C / prefix products)
//   then       17 words of scalar-prep output, 36 words for the generator part (grouped flow).  The prep kernel's own scratch (prefix products and
//              s in Montgomery form, 10 words each) borrows the start of the table region,
//              which is only written after the prep kernel has finished.
//   then       worklist: 1 counter + n indices
constexpr size_t WS_QT = 0, WS_FIN = TBL_WORDS, WS_PREP = TBL_WORDS + FIN_WORDS, WS_PREF = WS_QT, WS_SMONT = WS_QT + 10,
                 WS_GP = WS_PREP + PREP_WORDS, WS_LANE_WORDS = WS_GP + 3 * FQT_FE_WORDS;
static_assert(TBL_WORDS >= QT_WORDS, "the complete path's table must fit in the fast path's region");

size_t s2k_ecdsa_workspace_bytes(size_t n) {
  return (lane_stride(n) * WS_LANE_WORDS + 64 + lane_stride(n)) * sizeof(uint32_t);
}
size_t s2k_ctx_device_bytes(const s2k_ctx* ctx, size_t n) {
  if (!ctx) return 0;
  size_t total = s2k_ecdsa_workspace_bytes(n);
  {
    uint64_t info[4] = {0, 0, 0, 0};
    (void)s2k_ctx_gt_info(const_cast<s2k_ctx*>(ctx), info);
    total += (size_t)info[3] + (info[2] ? gt_bytes_of((int)info[1]) : 0);   // the tables held, and the one being built
  }
  if (ctx->kg_mode != S2K_KEYS_OFF && n >= KG_MIN_BATCH) total += s2k_internal_key_bytes(ctx, n);
  return total;
}

// (shared with ops.hip; not part of the ABI)
__attribute__((visibility("hidden"))) int s2k_internal_ensure_ws(s2k_ctx* ctx, size_t n) {
  size_t need = s2k_ecdsa_workspace_bytes(n);
  if (need <= ctx->ws_bytes) return S2K_OK;
  if (ctx->ws) {
    HIP_TRY(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    ctx->last_wl_count = nullptr;   // pointed into the old workspace (s2k_ctx_key_grouping_stats)
  }
  HIP_TRY(ctx, hipMalloc(&ctx->ws, need));
  ctx->ws_bytes = need;
  return S2K_OK;
}

// The resident generator tables are shared by the contexts of a device (one context per goroutine / thread is the intended
// use): per device one registry with its own lock, reference counted, the last context to go frees it.  A
// registry holds tables of SEVERAL widths:
//   * the first table (GT_BITS_FIRST = 20 bits, 0.8 GiB) is built inside the first s2k_ctx_create: < 0.1 s to a usable context;
//   * an automatic context then starts ONE background thread per device that allocates and builds the wide table
//     (GT_BITS_TARGET = 26 bits / 40 GiB when the device has twice that free and the budget allows, else 24 / 11 GiB, else 22 /
//     3 GiB, else nothing) on a stream of its own, beside whatever the contexts are doing; when it is done, the next CALL of
//     every automatic context uses it (ctx_enter loads the view once; a call never changes tables between its launches).  A
//     failed allocation or build leaves the contexts on the table they have, and is not tried again while the registry lives;
//   * s2k_ctx_create_ex with an explicit width uses exactly that table, built synchronously unless it exists or is being built.
// The builder thread is never detached: the last context to go cancels it (it looks at the flag before it allocates, between
// the windows of a table and before it publishes) and joins it; a process that exits with contexts alive does the same from an
// atexit handler, so that no thread of this library is inside the HIP runtime while that is torn down.
// The reference's analogue is a 510 KiB unpack at package init (point_mul_table.go:75-100).
namespace {
struct gtable_dev {
  std::mutex m;
  std::condition_variable cv;                     // a build has ended (either kind), the builder has been kicked or cancelled
  std::atomic<uint32_t*> table[GT_BITS_MAX + 1];  // by width; non-null = built and readable
  bool being_built[GT_BITS_MAX + 1] = {};         // a thread is building this width outside the lock (others wait for it)
  std::atomic<int> auto_bits{0};                  // width the automatic contexts use now (0: registry empty)
  int refs = 0;
  std::atomic<bool> kicked{false};                // an entry point has enqueued its first call on the device (ctx_leave)
  int target = 0;                                 // width the background build aims for (0: none wanted / none possible)
  bool building = false;                          // the builder thread is running
  bool gave_up = false;                           // a background build failed: not tried again by later contexts of this registry
  bool closing = false;                           // the last context is waiting for the builder to end: nobody acquires meanwhile
  std::atomic<bool> cancel{false};                // the builder is to stop at its next look
  std::atomic<size_t> pending{0};                 // bytes the builder is about to take from the device (s2k_internal_gt_pending_bytes)
  std::thread builder;
  char note[200] = {0};                           // why the target is what it is (s2k_ctx_gt_note; read and written under m)
  gtable_dev() {
    for (auto& t : table) t.store(nullptr);
  }
};
gtable_dev g_gtable[64];
std::atomic<size_t> g_gt_budget{0};               // s2k_set_generator_table_budget: bytes per device the tables may take (0: by free memory)
std::atomic<size_t> g_keyset_budget{0};           // s2k_set_table_memory_budgets: what counts as free for a key set's joint tables (0: what is free)

// allocate and build the table of `bits` on `stream` (null: the default stream); synchronises that stream.  With `cancel` the
// table is built window by window and the flag is looked at in between (hipErrorNotReady: cancelled, nothing is left allocated).
hipError_t gtable_build(int bits, hipStream_t stream, uint32_t** out, const std::atomic<bool>* cancel = nullptr,
                        std::atomic<size_t>* pending = nullptr) {
  const uint32_t windows = gt_windows_of(bits);
  uint32_t *table = nullptr, *bases = nullptr;
  hipError_t e = hipMalloc((void**)&table, gt_bytes_of(bits));
  if (pending) pending->store(0);                 // (taken, or not to be had)
  if (e == hipSuccess) e = hipMalloc((void**)&bases, (windows + 1) * 64);
  if (e == hipSuccess) {
    k_gen_gtable_bases<<<1, 1, 0, stream>>>(bases, (uint32_t)bits, windows);
    const unsigned per_window = (unsigned)(((size_t)1 << bits) / 256);
    if (!cancel) {
      k_gen_gtable<<<per_window * windows, 256, 0, stream>>>(table, bases, (uint32_t)bits, windows, 0u);
      e = hipGetLastError();
    } else {
      for (uint32_t w = 0; w < windows && e == hipSuccess; ++w) {
        if (cancel->load()) {
          e = hipErrorNotReady;
          break;
        }
        k_gen_gtable<<<per_window, 256, 0, stream>>>(table, bases, (uint32_t)bits, windows, w * per_window);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
      }
    }
  }
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  else (void)hipStreamSynchronize(stream);
  if (bases) (void)hipFree(bases);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    if (table) (void)hipFree(table);
    return e;
  }
  *out = table;
  return hipSuccess;
}

// does a wide table of `bits` fit NOW: twice its size free on the device, so that the tables never take more than half of
// what is left for key sets, workspaces and other processes; and inside the budget.  Asked when the target is chosen and again
// by the builder right before each allocation (half a second or more later).
bool gtable_fits(int bits, size_t* free_out) {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (free_out) *free_out = free_b;
  const size_t need = gt_bytes_of(bits), budget = g_gt_budget.load();
  if (budget && need + gt_bytes_of(GT_BITS_FIRST) > budget) return false;
  return free_b >= 2 * need;
}
// the widest of 26 / 24 / 22 (not above `from`, above the first table) that fits; 0: none.  Caller holds g.m (the note).
int gtable_pick_target(gtable_dev& g, int from) {
  size_t free_b = 0;
  const size_t budget = g_gt_budget.load();
  for (int bits : {26, 24, 22}) {
    if (bits > from || bits > GT_BITS_TARGET || bits <= GT_BITS_FIRST) continue;
    if (!gtable_fits(bits, &free_b)) continue;
    snprintf(g.note, sizeof g.note, "%d-bit windows: %.1f GiB of %.1f GiB free%s", bits, gt_bytes_of(bits) / 1073741824.0, free_b / 1073741824.0,
             budget ? " (inside the budget)" : "");
    return bits;
  }
  snprintf(g.note, sizeof g.note, "no wide table fits (%.1f GiB free, budget %.1f GiB): staying on %d bits", free_b / 1073741824.0,
           budget / 1073741824.0, GT_BITS_FIRST);
  return 0;
}

void gtable_builder_main(int device, int bits) {
  gtable_dev& g = g_gtable[device];
  // The allocation of tens of gigabytes holds the runtime's allocator for a second or two, and whatever another thread asks of
  // the runtime meanwhile (its first call's workspace ...) waits behind it: 0.1-0.3 s from s2k_ctx_create to the first verdict
  // became 2.3 s whenever the two collided.  So the build starts when the context's first call has been enqueued (ctx_leave),
  // or after half a second without one.
  {
    std::unique_lock<std::mutex> lock(g.m);
    g.cv.wait_for(lock, std::chrono::milliseconds(500), [&] { return g.kicked.load() || g.cancel.load(); });
  }
  hipError_t e = g.cancel.load() ? hipErrorNotReady : hipSetDevice(device);
  hipStream_t st = nullptr;
  if (e == hipSuccess) {
    int lo = 0, hi = 0;                            // lowest priority: the build yields to verification kernels
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = 0;
    e = hipStreamCreateWithPriority(&st, hipStreamNonBlocking, lo);
  }
  const int wanted = bits;
  int built = 0;
  uint32_t* table = nullptr;                       // ours to free unless it is published
  std::unique_lock<std::mutex> lock(g.m);
  while (e == hipSuccess && bits > GT_BITS_FIRST && !g.cancel.load()) {
    if (g.table[bits].load()) {                    // a context with that explicit width built it meanwhile: use that one
      built = bits;
      break;
    }
    if (g.being_built[bits]) {                     // ... or is building it right now: wait for that to end, look again
      g.cv.wait(lock, [&] { return !g.being_built[bits] || g.cancel.load(); });
      continue;
    }
    bits = gtable_pick_target(g, bits);            // (the memory may have gone since the target was chosen)
    if (!bits) break;
    if (g.table[bits].load() || g.being_built[bits]) continue;
    g.being_built[bits] = true;
    g.pending.store(gt_bytes_of(bits));
    lock.unlock();
    const hipError_t be = gtable_build(bits, st, &table, &g.cancel, &g.pending);
    lock.lock();
    g.pending.store(0);
    g.being_built[bits] = false;
    g.cv.notify_all();
    if (be == hipSuccess) {
      built = bits;
      break;
    }
    table = nullptr;
    if (be == hipErrorNotReady) break;             // cancelled
    bits -= 2;                                     // a failed allocation or build: the next width down that fits
  }
  if (g.cancel.load()) {                           // every context went away meanwhile: nobody wants the table
    if (table) (void)hipFree(table);
  } else if (built) {
    if (table && g.table[built].load()) {          // (never two tables of one width: the one that is there stays)
      (void)hipFree(table);
    } else if (table) {
      g.table[built].store(table, std::memory_order_release);
    }
    g.auto_bits.store(built, std::memory_order_release);
    if (built != wanted) snprintf(g.note, sizeof g.note, "%d-bit windows (the table of %d bits could not be had)", built, wanted);
    g.target = built;
  } else {
    if (e != hipSuccess || bits) snprintf(g.note, sizeof g.note, "the wide table could not be built: staying on %d bits", GT_BITS_FIRST);
    g.target = 0;
    g.gave_up = true;
  }
  g.building = false;
  g.cv.notify_all();
  lock.unlock();
  if (st) (void)hipStreamDestroy(st);
}

// a process that exits with contexts alive: the builders are stopped and joined before static destruction and before the HIP
// runtime's own exit handlers (registered earlier than this one, so run later)
void gtable_at_exit() {
  for (gtable_dev& g : g_gtable) {
    std::thread t;
    {
      std::lock_guard<std::mutex> lock(g.m);
      if (!g.builder.joinable()) continue;
      g.cancel.store(true);
      g.cv.notify_all();
      t = std::move(g.builder);
    }
    t.join();
  }
}
}  // namespace

// fixed_bits == 0: the shared automatic tables (first table now, wide table in the background); else exactly that width
static hipError_t gtable_acquire(int device, int fixed_bits) {
  if (device < 0 || device >= 64) return hipErrorInvalidDevice;
  static std::once_flag at_exit_once;
  std::call_once(at_exit_once, [] { (void)atexit(gtable_at_exit); });
  gtable_dev& g = g_gtable[device];
  std::unique_lock<std::mutex> lock(g.m);
  g.cv.wait(lock, [&] { return !g.closing; });    // (the last context of an earlier generation is still joining its builder)
  const int bits = fixed_bits ? fixed_bits : GT_BITS_FIRST;
  while (!g.table[bits].load()) {
    if (g.being_built[bits]) {                     // by the builder or by another context's creation: wait, do not build a second one
      g.cv.wait(lock, [&] { return !g.being_built[bits]; });
      continue;
    }
    g.being_built[bits] = true;
    lock.unlock();                                 // (seconds for a wide table: the contexts of this device keep working)
    uint32_t* t = nullptr;
    const hipError_t e = gtable_build(bits, nullptr, &t);
    lock.lock();
    g.being_built[bits] = false;
    g.cv.notify_all();
    if (e != hipSuccess) return e;
    g.table[bits].store(t, std::memory_order_release);
  }
  if (!fixed_bits) {
    if (g.auto_bits.load() == 0) g.auto_bits.store(GT_BITS_FIRST, std::memory_order_release);
    if (!g.building && !g.gave_up && g.target == 0 && g.auto_bits.load() == GT_BITS_FIRST && GT_BITS_TARGET > GT_BITS_FIRST) {
      g.target = gtable_pick_target(g, GT_BITS_TARGET);
      if (g.target) {
        if (g.table[g.target].load()) {             // (built earlier for a context with that explicit width)
          g.auto_bits.store(g.target, std::memory_order_release);
        } else {
          if (g.builder.joinable()) g.builder.join();   // (an earlier builder that has ended: building is false)
          g.building = true;
          g.cancel.store(false);
          g.pending.store(gt_bytes_of(g.target));
          g.builder = std::thread(gtable_builder_main, device, g.target);
        }
      }
    }
  }
  ++g.refs;
  return hipSuccess;
}
static void gtable_release(int device) {
  gtable_dev& g = g_gtable[device];
  std::unique_lock<std::mutex> lock(g.m);
  if (g.refs <= 0 || --g.refs > 0) return;
  // the last context: stop the builder (it frees what it has not published), wait for it, then free the tables
  g.closing = true;
  g.cancel.store(true);
  g.cv.notify_all();
  std::thread t = std::move(g.builder);
  lock.unlock();
  if (t.joinable()) t.join();
  lock.lock();
  (void)hipSetDevice(device);
  for (auto& tb : g.table) {
    uint32_t* p = tb.exchange(nullptr);
    if (p) (void)hipFree(p);
  }
  g.auto_bits.store(0);
  g.target = 0;
  g.gave_up = false;
  g.building = false;
  g.pending.store(0);
  g.kicked.store(false);
  g.cancel.store(false);
  g.closing = false;
  g.cv.notify_all();
}
extern "C++" __attribute__((visibility("hidden"))) void s2k_internal_gt_kick(int device) {
  gtable_dev& g = g_gtable[device];
  if (g.kicked.load(std::memory_order_relaxed)) return;         // (every call after the first: one relaxed load)
  g.kicked.store(true);
  std::lock_guard<std::mutex> lock(g.m);                        // (so that the notification cannot fall between the builder's test and its wait)
  g.cv.notify_all();
}
// the table a call that starts now uses (ctx_enter: once per call)
extern "C++" __attribute__((visibility("hidden"))) gt_view s2k_internal_gt_load(const s2k_ctx* ctx) {
  const gtable_dev& g = g_gtable[ctx->device];
  const int bits = ctx->gt_fixed ? ctx->gt_fixed : g.auto_bits.load(std::memory_order_acquire);
  gt_view v;
  v.p = g.table[bits].load(std::memory_order_acquire);
  v.bits = (uint32_t)bits;
  v.windows = gt_windows_of(bits);
  return v;
}
// bytes of device memory the background build of this device is about to allocate: whoever sizes something by what is free
// (key-set layouts, the per-key table buffer) takes them off first
extern "C++" __attribute__((visibility("hidden"))) size_t s2k_internal_gt_pending_bytes(int device) {
  return device >= 0 && device < 64 ? g_gtable[device].pending.load() : 0;
}
// Test hook behind s2k_debug_gt_swap_in_call: make `bits` the width of the device's automatic contexts NOW (building the table
// if the registry does not hold it) - what the background builder does at a moment nobody chooses.
static hipError_t gtable_debug_publish(int device, int bits) {
  gtable_dev& g = g_gtable[device];
  std::unique_lock<std::mutex> lock(g.m);
  while (!g.table[bits].load()) {
    if (g.being_built[bits]) {
      g.cv.wait(lock, [&] { return !g.being_built[bits]; });
      continue;
    }
    g.being_built[bits] = true;
    lock.unlock();
    uint32_t* t = nullptr;
    hipStream_t st = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e == hipSuccess) e = gtable_build(bits, st, &t);
    if (st) (void)hipStreamDestroy(st);
    lock.lock();
    g.being_built[bits] = false;
    g.cv.notify_all();
    if (e != hipSuccess) return e;
    g.table[bits].store(t, std::memory_order_release);
  }
  g.auto_bits.store(bits, std::memory_order_release);
  return hipSuccess;
}

extern "C" {
// Bytes per device the generator tables of this process may take (0 = no limit but the device's free memory).  Applies to
// automatic contexts created after the call; the first table (0.8 GiB) is always built.
void s2k_set_generator_table_budget(size_t bytes) { g_gt_budget.store(bytes); }
// Caps on the two other table kinds, per process (0 = none): what s2k_keyset_create_ex treats as free memory when it chooses or
// checks a joint-table layout, and the largest per-key table buffer a verification call may allocate (larger requests are
// treated as failed allocations: the table cap halves, below 1024 tables the batch is verified without tables).  These replace
// the S2K_TEST_* environment variables of round 4 (no ambient test hooks in allocation paths).
void s2k_set_table_memory_budgets(size_t keyset_free_bytes, size_t key_table_bytes) {
  g_keyset_budget.store(keyset_free_bytes);
  s2k_internal_key_table_limit().store(key_table_bytes);
}
// info[0] = window bits the context's next call uses, [1] = bits the background build aims for (0: none), [2] = 1 while it
// is running, [3] = bytes of generator tables the device holds for this process
int s2k_ctx_gt_info(s2k_ctx* ctx, uint64_t info[4]) {
  if (!ctx || !info) return fail(ctx, S2K_ERR_ARG, "null argument");
  gtable_dev& g = g_gtable[ctx->device];
  std::lock_guard<std::mutex> lock(g.m);
  info[0] = (uint64_t)(ctx->gt_fixed ? ctx->gt_fixed : g.auto_bits.load());
  info[1] = (uint64_t)(ctx->gt_fixed ? 0 : g.target);
  info[2] = g.building && !ctx->gt_fixed ? 1 : 0;
  uint64_t bytes = 0;
  for (int b = GT_BITS_MIN; b <= GT_BITS_MAX; ++b)
    if (g.table[b].load()) bytes += gt_bytes_of(b);
  info[3] = bytes;
  return S2K_OK;
}
// (a copy taken under the lock, owned by the context: the builder thread rewrites the registry's note)
const char* s2k_ctx_gt_note(s2k_ctx* ctx) {
  if (!ctx) return "";
  gtable_dev& g = g_gtable[ctx->device];
  std::lock_guard<std::mutex> lock(g.m);
  snprintf(ctx->gt_note, sizeof ctx->gt_note, "%s", g.note);
  return ctx->gt_note;
}
// blocks until the background build of the context's device has ended (either way); returns the window bits in use then
int s2k_ctx_gt_wait(s2k_ctx* ctx) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  gtable_dev& g = g_gtable[ctx->device];
  std::unique_lock<std::mutex> lock(g.m);
  g.cv.wait(lock, [&] { return !g.building; });
  return ctx->gt_fixed ? ctx->gt_fixed : g.auto_bits.load();
}
// Test hook (tests/test_gpu_round6.py): the next s2k_ecdsa_verify_batch_device call of this (automatic) context publishes the
// table of `bits` for the device's automatic contexts BETWEEN its ladder launch and its worklist launch - the moment the
// background build must not be visible inside a call.  One shot; 0 disarms.
int s2k_debug_gt_swap_in_call(s2k_ctx* ctx, int bits) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (bits != 0 && (bits < 16 || bits > GT_BITS_MAX)) return fail(ctx, S2K_ERR_ARG, "generator window width %d: 0 or 16 .. %d", bits, GT_BITS_MAX);
  if (ctx->gt_fixed && bits) return fail(ctx, S2K_ERR_ARG, "a context with an explicit table width never changes tables");
  ctx->dbg_gt_swap = bits;
  return S2K_OK;
}
}  // extern "C"

int s2k_ctx_create(int device_index, s2k_ctx** out) { return s2k_ctx_create_ex(device_index, 0, 0, out); }

// gt_bits: 0 = automatic (above); 16 .. 26 = exactly that window width for this context's generator tables, built before the
// call returns.  flags: S2K_CTX_WAIT_TABLES = return only when the background build has ended (a context that must run at
// full speed from its first call: benchmarks).
int s2k_ctx_create_ex(int device_index, int gt_bits, uint32_t flags, s2k_ctx** out) {
  if (!out) return fail(nullptr, S2K_ERR_ARG, "s2k_ctx_create: out is NULL");
  *out = nullptr;
  if (gt_bits != 0 && (gt_bits < 16 || gt_bits > GT_BITS_MAX)) return fail(nullptr, S2K_ERR_ARG, "generator window width %d: 0 (automatic) or 16 .. %d", gt_bits, GT_BITS_MAX);
  if (flags & ~(uint32_t)S2K_CTX_WAIT_TABLES) return fail(nullptr, S2K_ERR_ARG, "unknown context flags");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
    return fail(nullptr, S2K_ERR_NO_DEVICE, "no HIP device visible (this engine has no CPU fallback)");
  if (device_index < 0 || device_index >= count)
    return fail(nullptr, S2K_ERR_ARG, "device index %d out of range (0..%d)", device_index, count - 1);
  s2k_ctx* ctx = new (std::nothrow) s2k_ctx();
  if (!ctx) return fail(nullptr, S2K_ERR_NOMEM, "out of host memory");
  ctx->device = device_index;
  {
    static std::atomic<uint64_t> next_generation{1};
    ctx->generation = next_generation.fetch_add(1);
  }
  if (getrandom(&ctx->kg_seed, sizeof ctx->kg_seed, 0) != (ssize_t)sizeof ctx->kg_seed) ctx->kg_seed = 0x5ec9u;   // hash seed of the key grouping
  if (const char* v = getenv("S2K_KEYED_PARTS")) {        // measurement knob: 2 = the two-part flow (grouped_front)
    int np = atoi(v);
    if (np == 1 || np == 2) ctx->kg_parts = (uint32_t)np;
  }
  if (const char* v = getenv("S2K_GP_FIRST_PERCENT")) {   // measurement knob (tools/keyed_probe.py)
    int pc = atoi(v);
    if (pc >= 0 && pc <= 100) ctx->gp_first_percent = (uint32_t)pc;
  }
  hipError_t e = hipSetDevice(device_index);
  hipDeviceProp_t prop;
  if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device_index);
  if (e == hipSuccess) ctx->cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (e == hipSuccess) e = hipEventCreateWithFlags(&ctx->ev_done, hipEventDisableTiming);
  if (e == hipSuccess) e = hipMalloc((void**)&ctx->clk, 64);
  if (e == hipSuccess) e = hipMemset(ctx->clk, 0, 64);
  if (e == hipSuccess) {
    e = gtable_acquire(device_index, gt_bits);
    if (e == hipSuccess) {
      ctx->gt_held = true;
      ctx->gt_fixed = gt_bits;
    } else if (e == hipErrorOutOfMemory) {
      (void)fail(nullptr, S2K_ERR_NOMEM, "context creation failed: no device memory for the %d-bit generator tables (%.1f GiB)",
                 gt_bits ? gt_bits : GT_BITS_FIRST, gt_bytes_of(gt_bits ? gt_bits : GT_BITS_FIRST) / 1073741824.0);
    }
  }
  if (e != hipSuccess) {
    int rc = e == hipErrorOutOfMemory ? S2K_ERR_NOMEM : fail(nullptr, S2K_ERR_HIP, "context creation failed: %s", hipGetErrorString(e));
    if (ctx->clk) (void)hipFree(ctx->clk);
    if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
    delete ctx;
    return rc;
  }
  if (flags & S2K_CTX_WAIT_TABLES) (void)s2k_ctx_gt_wait(ctx);
  *out = ctx;
  return S2K_OK;
}

void s2k_ctx_destroy(s2k_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (s2k_ctx::pipe_slot& sl : ctx->pipe) {          // batches still in flight are waited for (their verdicts are delivered)
    if (sl.ctx) {
      (void)s2k_internal_pipe_retire(ctx, sl);
      s2k_ctx_destroy(sl.ctx);
      sl.ctx = nullptr;
    }
    if (sl.h_valid) (void)hipHostFree(sl.h_valid);
    sl.h_valid = nullptr;
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl.done = nullptr;
    for (hipEvent_t* e : {&sl.t_begin, &sl.t_copied, &sl.t_end}) {
      if (*e) (void)hipEventDestroy(*e);
      *e = nullptr;
    }
  }
  if (ctx->kga_note) {                                // (a kernel still in flight may hold the address)
    (void)hipDeviceSynchronize();
    (void)hipHostFree(ctx->kga_note);
  }
  if (ctx->gt_held) gtable_release(ctx->device);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->msm_ws) (void)hipFree(ctx->msm_ws);
  if (ctx->rlc_save) (void)hipFree(ctx->rlc_save);
  if (ctx->io) (void)hipFree(ctx->io);
  if (ctx->h_small) (void)hipHostFree(ctx->h_small);
  if (ctx->clk) (void)hipFree(ctx->clk);
  if (ctx->ev_done) (void)hipEventDestroy(ctx->ev_done);
  if (ctx->s_aux && !ctx->streams_shared) (void)hipStreamDestroy(ctx->s_aux);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_mid) (void)hipEventDestroy(ctx->ev_mid);
  for (hipEvent_t& e : ctx->ev_arrival)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev_part0) (void)hipEventDestroy(ctx->ev_part0);
  if (ctx->ev_part1) (void)hipEventDestroy(ctx->ev_part1);
  if (ctx->s_aux2) (void)hipStreamDestroy(ctx->s_aux2);
  if (ctx->s_msm_tail) (void)hipStreamDestroy(ctx->s_msm_tail);
  if (ctx->kg) (void)hipFree(ctx->kg);
  if (ctx->ktab) (void)hipFree(ctx->ktab);
  if (ctx->xkeys) (void)hipFree(ctx->xkeys);
  for (size_t i = 0; i < ctx->prof_cap; ++i) (void)hipEventDestroy(ctx->prof_ev[i]);
  for (size_t i = 0; i < ctx->msm_prof_cap; ++i) (void)hipEventDestroy(ctx->msm_prof_ev[i]);
  delete[] ctx->msm_prof_ev;
  delete[] ctx->prof_ev;
  for (hipEvent_t e : ctx->ev_copied)
    if (e) (void)hipEventDestroy(e);
  if (ctx->lane1_comp) (void)hipStreamDestroy(ctx->lane1_comp);
  if (ctx->lane1_aux) (void)hipStreamDestroy(ctx->lane1_aux);
  if (ctx->s_copy && !ctx->streams_shared) (void)hipStreamDestroy(ctx->s_copy);
  if (ctx->s_comp && !ctx->streams_shared) (void)hipStreamDestroy(ctx->s_comp);
  delete ctx;
}

int s2k_ctx_profile(s2k_ctx* ctx, int enable) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (enable && !ctx->prof_ev) {
    const size_t cap = PROF_EV * 1024;   // 1024 verification calls between two reads
    ctx->prof_ev = new (std::nothrow) hipEvent_t[cap];
    if (!ctx->prof_ev) return fail(ctx, S2K_ERR_NOMEM, "out of host memory");
    for (size_t i = 0; i < cap; ++i) {
      hipError_t e = hipEventCreate(&ctx->prof_ev[i]);
      if (e != hipSuccess) {
        for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(ctx->prof_ev[j]);
        delete[] ctx->prof_ev;
        ctx->prof_ev = nullptr;
        return fail(ctx, S2K_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
      }
    }
    ctx->prof_cap = cap;
  }
  ctx->prof_on = enable != 0;
  ctx->prof_used = 0;
  return S2K_OK;
}

int s2k_ctx_profile_read_stages(s2k_ctx* ctx, double* ms_sum5, double* ms_fast_each, size_t cap, size_t* calls,
                                double* shader_mhz) {
  if (!ctx || !ms_sum5 || !calls) return fail(ctx, S2K_ERR_ARG, "null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  for (int j = 0; j < PROF_EV - 1; ++j) ms_sum5[j] = 0.0;
  const size_t k = ctx->prof_used / PROF_EV;
  for (size_t i = 0; i < k; ++i) {
    for (int j = 0; j < PROF_EV - 1; ++j) {
      float ms = 0.f;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->prof_ev[PROF_EV * i + j], ctx->prof_ev[PROF_EV * i + j + 1]));
      ms_sum5[j] += ms;
      if (j == 2 && ms_fast_each && i < cap) ms_fast_each[i] = ms;
    }
  }
  *calls = k;
  ctx->prof_used = 0;
  if (shader_mhz) {   // [0]: first wave of the last ladder launch, [1]: a wave of its final round
    uint64_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int wall_khz = 0;
    HIP_TRY(ctx, hipMemcpy(h, ctx->clk, sizeof h, hipMemcpyDeviceToHost));
    HIP_TRY(ctx, hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, ctx->device));
    for (int j = 0; j < 2; ++j) {
      const uint64_t* c = h + 4 * j;
      shader_mhz[j] = (c[3] > c[2] && wall_khz > 0) ? (double)(c[1] - c[0]) / (double)(c[3] - c[2]) * (wall_khz * 1e-3) : 0.0;
    }
  }
  return S2K_OK;
}

// the three-stage view: preparation (with grouping and tables) | ladder | everything after it
int s2k_ctx_profile_read(s2k_ctx* ctx, double* ms_sum3, double* ms_fast_each, size_t cap, size_t* calls,
                         double* shader_mhz) {
  if (!ctx || !ms_sum3 || !calls) return fail(ctx, S2K_ERR_ARG, "null argument");
  double s5[5];
  int rc = s2k_ctx_profile_read_stages(ctx, s5, ms_fast_each, cap, calls, shader_mhz);
  if (rc) return rc;
  ms_sum3[0] = s5[0] + s5[1];
  ms_sum3[1] = s5[2];
  ms_sum3[2] = s5[3] + s5[4];
  return S2K_OK;
}

// `arrivals` (optional): d_dig / d_r / d_s become valid piece by piece - signatures [lo[c], lo[c] + cnt[c]) after event
// ev[c]; the keys (d_pub) are valid in stream order.  The host-buffer entry point copies the keys first and lets the
// grouping and the per-key tables - which need nothing else - start while the rest of the batch is still crossing PCIe;
// the scalar preparation and the generator part of a piece follow its arrival on the second stream.
struct sig_arrivals {
  int count;
  size_t lo[8], cnt[8];
  hipEvent_t ev[8];
};
static int verify_batch_device(s2k_ctx* ctx, size_t n, const void* d_pub, const void* d_dig, const void* d_r, const void* d_s,
                               uint32_t flags, void* d_valid, void* hip_stream, const sig_arrivals* arrivals);
int s2k_ecdsa_verify_batch_device(s2k_ctx* ctx, size_t n, const void* d_pub, const void* d_dig, const void* d_r,
                                  const void* d_s, uint32_t flags, void* d_valid, void* hip_stream) {
  return verify_batch_device(ctx, n, d_pub, d_dig, d_r, d_s, flags, d_valid, hip_stream, nullptr);
}
static int verify_batch_device(s2k_ctx* ctx, size_t n, const void* d_pub, const void* d_dig, const void* d_r, const void* d_s,
                               uint32_t flags, void* d_valid, void* hip_stream, const sig_arrivals* arrivals) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!d_pub || !d_dig || !d_r || !d_s || !d_valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  ctx->kg_counters = nullptr;
  ctx->last_wl_count = nullptr;
  auto wait_all = [&](hipStream_t on) {
    if (arrivals)
      for (int c = 0; c < arrivals->count; ++c) (void)hipStreamWaitEvent(on, arrivals->ev[c], 0);
  };
  if (flags & S2K_ECDSA_FORCE_COMPLETE) {
    wait_all(st);
    k_ecdsa_verify<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, (const uint8_t*)d_pub, (const uint8_t*)d_dig,
                                                  (const uint8_t*)d_r, (const uint8_t*)d_s, flags, (uint8_t*)d_valid,
                                                  ctx->gt_call, qt, stride);
    HIP_TRY(ctx, hipGetLastError());
    return ctx_leave(ctx, st);
  }
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* gp = ws + WS_GP * stride;
  uint32_t* pref = ws + WS_PREF * stride;
  uint32_t* smont = ws + WS_SMONT * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  const uint32_t T = prep_lanes(n);
  const uint32_t kvf = (flags & S2K_ECDSA_FORCE_WORKLIST) ? KVF_FORCE_WORKLIST : 0u;
  uint64_t* clk = ctx->prof_on ? ctx->clk : nullptr;
  bool grouped = ctx->kg_mode != S2K_KEYS_OFF && n >= KG_MIN_BATCH;
  // small batches: a wave per signature beats any table (k_verify_row); a grouping mode somebody asked for by name
  // (S2K_KEYS_AUTO / S2K_KEYS_ALWAYS) is obeyed instead
  const bool row = n <= ctx->row_max && !kvf && (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF);
  // ... and up to quad_max four lanes per signature (k_verify_quad): half the latency of the lane kernels for calls that fill
  // neither
  const bool quad = !row && n <= ctx->quad_max && !kvf && (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF);
  if (row || quad) grouped = false;
  else HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));      // (these ladders have no worklist: one launch less)
  ctx->kg_note_dst = nullptr;
  if (grouped && ctx->kg_mode == S2K_KEYS_ADAPTIVE && n >= KG_ADAPT_MIN_BATCH) grouped = kg_adaptive_decide(ctx);
  if (grouped) {
    // the grouping arrays and the table buffer, before anything of this call is in flight; a device without room for the
    // tables verifies without them
    rc = s2k_internal_key_reserve(ctx, n, 64);
    if (rc == S2K_ERR_NOMEM) grouped = false;
    else if (rc) return rc;
  }
  ctx->last_wl_count = (row || quad) ? nullptr : wl_count;
  prof_mark(ctx, st, 0);
  if (grouped) {
    // Signatures of keys that occur often enough: per-key tables (keyed.hip) and the short ladder; the
    // rest: the general kernel over the list `left`.
    key_groups kg;
    rc = grouped_front(ctx, st, n, (const uint8_t*)d_pub, 64, prep, gp, stride,
                       [&](hipStream_t aux) {
                         if (!arrivals) {
                           k_scalar_prep<<<(T + 63) / 64, 64, 0, aux>>>((uint32_t)n, T, (const uint8_t*)d_dig, (const uint8_t*)d_r,
                                                                        (const uint8_t*)d_s, nullptr, flags, prep, pref, smont, stride);
                           return;
                         }
                         // piece by piece: the planes of the workspace are linear in the signature index, so a piece is
                         // the same kernels on shifted pointers
                         for (int c = 0; c < arrivals->count; ++c) {
                           const size_t lo = arrivals->lo[c], cnt = arrivals->cnt[c];
                           const uint32_t Tc = prep_lanes(cnt);
                           (void)hipStreamWaitEvent(aux, arrivals->ev[c], 0);
                           k_scalar_prep<<<(Tc + 63) / 64, 64, 0, aux>>>((uint32_t)cnt, Tc, (const uint8_t*)d_dig + lo * 32,
                                                                         (const uint8_t*)d_r + lo * 32, (const uint8_t*)d_s + lo * 32,
                                                                         nullptr, flags, prep + lo, pref + lo, smont + lo, stride);
                           k_generator_part<<<blocks_for(cnt), 256, 0, aux>>>((uint32_t)lo, (uint32_t)(lo + cnt), prep, ctx->gt_call, gp, stride);
                         }
                       },
                       &kg, /*gp_in_prep=*/arrivals != nullptr);
    if (rc) return rc;
    prof_mark(ctx, st, 2);
    k_verify_fast<MODE_ECDSA_KEYED><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, (const uint8_t*)d_pub, (const uint8_t*)d_r, prep,
                                                                   qt, fin, ctx->gt_call, (uint8_t*)d_valid, wl_count, wl,
                                                                   stride, nullptr, clk, kg);
    HIP_TRY(ctx, hipGetLastError());
    if (kg.nparts > 1) {   // the other side of the split, once its tables (third stream) are there
      HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_part1, 0));
      kg.part = 1;
      k_verify_fast<MODE_ECDSA_KEYED><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, (const uint8_t*)d_pub, (const uint8_t*)d_r, prep,
                                                                     qt, fin, ctx->gt_call, (uint8_t*)d_valid, wl_count, wl,
                                                                     stride, nullptr, nullptr, kg);
      HIP_TRY(ctx, hipGetLastError());
    }
    prof_mark(ctx, st, 3);
    k_verify_fast<MODE_ECDSA_LEFT><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, (const uint8_t*)d_pub, (const uint8_t*)d_r, prep,
                                                                  qt, fin, ctx->gt_call, (uint8_t*)d_valid, wl_count, wl,
                                                                  stride, nullptr, nullptr, kg);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 4);
  } else if (quad) {
    // four lanes per signature on the complete formulas (k_verify_quad), behind a preparation with one lane per signature
    wait_all(st);
    k_scalar_prep<<<(unsigned)((n + 63) / 64), 64, 0, st>>>((uint32_t)n, (uint32_t)n, (const uint8_t*)d_dig, (const uint8_t*)d_r,
                                                            (const uint8_t*)d_s, nullptr, flags, prep, pref, smont, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_verify_quad<<<(unsigned)((n + 63) / 64), 256, 0, st>>>((uint32_t)n, (const uint8_t*)d_pub, (const uint8_t*)d_r, prep,
                                                             ctx->gt_call, (uint8_t*)d_valid, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  } else if (row) {
    // small batches: a wave per signature on the complete formulas (k_verify_row); nothing is left for the worklist kernel
    wait_all(st);
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_verify_row<<<(unsigned)((n + 3) / 4), 320, 0, st>>>((uint32_t)n, (const uint8_t*)d_pub, (const uint8_t*)d_dig, (const uint8_t*)d_r,
                                                          (const uint8_t*)d_s, flags, ctx->gt_call, (uint8_t*)d_valid);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  } else {
    wait_all(st);
    k_scalar_prep<<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, (const uint8_t*)d_dig, (const uint8_t*)d_r,
                                                (const uint8_t*)d_s, nullptr, flags, prep, pref, smont, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_verify_fast<MODE_ECDSA><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, (const uint8_t*)d_pub, (const uint8_t*)d_r, prep, qt, fin,
                                                             ctx->gt_call, (uint8_t*)d_valid, wl_count, wl, stride, nullptr, clk,
                                                             key_groups{});
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
  }
  if (ctx->dbg_gt_swap) {   // test hook: the wide table appears HERE, between the ladder and the kernel that finishes its tagged lanes
    const int bits = ctx->dbg_gt_swap;
    ctx->dbg_gt_swap = 0;
    HIP_TRY(ctx, gtable_debug_publish(ctx->device, bits));
#ifdef S2K_DEBUG_GT_PER_LAUNCH   // (variant build that restores the fault of round 5 - a view per launch - to show the test sees it)
    ctx->gt_call = s2k_internal_gt_load(ctx);
#endif
  }
  k_verify_fallback<<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, (const uint8_t*)d_pub, (const uint8_t*)d_dig,
                                        (const uint8_t*)d_r, (const uint8_t*)d_s, flags, (uint8_t*)d_valid,
                                        ctx->gt_call, qt, stride);
  HIP_TRY(ctx, hipGetLastError());
  prof_mark(ctx, st, 5);
  return ctx_leave(ctx, st);
}

// ---------------------------------------------------------------------------------------
// Key sets: the per-key tables of a fixed list of public keys, built once and kept (the device analogue of what a
// secec.PublicKey caches per key, secec/secec.go:80-85,188-216: the decoded point and its encoding).  A verification
// call names each signature's key by index; nothing is grouped and no table is built inside the call.
// ---------------------------------------------------------------------------------------
struct s2k_keyset {
  s2k_ctx* ctx;       // the owner (compared, never followed after creation: the set may outlive it by accident)
  uint64_t generation;   // of the owner: a context destroyed and another created at the same address is not the owner
  int device;
  size_t n;
  uint8_t* base;      // device: keys | tables | validity | identity | counters (s2k_internal_keyset_bytes)
  size_t bytes;
  uint4* joint;       // device: the joint tables (320 KiB per key; 0.81 / 2.75 MiB at 5- / 6-bit digits), or null: the ladder over the 32-chunk tables
  size_t joint_bytes;
  int jw;             // digit width of the joint tables: 4, 5 or 6 (0: none)
};

int s2k_keyset_create(s2k_ctx* ctx, size_t n_keys, const uint8_t* pub_xy, s2k_keyset** out) {
  return s2k_keyset_create_ex(ctx, n_keys, pub_xy, S2K_KEYSET_AUTO, out);
}
int s2k_keyset_create_ex(s2k_ctx* ctx, size_t n_keys, const uint8_t* pub_xy, int layout, s2k_keyset** out) {
  if (!ctx || !out) return fail(ctx, S2K_ERR_ARG, "null argument");
  if (layout != S2K_KEYSET_AUTO && layout != S2K_KEYSET_CHUNKS && layout != S2K_KEYSET_JOINT && layout != S2K_KEYSET_JOINT5 &&
      layout != S2K_KEYSET_JOINT6)
    return fail(ctx, S2K_ERR_ARG, "unknown key-set layout");
  *out = nullptr;
  if (n_keys == 0 || !pub_xy) return fail(ctx, S2K_ERR_ARG, "empty key set");
  if (n_keys > 0x0fffffffu) return fail(ctx, S2K_ERR_ARG, "key set too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  s2k_keyset* ks = new (std::nothrow) s2k_keyset();
  if (!ks) return fail(ctx, S2K_ERR_NOMEM, "out of host memory");
  ks->ctx = ctx;
  ks->generation = ctx->generation;
  ks->device = ctx->device;
  ks->n = n_keys;
  size_t off[5];
  ks->bytes = s2k_internal_keyset_bytes(n_keys, off);
  hipError_t e = hipMalloc((void**)&ks->base, ks->bytes);
  if (e != hipSuccess) {
    delete ks;
    return fail(ctx, S2K_ERR_HIP, "key set of %zu keys (%zu bytes): %s", n_keys, ks->bytes, hipGetErrorString(e));
  }
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc == S2K_OK && hipMemcpyAsync(ks->base + off[0], pub_xy, n_keys * 64, hipMemcpyHostToDevice, st) != hipSuccess)
    rc = fail(ctx, S2K_ERR_HIP, "copy of the keys failed");
  if (rc == S2K_OK) rc = s2k_internal_keyset_build(ctx, ks->base, n_keys, st);
  // joint tables (one table addition per digit position instead of two; 320 KiB per key on top, 0.81 MiB at 5-bit digits: 26
  // positions, 2.75 MiB at 6-bit digits: 22): the layout asked for, or - S2K_KEYSET_AUTO - the widest of 5 and 4 bits that takes no
  // more than half of the device memory that is free now (2^16 keys at 5 bits are 56 GB: a quarter would ask for a device with
  // 224 GB free, which the 288 GB one never has once a context lives on it)
  ks->joint = nullptr;
  ks->joint_bytes = 0;
  ks->jw = 0;
  uint4* scratch = nullptr;
  if (rc == S2K_OK && layout != S2K_KEYSET_CHUNKS) {
    int w = layout == S2K_KEYSET_JOINT6 ? 6 : layout == S2K_KEYSET_JOINT5 ? 5 : layout == S2K_KEYSET_JOINT ? 4 : 0;
    // s2k_set_table_memory_budgets: no more than this many bytes count as free for joint tables (an operator's cap - and how
    // the choice of S2K_KEYSET_AUTO and the failure of an explicit layout are exercised without filling a 288 GB device)
    const size_t pretend_free = g_keyset_budget.load() ? g_keyset_budget.load() : ~(size_t)0;
    if (w == 0) {
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const size_t promised = s2k_internal_gt_pending_bytes(ctx->device);   // what the background table build is about to take
        free_b -= promised < free_b ? promised : free_b;
        if (free_b > pretend_free) free_b = pretend_free;
        if (s2k_internal_keyset_joint_bytes(n_keys, 5) + s2k_internal_keyset_joint_scratch_bytes(n_keys, 5) <= free_b / 2) w = 5;
        else if (s2k_internal_keyset_joint_bytes(n_keys, 4) <= free_b / 2) w = 4;
      }
    }
    if (w) {
      const size_t want = s2k_internal_keyset_joint_bytes(n_keys, w), want_scr = s2k_internal_keyset_joint_scratch_bytes(n_keys, w);
      hipError_t e2 = want + want_scr > pretend_free ? hipErrorOutOfMemory : hipMalloc((void**)&ks->joint, want);
      if (e2 == hipSuccess && want_scr) e2 = hipMalloc((void**)&scratch, want_scr);
      if (e2 == hipSuccess) {
        ks->joint_bytes = want;
        ks->jw = w;
        rc = w == 4 ? s2k_internal_keyset_build_joint(ctx, ks->base, n_keys, ks->joint, st)
                    : s2k_internal_keyset_build_joint_wide(ctx, ks->base, n_keys, w, ks->joint, scratch, st);
      } else {
        (void)hipGetLastError();
        if (ks->joint) (void)hipFree(ks->joint);
        ks->joint = nullptr;
        if (layout != S2K_KEYSET_AUTO) rc = fail(ctx, S2K_ERR_HIP, "joint tables of %zu keys (%zu bytes): %s", n_keys, want + want_scr, hipGetErrorString(e2));
      }
    }
  }
  if (rc == S2K_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ctx, S2K_ERR_HIP, "key set build failed");
  if (scratch) {
    (void)hipStreamSynchronize(st);
    (void)hipFree(scratch);
  }
  ctx->have_last = false;
  if (rc) {
    if (ks->joint) (void)hipFree(ks->joint);
    (void)hipFree(ks->base);
    delete ks;
    return rc;
  }
  *out = ks;
  return S2K_OK;
}
void s2k_keyset_destroy(s2k_keyset* ks) {
  if (!ks) return;
  (void)hipSetDevice(ks->device);
  (void)hipDeviceSynchronize();
  if (ks->joint) (void)hipFree(ks->joint);
  (void)hipFree(ks->base);
  delete ks;
}
size_t s2k_keyset_size(const s2k_keyset* ks) { return ks ? ks->n : 0; }
size_t s2k_keyset_device_bytes(const s2k_keyset* ks) { return ks ? ks->bytes + ks->joint_bytes : 0; }
int s2k_keyset_layout(const s2k_keyset* ks) {
  return !ks ? 0 : !ks->joint ? S2K_KEYSET_CHUNKS : ks->jw == 6 ? S2K_KEYSET_JOINT6 : ks->jw == 5 ? S2K_KEYSET_JOINT5 : S2K_KEYSET_JOINT;
}
int s2k_keyset_valid_keys(s2k_keyset* ks, uint8_t* valid) {
  if (!ks || !valid) return fail(nullptr, S2K_ERR_ARG, "null argument");
  size_t off[5];
  (void)s2k_internal_keyset_bytes(ks->n, off);
  HIP_TRY(nullptr, hipSetDevice(ks->device));
  HIP_TRY(nullptr, hipMemcpy(valid, ks->base + off[2], ks->n, hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_ecdsa_verify_batch_keyset_device(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const void* d_key_index, const void* d_dig,
                                         const void* d_r, const void* d_s, uint32_t flags, void* d_valid, void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  // (a child context of submit / wait verifies over its parent's key sets: the tables are only read)
  const s2k_ctx* owner = ctx->parent ? ctx->parent : ctx;
  if (!ks || ks->ctx != owner || ks->generation != owner->generation || ks->device != ctx->device)
    return fail(ctx, S2K_ERR_ARG, "key set of another context");
  if (n == 0) return S2K_OK;
  if (!d_key_index || !d_dig || !d_r || !d_s || !d_valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  if (flags & S2K_ECDSA_FORCE_COMPLETE) return fail(ctx, S2K_ERR_ARG, "S2K_ECDSA_FORCE_COMPLETE does not apply to key sets");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  rc = ctx_aux_streams(ctx);
  if (rc) return rc;
  rc = s2k_internal_keyset_reserve(ctx, ks->n, n);      // before the fork: growing a buffer synchronises the device
  if (rc) return rc;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* gp = ws + WS_GP * stride;
  uint32_t* pref = ws + WS_PREF * stride;
  uint32_t* smont = ws + WS_SMONT * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  size_t off[5];
  (void)s2k_internal_keyset_bytes(ks->n, off);
  ctx->last_wl_count = wl_count;
  HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));
  HIP_TRY(ctx, hipMemsetAsync(d_valid, 0, n, st));       // signatures naming no key of the set stay invalid
  const uint32_t T = prep_lanes(n);
  const uint32_t kvf = (flags & S2K_ECDSA_FORCE_WORKLIST) ? KVF_FORCE_WORKLIST : 0u;
  // second stream: scalar preparation and generator part; caller's: the sort by key index
  // (stage times, s2k_ctx_profile_read_stages: [0] + [1] the sort with the second stream's work beside it, [2] the ladder)
  prof_mark(ctx, st, 0);
  prof_mark(ctx, st, 1);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_fork, 0));
  k_scalar_prep<<<(T + 63) / 64, 64, 0, ctx->s_aux>>>((uint32_t)n, T, (const uint8_t*)d_dig, (const uint8_t*)d_r, (const uint8_t*)d_s,
                                                      nullptr, flags, prep, pref, smont, stride);
  k_generator_part<<<blocks_for(n), 256, 0, ctx->s_aux>>>(0u, (uint32_t)n, prep, ctx->gt_call, gp, stride);
  rc = hipGetLastError() == hipSuccess ? S2K_OK : fail(ctx, S2K_ERR_HIP, "launch failed");
  key_groups kg{};
  if (rc == S2K_OK) rc = s2k_internal_keyset_sort(ctx, ks->base, ks->n, n, (const uint32_t*)d_key_index, st, &kg);
  ctx_aux_join(ctx, st);
  if (rc) {
    (void)ctx_leave(ctx, st);
    return rc;
  }
  kg.gp = gp;
  uint64_t* clk = ctx->prof_on ? ctx->clk : nullptr;
  kg.jtab = ks->joint;
  prof_mark(ctx, st, 2);
  if (ks->joint && ks->jw == 6)
    k_verify_fast<MODE_ECDSA_KEYSET_JOINT6><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, nullptr, (const uint8_t*)d_r, prep, qt, fin,
                                                                           ctx->gt_call, (uint8_t*)d_valid, wl_count, wl, stride, nullptr, clk, kg);
  else if (ks->joint && ks->jw == 5)
    k_verify_fast<MODE_ECDSA_KEYSET_JOINT5><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, nullptr, (const uint8_t*)d_r, prep, qt, fin,
                                                                           ctx->gt_call, (uint8_t*)d_valid, wl_count, wl, stride, nullptr, clk, kg);
  else if (ks->joint)
    k_verify_fast<MODE_ECDSA_KEYSET_JOINT><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, nullptr, (const uint8_t*)d_r, prep, qt, fin,
                                                                          ctx->gt_call, (uint8_t*)d_valid, wl_count, wl, stride, nullptr, clk, kg);
  else
    k_verify_fast<MODE_ECDSA_KEYSET><<<blocks_for(n), 256, 0, st>>>((uint32_t)n | kvf, nullptr, (const uint8_t*)d_r, prep, qt, fin, ctx->gt_call,
                                                                    (uint8_t*)d_valid, wl_count, wl, stride, nullptr, clk, kg);
  prof_mark(ctx, st, 3);
  HIP_TRY(ctx, hipGetLastError());
  k_verify_fallback_keyset<<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, ks->base + off[0], (const uint32_t*)d_key_index,
                                                                    (const uint8_t*)d_dig, (const uint8_t*)d_r, (const uint8_t*)d_s, flags,
                                                                    (uint8_t*)d_valid, ctx->gt_call, qt, stride);
  HIP_TRY(ctx, hipGetLastError());
  prof_mark(ctx, st, 4);
  prof_mark(ctx, st, 5);
  return ctx_leave(ctx, st);
}

int s2k_ecdsa_verify_batch_keyset(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const uint32_t* key_index, const uint8_t* dig,
                                  const uint8_t* r, const uint8_t* s, uint32_t flags, uint8_t* valid) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!key_index || !dig || !r || !s || !valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  const size_t sizes[5] = {n * 4, n * 32, n * 32, n * 32, n};
  if (s2k_internal_small_call(ctx, n, flags)) {             // (s2k_internal_small_block: no DMA transfers)
    uint8_t *h[5], *dv[5];
    rc = s2k_internal_small_block(ctx, sizes, 5, h, dv);
    if (rc) return rc;
    memcpy(h[0], key_index, n * 4);
    memcpy(h[1], dig, n * 32);
    memcpy(h[2], r, n * 32);
    memcpy(h[3], s, n * 32);
    rc = s2k_ecdsa_verify_batch_keyset_device(ctx, ks, n, dv[0], dv[1], dv[2], dv[3], flags, dv[4], ctx->s_comp);
    if (rc) {
      s2k_internal_drain(ctx);
      return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
    memcpy(valid, h[4], n);
    ctx->have_last = false;
    return S2K_OK;
  }
  uint8_t* d[5];
  rc = ctx_stage(ctx, sizes, 5, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc) return rc;
  s2k_phase_guard phase(ctx->device, n * 100);           // (two verifiers on two threads: engine_internal.h)
  HIP_TRY(ctx, hipMemcpyAsync(d[0], key_index, n * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[1], dig, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[2], r, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[3], s, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_ecdsa_verify_batch_keyset_device(ctx, ks, n, d[0], d[1], d[2], d[3], flags, d[4], st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(valid, d[4], n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

int s2k_ctx_set_key_grouping(s2k_ctx* ctx, int mode, uint32_t min_group, uint32_t hash_bits, uint32_t max_tables) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (mode != S2K_KEYS_OFF && mode != S2K_KEYS_AUTO && mode != S2K_KEYS_ALWAYS && mode != S2K_KEYS_ADAPTIVE)
    return fail(ctx, S2K_ERR_ARG, "key grouping mode %d", mode);
  if (hash_bits > 30) return fail(ctx, S2K_ERR_ARG, "hash_bits %u > 30", hash_bits);
  ctx->kg_mode = mode;
  ctx->kg_min_group = min_group;
  ctx->kg_hash_bits = hash_bits;
  ctx->kg_max_tables = max_tables ? max_tables : KG_MAX_TABLES_DEFAULT;
  ctx->kg_table_cap = 0;           // (a cap found by an earlier allocation failure is tried afresh)
  return S2K_OK;
}

int s2k_ctx_key_grouping_stats(s2k_ctx* ctx, uint32_t* stats) {
  if (!ctx || !stats) return fail(ctx, S2K_ERR_ARG, "null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (ctx->kg_counters) {
    uint32_t c[KG_COUNTERS];
    HIP_TRY(ctx, hipMemcpy(c, ctx->kg_counters, sizeof c, hipMemcpyDeviceToHost));
    stats[0] = c[KG_NKEYED];
    stats[1] = c[KG_NTAB] < ctx->kg_last_max_tables ? c[KG_NTAB] : ctx->kg_last_max_tables;
    stats[2] = c[KG_NLEFT];
  }
  if (ctx->last_wl_count) HIP_TRY(ctx, hipMemcpy(&stats[3], ctx->last_wl_count, sizeof(uint32_t), hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_ctx_key_grouping_adaptive(s2k_ctx* ctx, uint32_t* out, int reset) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (out) {
    out[0] = ctx->kga_miss_streak;
    out[1] = ctx->kga_miss_streak >= KG_ADAPT_MISSES ? ctx->kga_skip_left : 0;
    out[2] = ctx->kga_skipped;
    out[3] = ctx->kga_probes;
    out[4] = ctx->kga_observed;
  }
  if (reset) {
    ctx->kga_miss_streak = ctx->kga_skip_left = 0;
    ctx->kga_skipping = false;
    ctx->kga_seen = ctx->kga_seq;                      // notes still on their way are not taken
  }
  return S2K_OK;
}

int s2k_ecdsa_recover_batch_device(s2k_ctx* ctx, size_t n, const void* d_dig, const void* d_r, const void* d_s,
                                   const void* d_recid, uint32_t flags, void* d_pub65, void* d_ok, void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!d_dig || !d_r || !d_s || !d_recid || !d_pub65 || !d_ok) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* pref = ws + WS_PREF * stride;
  uint32_t* smont = ws + WS_SMONT * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  const uint8_t *dig = (const uint8_t*)d_dig, *r = (const uint8_t*)d_r, *s = (const uint8_t*)d_s,
                *rid = (const uint8_t*)d_recid;
  if (flags & S2K_ECDSA_FORCE_COMPLETE) {
    k_recover_fallback<<<blocks_for(n), 256, 0, st>>>(wl_count, wl, (uint32_t)n, dig, r, s, rid, (uint8_t*)d_ok,
                                                      (uint8_t*)d_pub65, ctx->gt_call, qt, stride);
    HIP_TRY(ctx, hipGetLastError());
    return ctx_leave(ctx, st);
  }
  // (stage times, s2k_ctx_profile_read_stages: [0] the scalar preparation, [2] the ladder, [3] the shared inversions of the
  // affine results, [4] the worklist)
  prof_mark(ctx, st, 0);
  if (n <= ctx->row_max) {
    // small batches: a wave per item, one launch (k_recover_row: preparation, ladder, inversion and record)
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_recover_row<<<(unsigned)((n + 3) / 4), 320, 0, st>>>((uint32_t)n, dig, r, s, rid, ctx->gt_call, (uint8_t*)d_ok, (uint8_t*)d_pub65);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  }
  if (n <= ctx->quad_max) {
    // four lanes per item (k_recover_quad) between a preparation and a finish with one lane per item
    k_scalar_prep<<<(unsigned)((n + 63) / 64), 64, 0, st>>>((uint32_t)n, (uint32_t)n, dig, r, s, rid, 0u, prep, pref, smont, stride);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemsetAsync(d_pub65, 0, n * 65, st));      // items without a key keep the zero record
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_recover_quad<<<(unsigned)((n + 63) / 64), 256, 0, st>>>((uint32_t)n, r, prep, ctx->gt_call, fin, (uint8_t*)d_ok, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    k_affine_finish<MODE_RECOVER><<<(unsigned)((n + 63) / 64), 64, 0, st>>>((uint32_t)n, (uint32_t)n, nullptr, fin, (uint8_t*)d_ok, stride,
                                                                            (uint8_t*)d_pub65);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  }
  HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));
  const uint32_t T = prep_lanes(n);
  k_scalar_prep<<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, dig, r, s, rid, 0u, prep, pref, smont, stride);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemsetAsync(d_pub65, 0, n * 65, st));      // items without a key keep the zero record
  prof_mark(ctx, st, 1);
  prof_mark(ctx, st, 2);
  k_verify_fast<MODE_RECOVER><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, nullptr, r, prep, qt, fin, ctx->gt_call,
                                                             (uint8_t*)d_ok, wl_count, wl, stride, (uint8_t*)d_pub65, ctx->prof_on ? ctx->clk : nullptr,
                                                             key_groups{});
  HIP_TRY(ctx, hipGetLastError());
  prof_mark(ctx, st, 3);
  {
    const uint32_t T = (uint32_t)((n + FIN_M - 1) / FIN_M);
    k_affine_finish<MODE_RECOVER><<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, nullptr, fin, (uint8_t*)d_ok, stride,
                                                                 (uint8_t*)d_pub65);
    HIP_TRY(ctx, hipGetLastError());
  }
  prof_mark(ctx, st, 4);
  k_recover_worklist<<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, dig, r, s, rid, (uint8_t*)d_ok, (uint8_t*)d_pub65,
                                                              ctx->gt_call, qt, stride);
  HIP_TRY(ctx, hipGetLastError());
  prof_mark(ctx, st, 5);
  return ctx_leave(ctx, st);
}

}  // extern "C"
// Small synchronous calls from host memory (up to the small-batch threshold, s2k_ctx_set_small_batch_max): the inputs are
// copied by the CPU into a page-locked block that the kernels read in place, and the kernels write the results there - no
// DMA transfer at all.  The five transfers of a 1024-signature call cost 45 us of its 0.23 ms, more than its 160 KB take
// either way.  `count` pieces of `sizes[i]` bytes, 256-byte aligned; host[i] / dev[i]: the same piece as the CPU and as the
// device address it.  (Synchronous entry points only: the block is free again when the call returns.)
__attribute__((visibility("hidden"))) int s2k_internal_small_block(s2k_ctx* ctx, const size_t* sizes, int count, uint8_t** host, uint8_t** dev) {
  size_t total = 0;
  for (int i = 0; i < count; ++i) total += (sizes[i] + 255) & ~(size_t)255;
  if (total > ctx->h_small_bytes) {
    if (ctx->h_small) (void)hipHostFree(ctx->h_small);
    ctx->h_small = ctx->d_small = nullptr;
    ctx->h_small_bytes = 0;
    const size_t want = (total + ((size_t)1 << 20) - 1) & ~(((size_t)1 << 20) - 1);
    void *h = nullptr, *d = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&h, want, hipHostMallocDefault));
    if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipHostFree(h);
      return fail(ctx, S2K_ERR_HIP, "hipHostGetDevicePointer failed for the small-call block");
    }
    ctx->h_small = (uint8_t*)h;
    ctx->d_small = (uint8_t*)d;
    ctx->h_small_bytes = want;
  }
  size_t off = 0;
  for (int i = 0; i < count; ++i) {
    host[i] = ctx->h_small + off;
    dev[i] = ctx->d_small + off;
    off += (sizes[i] + 255) & ~(size_t)255;
  }
  return S2K_OK;
}
__attribute__((visibility("hidden"))) bool s2k_internal_small_call(const s2k_ctx* ctx, size_t n, uint32_t flags) {
  static const bool off = [] { const char* v = getenv("S2K_SMALL_CALLS_DMA"); return v && atoi(v) != 0; }();   // (A/B knob)
  return !off && n <= ctx->row_max && !(flags & (S2K_ECDSA_FORCE_COMPLETE | S2K_ECDSA_FORCE_WORKLIST));
}
extern "C" {
int s2k_ecdsa_recover_batch(s2k_ctx* ctx, size_t n, const uint8_t* dig, const uint8_t* r, const uint8_t* s,
                            const uint8_t* recid, uint32_t flags, uint8_t* pub65, uint8_t* ok) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!dig || !r || !s || !recid || !pub65 || !ok) return fail(ctx, S2K_ERR_ARG, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  if (s2k_internal_small_call(ctx, n, flags)) {
    const size_t sizes[6] = {n * 32, n * 32, n * 32, n, n * 65, n};
    uint8_t *h[6], *d[6];
    rc = s2k_internal_small_block(ctx, sizes, 6, h, d);
    if (rc) return rc;
    memcpy(h[0], dig, n * 32);
    memcpy(h[1], r, n * 32);
    memcpy(h[2], s, n * 32);
    memcpy(h[3], recid, n);
    rc = s2k_ecdsa_recover_batch_device(ctx, n, d[0], d[1], d[2], d[3], flags, d[4], d[5], ctx->s_comp);
    if (rc) {
      s2k_internal_drain(ctx);
      return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
    memcpy(pub65, h[4], n * 65);
    memcpy(ok, h[5], n);
    ctx->have_last = false;
    return S2K_OK;
  }
  // staged in the context's buffers on its compute stream (no allocation per call)
  const size_t sizes[6] = {n * 32, n * 32, n * 32, n, n * 65, n};
  uint8_t* d[6];
  rc = ctx_stage(ctx, sizes, 6, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc) return rc;
  s2k_phase_guard phase(ctx->device, n * 97);            // (two verifiers on two threads: engine_internal.h)
  HIP_TRY(ctx, hipMemcpyAsync(d[0], dig, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[1], r, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[2], s, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[3], recid, n, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_ecdsa_recover_batch_device(ctx, n, d[0], d[1], d[2], d[3], flags, d[4], d[5], st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(pub65, d[4], n * 65, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(ok, d[5], n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

int s2k_schnorr_verify_batch_device(s2k_ctx* ctx, size_t n, const void* d_pk, const void* d_msgs,
                                    const void* d_msg_offsets, size_t msg_len, const void* d_sig, uint32_t flags,
                                    void* d_valid, void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!d_pk || !d_sig || !d_valid || (!d_msgs && (d_msg_offsets || msg_len)))
    return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH || msg_len > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  const uint8_t* pk = (const uint8_t*)d_pk;
  const uint8_t* sig = (const uint8_t*)d_sig;
  const uint8_t* msgs = (const uint8_t*)d_msgs;
  const uint64_t* offs = (const uint64_t*)d_msg_offsets;
  if (flags & S2K_ECDSA_FORCE_COMPLETE) {
    k_schnorr_fallback<<<blocks_for(n), 256, 0, st>>>(wl_count, wl, (uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len,
                                                      (uint8_t*)d_valid, ctx->gt_call, qt, stride);
    HIP_TRY(ctx, hipGetLastError());
    return ctx_leave(ctx, st);
  }
  // (stage times, s2k_ctx_profile_read_stages: [0] the front end - challenge hashes, and the grouping and per-key tables when
  // keys repeat -, [2] the ladder over per-key tables or the general one, [3] what the key tables left + the shared inversions,
  // [4] the worklist)
  prof_mark(ctx, st, 0);
  ctx->kg_counters = nullptr;
  bool grouped = ctx->kg_mode != S2K_KEYS_OFF && n >= KG_MIN_BATCH;
  const bool row = n <= ctx->row_max && (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF);
  const bool quad = !row && n <= ctx->quad_max && (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF);
  if (row || quad) grouped = false;
  else HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));
  ctx->last_wl_count = (row || quad) ? nullptr : wl_count;
  if (grouped) {
    rc = s2k_internal_key_reserve(ctx, n, 32);
    if (rc == S2K_ERR_NOMEM) grouped = false;      // no room for per-key tables: the general ladder for everything
    else if (rc) return rc;
  }
  if (grouped) {
    // signatures that share an x-only key: the grouped flow of the ECDSA path (s2k_ctx_set_key_grouping)
    uint32_t* gp = ws + WS_GP * stride;
    key_groups kg;
    rc = grouped_front(ctx, st, n, pk, 32, prep, gp, stride,
                       [&](hipStream_t aux) {
                         k_schnorr_prep<<<blocks_for(n), 256, 0, aux>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride);
                       },
                       &kg);
    if (rc) return rc;
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_verify_fast<MODE_SCHNORR_KEYED><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, prep, qt, fin, ctx->gt_call,
                                                                     (uint8_t*)d_valid, wl_count, wl, stride, nullptr, ctx->prof_on ? ctx->clk : nullptr, kg);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    if (kg.nparts > 1) {
      HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_part1, 0));
      kg.part = 1;
      k_verify_fast<MODE_SCHNORR_KEYED><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, prep, qt, fin, ctx->gt_call,
                                                                       (uint8_t*)d_valid, wl_count, wl, stride, nullptr, nullptr, kg);
      HIP_TRY(ctx, hipGetLastError());
    }
    k_verify_fast<MODE_SCHNORR_LEFT><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, prep, qt, fin, ctx->gt_call,
                                                                    (uint8_t*)d_valid, wl_count, wl, stride, nullptr, nullptr, kg);
    HIP_TRY(ctx, hipGetLastError());
  } else if (quad) {
    // four lanes per signature (k_schnorr_quad) behind the preparation kernel; nothing left for the finish and worklist kernels
    k_schnorr_prep<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_schnorr_quad<<<(unsigned)((n + 63) / 64), 256, 0, st>>>((uint32_t)n, pk, sig, prep, ctx->gt_call, (uint8_t*)d_valid, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  } else if (row) {
    // small batches: a wave per signature, nothing left for the finish and worklist kernels (k_schnorr_row): one launch up to
    // 1024 signatures, the preparation kernel in front above
    if (n <= 1024) {
      prof_mark(ctx, st, 1);
      prof_mark(ctx, st, 2);
      k_schnorr_row<true><<<(unsigned)((n + 3) / 4), 320, 0, st>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, nullptr, 0,
                                                                   ctx->gt_call, (uint8_t*)d_valid);
    } else {
      k_schnorr_prep<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride);
      HIP_TRY(ctx, hipGetLastError());
      prof_mark(ctx, st, 1);
      prof_mark(ctx, st, 2);
      k_schnorr_row<false><<<(unsigned)((n + 3) / 4), 256, 0, st>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride,
                                                                    ctx->gt_call, (uint8_t*)d_valid);
    }
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
    prof_mark(ctx, st, 4);
    prof_mark(ctx, st, 5);
    return ctx_leave(ctx, st);
  } else {
    k_schnorr_prep<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride);
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 1);
    prof_mark(ctx, st, 2);
    k_verify_fast<MODE_SCHNORR><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, pk, sig, prep, qt, fin, ctx->gt_call,
                                                               (uint8_t*)d_valid, wl_count, wl, stride, nullptr, ctx->prof_on ? ctx->clk : nullptr,
                                                               key_groups{});
    HIP_TRY(ctx, hipGetLastError());
    prof_mark(ctx, st, 3);
  }
  {
    const uint32_t T = (uint32_t)((n + FIN_M - 1) / FIN_M);
    k_affine_finish<MODE_SCHNORR><<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, sig, fin, (uint8_t*)d_valid, stride, nullptr);
    HIP_TRY(ctx, hipGetLastError());
  }
  prof_mark(ctx, st, 4);
  k_schnorr_worklist<<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, pk, sig, msgs, offs, (uint32_t)msg_len,
                                                              (uint8_t*)d_valid, ctx->gt_call, qt, stride);
  HIP_TRY(ctx, hipGetLastError());
  prof_mark(ctx, st, 5);
  return ctx_leave(ctx, st);
}

// ---------------------------------------------------------------------------------------
// BIP-340 over a key set: SchnorrPublicKey.Verify (secec/bitcoin/schnorr.go:221-253) for signatures that name their key by its
// index in a key set.  The set holds X || Y keys; BIP-340's key is the x coordinate and its point lift_x(x), the one with
// even y (NewSchnorrPublicKeyFromPoint, schnorr.go:276-300, does the same to a point): the challenge hashes X, and the
// ladder takes the set's tables with both half scalars' signs flipped where the stored Y is odd.  Same verdicts as
// s2k_schnorr_verify_batch on the expanded x-only key array; what is saved is grouping, key lifts and table build of every
// call, and with joint tables half (or more) of the ladder.
// ---------------------------------------------------------------------------------------
}  // extern "C"
namespace {
// pk32[i] = X of key kidx[i] (zeros for an index outside the set: that signature is on no ladder lane and stays invalid)
__global__ void __launch_bounds__(256)
k_ks_expand_x(uint32_t n, uint32_t nkeys, const uint32_t* __restrict__ kidx, const uint8_t* __restrict__ keys, uint8_t* __restrict__ pk32) {
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;          // one 16-byte half of a key per thread
  const uint32_t i = t >> 1, h = t & 1u;
  if (i >= n) return;
  const uint32_t k = kidx[i];
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (k < nkeys) v = reinterpret_cast<const uint4*>(keys + (size_t)k * 64)[h];
  reinterpret_cast<uint4*>(pk32 + (size_t)i * 32)[h] = v;
}
}  // namespace
extern "C" {
int s2k_schnorr_verify_batch_keyset_device(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const void* d_key_index, const void* d_msgs,
                                           const void* d_msg_offsets, size_t msg_len, const void* d_sig, uint32_t flags, void* d_valid,
                                           void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  const s2k_ctx* owner = ctx->parent ? ctx->parent : ctx;
  if (!ks || ks->ctx != owner || ks->generation != owner->generation || ks->device != ctx->device)
    return fail(ctx, S2K_ERR_ARG, "key set of another context");
  if (n == 0) return S2K_OK;
  if (!d_key_index || !d_sig || !d_valid || (!d_msgs && (d_msg_offsets || msg_len))) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH || msg_len > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  if (flags) return fail(ctx, S2K_ERR_ARG, "s2k_schnorr_verify_batch_keyset takes no flags");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  rc = ctx_aux_streams(ctx);
  if (rc) return rc;
  rc = s2k_internal_keyset_reserve(ctx, ks->n, n);      // before the fork: growing a buffer synchronises the device
  if (rc) return rc;
  // the expanded x-only keys (what the preparation hashes and the worklist kernel lifts): the context's key-expansion buffer
  rc = ctx_reserve(ctx, &ctx->xkeys, &ctx->xkeys_bytes, n * 32 + 256);
  if (rc) return rc;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* gp = ws + WS_GP * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  uint8_t* pk = (uint8_t*)ctx->xkeys;
  const uint8_t* sig = (const uint8_t*)d_sig;
  const uint8_t* msgs = (const uint8_t*)d_msgs;
  const uint64_t* offs = (const uint64_t*)d_msg_offsets;
  size_t off[5];
  (void)s2k_internal_keyset_bytes(ks->n, off);
  const uint8_t* set_keys = ks->base + off[0];
  ctx->kg_counters = nullptr;
  ctx->last_wl_count = wl_count;
  HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));
  HIP_TRY(ctx, hipMemsetAsync(d_valid, 0, n, st));       // signatures naming no key of the set stay invalid
  k_ks_expand_x<<<blocks_for(2 * n), 256, 0, st>>>((uint32_t)n, (uint32_t)ks->n, (const uint32_t*)d_key_index, set_keys, pk);
  HIP_TRY(ctx, hipGetLastError());
  // second stream: challenge hashes, scalars and the generator part s*G; caller's: the sort by key index
  HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
  HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_fork, 0));
  k_schnorr_prep<<<blocks_for(n), 256, 0, ctx->s_aux>>>((uint32_t)n, pk, sig, msgs, offs, (uint32_t)msg_len, prep, stride);
  k_generator_part<<<blocks_for(n), 256, 0, ctx->s_aux>>>(0u, (uint32_t)n, prep, ctx->gt_call, gp, stride);
  rc = hipGetLastError() == hipSuccess ? S2K_OK : fail(ctx, S2K_ERR_HIP, "launch failed");
  key_groups kg{};
  if (rc == S2K_OK) rc = s2k_internal_keyset_sort(ctx, ks->base, ks->n, n, (const uint32_t*)d_key_index, st, &kg);
  ctx_aux_join(ctx, st);
  if (rc) {
    (void)ctx_leave(ctx, st);
    return rc;
  }
  kg.gp = gp;
  kg.jtab = ks->joint;
#define S2K_SKS_LAUNCH(M) \
  k_verify_fast<M><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, set_keys, sig, prep, qt, fin, ctx->gt_call, (uint8_t*)d_valid, wl_count, wl, stride, \
                                                  nullptr, nullptr, kg)
  if (ks->joint && ks->jw == 6) S2K_SKS_LAUNCH(MODE_SCHNORR_KEYSET_JOINT6);
  else if (ks->joint && ks->jw == 5) S2K_SKS_LAUNCH(MODE_SCHNORR_KEYSET_JOINT5);
  else if (ks->joint) S2K_SKS_LAUNCH(MODE_SCHNORR_KEYSET_JOINT);
  else S2K_SKS_LAUNCH(MODE_SCHNORR_KEYSET);
#undef S2K_SKS_LAUNCH
  HIP_TRY(ctx, hipGetLastError());
  {
    const uint32_t T = (uint32_t)((n + FIN_M - 1) / FIN_M);
    k_affine_finish<MODE_SCHNORR><<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, sig, fin, (uint8_t*)d_valid, stride, nullptr);
    HIP_TRY(ctx, hipGetLastError());
  }
  // (the worklist kernel re-does its lanes from the x-only key: lift_x gives the even-y point whatever the set's Y says)
  k_schnorr_worklist<<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, pk, sig, msgs, offs, (uint32_t)msg_len, (uint8_t*)d_valid,
                                                              ctx->gt_call, qt, stride);
  HIP_TRY(ctx, hipGetLastError());
  return ctx_leave(ctx, st);
}

// Host-buffer form: key indices, messages and signatures staged on the context's compute stream
int s2k_schnorr_verify_batch_keyset(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const uint32_t* key_index, const uint8_t* msgs,
                                    const uint64_t* msg_offsets, size_t msg_len, const uint8_t* sig, uint32_t flags, uint8_t* valid) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!key_index || !sig || !valid || (!msgs && (msg_offsets || msg_len))) return fail(ctx, S2K_ERR_ARG, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  const size_t msg_bytes = msg_offsets ? (size_t)msg_offsets[n] : n * msg_len;
  const size_t sizes[5] = {n * 4, msg_bytes + 16, msg_offsets ? (n + 1) * 8 : 8, n * 64, n};
  uint8_t* d[5];
  rc = ctx_stage(ctx, sizes, 5, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc) return rc;
  s2k_phase_guard phase(ctx->device, n * 68 + msg_bytes);   // (two verifiers on two threads: engine_internal.h)
  HIP_TRY(ctx, hipMemcpyAsync(d[0], key_index, n * 4, hipMemcpyHostToDevice, st));
  if (msg_bytes) HIP_TRY(ctx, hipMemcpyAsync(d[1], msgs, msg_bytes, hipMemcpyHostToDevice, st));
  if (msg_offsets) HIP_TRY(ctx, hipMemcpyAsync(d[2], msg_offsets, (n + 1) * 8, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[3], sig, n * 64, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_schnorr_verify_batch_keyset_device(ctx, ks, n, d[0], msgs ? d[1] : nullptr, msg_offsets ? d[2] : nullptr, msg_len, d[3], flags, d[4], st);
  if (rc) {
    s2k_internal_drain(ctx);
    return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(valid, d[4], n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

// Host-buffer entry point.  The batch is cut into chunks that are whole rounds of k_verify_fast
// on this device (3 waves/SIMD x 4 SIMDs x CUs x 64 lanes; one round first, then two per chunk), and the
// host-to-device copy of chunk j+1 runs on its own stream while chunk j is verified: for pageable
// callers' memory the copy costs about half as much as the verification, so most of it hides.
// Staging buffers and streams live in the context (no hipMalloc per call).
// page-locked host memory the runtime can copy from asynchronously (hipHostMalloc or hipHostRegister)
static bool host_pinned_at(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();          // an ordinary malloc'ed pointer: "invalid value", not an error of ours
    return false;
  }
  return a.type == hipMemoryTypeHost;
}
// (first AND last byte: a buffer registered for fewer bytes than the call reads must not take the asynchronous path)
static bool host_pinned(const void* p, size_t bytes) {
  return host_pinned_at(p) && (bytes < 2 || host_pinned_at((const uint8_t*)p + bytes - 1));
}
void* s2k_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void s2k_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}
int s2k_host_register(void* p, size_t bytes) {
  if (!p || !bytes) return fail(nullptr, S2K_ERR_ARG, "s2k_host_register: null buffer");
  // Whole pages only.  The runtime pins and, on unregister, unmaps PAGES: registering a block of the C heap takes the
  // neighbouring blocks of its first and last page along.  Seen on ROCm 7.2 / MI355X: four small malloc'ed arrays
  // registered and unregistered, the allocator hands the same pages out again as part of a 2 MB block, and a plain
  // pageable hipMemcpy from that block dies with a GPU memory access fault (tests/test_gpu_round3.py history).
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  if (((uintptr_t)p % page) || (bytes % page))
    return fail(nullptr, S2K_ERR_ARG, "s2k_host_register: buffer must start on a page boundary and cover whole pages (%zu bytes): "
                                      "register memory that owns its pages (mmap, aligned_alloc), or use s2k_host_alloc", page);
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, S2K_ERR_HIP, "hipHostRegister: %s", hipGetErrorString(e));
  }
  return S2K_OK;
}
int s2k_host_unregister(void* p) {
  if (!p) return fail(nullptr, S2K_ERR_ARG, "s2k_host_unregister: null buffer");
  hipError_t e = hipHostUnregister(p);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return fail(nullptr, S2K_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(e));
  }
  return S2K_OK;
}

// Everything of a host-buffer verification except the final wait: copies in, kernels, verdicts to h_out (host memory; the
// copy is asynchronous when h_out is page-locked).  All work ends on ctx->s_comp.  one_shot: the batch is not cut into
// chunks (submit / wait: the other slot's batch is what this one's transfer hides behind, and ONE grouped call sees every
// signature of a key).  On an error nothing of the call is left in flight.
static int verify_batch_enqueue_inner(s2k_ctx* ctx, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r,
                                      const uint8_t* s, uint32_t flags, uint8_t* h_out, bool one_shot) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  rc = ctx_reserve(ctx, &ctx->io, &ctx->io_bytes, n * 161 + 1024);
  if (rc) return rc;
  uint8_t* d_pub = (uint8_t*)ctx->io;
  uint8_t* d_dig = d_pub + n * 64;
  uint8_t* d_r = d_dig + n * 32;
  uint8_t* d_s = d_r + n * 32;
  uint8_t* d_valid = d_s + n * 32;
  // Pinned host buffers (s2k_host_alloc / s2k_host_register, or the caller's own hipHostMalloc): copies are truly
  // asynchronous, so with the grouping on the whole batch goes through ONE grouped call - the keys are copied first, the
  // grouping and the per-key tables start on them, and the digests and signatures cross PCIe meanwhile (the scalar
  // preparation on the second stream waits for them).  From pageable memory a copy is staged by the runtime and the
  // same order measured slower than two independent halves (9.0 against 8.4-8.6 ms per 2^20), which stay the path there.
  const bool pinned = host_pinned(pub, n * 64) && host_pinned(dig, n * 32) && host_pinned(r, n * 32) && host_pinned(s, n * 32);
  // (submit / wait, one_shot: no pieces - the other tickets' kernels are what the transfer hides behind, and the ticket runs
  // one compute stream; measured 5.07 against 5.13 ms per 2^20 with the pieces.  S2K_SUBMIT_ARRIVALS=1: measurement knob)
  static const bool submit_arrivals = [] { const char* v = getenv("S2K_SUBMIT_ARRIVALS"); return v && atoi(v) != 0; }();
  if (pinned && ctx->kg_mode != S2K_KEYS_OFF && n >= KG_MIN_BATCH && (!one_shot || submit_arrivals)) {
    rc = ctx_arrival_events(ctx);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(d_pub, pub, n * 64, hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_copied[0], ctx->s_copy));
    sig_arrivals arr;
    arr.count = n >= ((size_t)1 << 16) ? 4 : 1;
    const size_t per = ((n + arr.count - 1) / arr.count + 255) & ~(size_t)255;
    for (int c = 0; c < arr.count; ++c) {
      const size_t lo = (size_t)c * per, cnt = lo >= n ? 0 : (n - lo < per ? n - lo : per);
      arr.lo[c] = lo;
      arr.cnt[c] = cnt;
      arr.ev[c] = ctx->ev_arrival[c];
      if (cnt) {
        HIP_TRY(ctx, hipMemcpyAsync(d_r + lo * 32, r + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
        HIP_TRY(ctx, hipMemcpyAsync(d_s + lo * 32, s + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
        HIP_TRY(ctx, hipMemcpyAsync(d_dig + lo * 32, dig + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
      }
      HIP_TRY(ctx, hipEventRecord(arr.ev[c], ctx->s_copy));
    }
    while (arr.count > 1 && arr.cnt[arr.count - 1] == 0) --arr.count;
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_comp, ctx->ev_copied[0], 0));
    rc = verify_batch_device(ctx, n, d_pub, d_dig, d_r, d_s, flags, d_valid, ctx->s_comp, &arr);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_out, d_valid, n, hipMemcpyDeviceToHost, ctx->s_comp));
    return S2K_OK;
  }
  const size_t round = (size_t)S2K_FAST_WAVES * 4 * (size_t)ctx->cu_count * 64;
  // chunks are whole rounds of the ladder kernel; the first one is a single round, because its copy is
  // the only one nothing hides (63 -> 31 MB up front at 256 CUs), the others two rounds
  const bool chunked = !one_shot && n > 3 * round;         // small batches: one shot
  // with key grouping on, every chunk groups (and builds tables) on its own, so fewer and larger chunks:
  // two halves, the second copy hidden behind the first half's kernels
  const bool halves = chunked && ctx->kg_mode != S2K_KEYS_OFF;
  int k = 0;
  for (size_t lo = 0, chunk = 0; lo < n; lo += chunk, ++k) {
    chunk = !chunked ? n : halves ? ((n / 2 + 255) & ~(size_t)255) : (lo == 0 ? round : 2 * round);
    if (n - lo < chunk + round) chunk = n - lo;            // no launch of less than a round at the end
    const size_t cnt = chunk;
    // the event of two chunks ago has been waited on by s_comp (in stream order) before it is re-recorded
    HIP_TRY(ctx, hipMemcpyAsync(d_pub + lo * 64, pub + lo * 64, cnt * 64, hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipMemcpyAsync(d_dig + lo * 32, dig + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipMemcpyAsync(d_r + lo * 32, r + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipMemcpyAsync(d_s + lo * 32, s + lo * 32, cnt * 32, hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_copied[k & 1], ctx->s_copy));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_comp, ctx->ev_copied[k & 1], 0));
    rc = s2k_ecdsa_verify_batch_device(ctx, cnt, d_pub + lo * 64, d_dig + lo * 32, d_r + lo * 32, d_s + lo * 32, flags,
                                       d_valid + lo, ctx->s_comp);
    if (rc) return rc;
  }
  HIP_TRY(ctx, hipMemcpyAsync(h_out, d_valid, n, hipMemcpyDeviceToHost, ctx->s_comp));
  return S2K_OK;
}
// An error return leaves copies from (and to) caller-owned memory in flight on the context's streams: wait for them before the
// caller may free or unregister that memory (the message of the error is kept).
}  // extern "C"
__attribute__((visibility("hidden"))) void s2k_internal_drain(s2k_ctx* ctx) {
  if (ctx->s_copy) (void)hipStreamSynchronize(ctx->s_copy);
  if (ctx->s_aux) (void)hipStreamSynchronize(ctx->s_aux);
  if (ctx->s_comp) (void)hipStreamSynchronize(ctx->s_comp);
  (void)hipGetLastError();
}
extern "C" {
static int verify_batch_enqueue(s2k_ctx* ctx, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r,
                                const uint8_t* s, uint32_t flags, uint8_t* h_out, bool one_shot) {
  const int rc = verify_batch_enqueue_inner(ctx, n, pub, dig, r, s, flags, h_out, one_shot);
  if (rc) s2k_internal_drain(ctx);
  return rc;
}

int s2k_ecdsa_verify_batch(s2k_ctx* ctx, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r,
                           const uint8_t* s, uint32_t flags, uint8_t* valid) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!pub || !dig || !r || !s || !valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  int rc;
  if (s2k_internal_small_call(ctx, n, flags) && (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF)) {
    // (ctx_small_block: the wave-per-signature ladder reads the caller's bytes from page-locked memory and writes the verdicts
    // there - no DMA transfers)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    rc = ctx_streams(ctx);
    if (rc) return rc;
    const size_t sizes[5] = {n * 64, n * 32, n * 32, n * 32, n};
    uint8_t *h[5], *d[5];
    rc = s2k_internal_small_block(ctx, sizes, 5, h, d);
    if (rc) return rc;
    memcpy(h[0], pub, n * 64);
    memcpy(h[1], dig, n * 32);
    memcpy(h[2], r, n * 32);
    memcpy(h[3], s, n * 32);
    rc = s2k_ecdsa_verify_batch_device(ctx, n, d[0], d[1], d[2], d[3], flags, d[4], ctx->s_comp);
    if (rc) {
      s2k_internal_drain(ctx);
      return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
    memcpy(valid, h[4], n);
    return S2K_OK;
  }
  rc = verify_batch_enqueue(ctx, n, pub, dig, r, s, flags, valid, /*one_shot=*/false);
  if (rc) return rc;
  HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
  return S2K_OK;
}

// ---------------------------------------------------------------------------------------
// submit / wait: the host-pointer entry points without the wait at their end.  The reference's caller holds its data in
// host memory (secec/ecdsa.go:171-228) and a synchronous call pays transfer and compute in series: 7.1 ms from pinned
// memory, 8.2 from pageable against 4.9 ms resident per 2^20 signatures (round 3).  Here the context owns child contexts
// ("slots": own workspaces and staging buffers; the 40 GiB generator tables are shared per device) that take the submitted
// batches in turn.  The children run on the PARENT's streams, in two lanes: even tickets on the parent's own two compute
// streams, odd tickets on a second pair, one copy stream for all.  Within a lane the kernels of consecutive tickets follow
// each other in stream order, each ticket with the two-stream overlap of a resident call (grouping and tables beside
// preparation and generator part); the lanes run beside each other as two resident contexts do (worth 3 %); a ticket's
// 160 MiB cross PCIe (3.0 ms) on the copy stream beside the kernels of the tickets before it.  Up to four tickets are in
// flight (per lane one computing, one arriving); a fifth submit first retires the oldest one (its verdicts are delivered;
// a later s2k_wait on its ticket returns at once).
// (Earlier forms gave every ticket streams of its own so that one ticket's grouping and tables could run beside another's
// ladder.  They do not: the keyed ladder holds 4 x 128 of a SIMD's 512 VGPRs, so the other ticket's kernels crawl, its
// empty tail launches - 161 VGPRs a wave - wait milliseconds for room, and nine streams on the runtime's hardware queues
// serialise at random.  5.2-5.7 ms per batch against 4.7-4.9 this way; profiles/r04_pipeline_timeline.txt.)
// ---------------------------------------------------------------------------------------
}  // extern "C"
static void pipe_note_failure(s2k_ctx* ctx, uint64_t ticket, int rc) {
  const unsigned i = ctx->pipe_failed_n++ % 8u;
  ctx->pipe_failed[i] = ticket;
  ctx->pipe_failed_rc[i] = rc;
}
// waits for the slot's batch and delivers its verdicts; the slot is free afterwards
__attribute__((visibility("hidden"))) int s2k_internal_pipe_retire(s2k_ctx* ctx, s2k_ctx::pipe_slot& sl) {
  if (!sl.ticket) return S2K_OK;
  int rc = S2K_OK;
  if (hipSetDevice(ctx->device) != hipSuccess || hipEventSynchronize(sl.done) != hipSuccess) {
    rc = fail(ctx, S2K_ERR_HIP, "ticket %llu: the batch did not complete: %s", (unsigned long long)sl.ticket,
              hipGetErrorString(hipGetLastError()));
    s2k_internal_drain(sl.ctx);
  } else if (!sl.direct) {
    memcpy(sl.dst, sl.h_valid, sl.n);
  }
  if (rc == S2K_OK && sl.timed) {                    // s2k_ctx_ticket_timing: transfer and whole-ticket time on the device's clock
    float h2d = 0, all = 0;
    // (t_copied sits on the copy stream behind the ticket's transfers; with more streams in the process than hardware queues
    // its marker can still be waiting in a shared queue when the ticket's verdicts are there - eight members on one device
    // showed it: then it is waited for, the copies themselves are long done)
    const hipError_t e0 = hipEventSynchronize(sl.t_end);
    hipError_t e1 = hipEventElapsedTime(&h2d, sl.t_begin, sl.t_copied);
    if (e1 == hipErrorNotReady) {
      (void)hipGetLastError();
      (void)hipEventSynchronize(sl.t_copied);
      e1 = hipEventElapsedTime(&h2d, sl.t_begin, sl.t_copied);
    }
    const hipError_t e2 = hipEventElapsedTime(&all, sl.t_begin, sl.t_end);
    const unsigned i = ctx->pipe_times_n++ % 8u;
    ctx->pipe_times_ticket[i] = sl.ticket;
    if (e0 == hipSuccess && e1 == hipSuccess && e2 == hipSuccess) {
      ctx->pipe_times_ms[i][0] = h2d;
      ctx->pipe_times_ms[i][1] = all;
    } else {                                           // no times for this ticket: the runtime's codes, negated, say why
      (void)hipGetLastError();
      ctx->pipe_times_ms[i][0] = -(float)(e1 != hipSuccess ? e1 : e0);
      ctx->pipe_times_ms[i][1] = -(float)(e2 != hipSuccess ? e2 : e0);
    }
  }
  sl.timed = false;
  sl.ticket = 0;
  return rc;
}
// the slot the next ticket runs on, free, with its child context and landing buffer in place
__attribute__((visibility("hidden"))) int s2k_internal_pipe_slot(s2k_ctx* ctx, size_t n, uint8_t* valid, s2k_ctx::pipe_slot** out) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  s2k_ctx::pipe_slot& sl = ctx->pipe[ctx->pipe_next % s2k_ctx::PIPE_SLOTS];
  if (sl.ticket) {                                   // every slot holds a batch: the oldest one (this slot's) is retired first
    const uint64_t t = sl.ticket;
    const int rc = s2k_internal_pipe_retire(ctx, sl);
    if (rc) pipe_note_failure(ctx, t, rc);           // (reported by s2k_wait on that ticket)
  }
  if (!sl.ctx) {
    // first use of the slot.  All or nothing: a failure anywhere leaves the slot as it was found (no child context, no
    // event), so that the next submit tries again from the start instead of running on a half-built slot.
    int rc = ctx_streams(ctx);                        // the parent's three streams carry every ticket
    if (rc == S2K_OK) rc = ctx_aux_streams(ctx);
    if (rc) return rc;
    const bool odd_lane = (ctx->pipe_next & 1u) != 0;   // (PIPE_SLOTS is even: a slot always serves the same lane)
    static const bool one_lane = [] { const char* v = getenv("S2K_SUBMIT_ONE_LANE"); return v && atoi(v) != 0; }();   // measurement knob
    if (odd_lane && !one_lane && !(ctx->lane1_comp && ctx->lane1_aux)) {   // the second lane's two streams: both or neither
      hipStream_t a = nullptr, b = nullptr;
      if (hipStreamCreateWithFlags(&a, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&b, hipStreamNonBlocking) != hipSuccess) {
        const hipError_t e = hipGetLastError();
        if (a) (void)hipStreamDestroy(a);
        if (b) (void)hipStreamDestroy(b);
        return fail(ctx, S2K_ERR_HIP, "submit: streams of the second lane: %s", hipGetErrorString(e));
      }
      ctx->lane1_comp = a;
      ctx->lane1_aux = b;
    }
    s2k_ctx* child = nullptr;
    rc = s2k_ctx_create_ex(ctx->device, ctx->gt_fixed, 0, &child);   // the parent's table width, explicit or automatic
    if (rc) return fail(ctx, rc, "submit: child context: %s", s2k_last_error(nullptr));
    child->s_copy = ctx->s_copy;
    child->s_comp = (odd_lane && !one_lane) ? ctx->lane1_comp : ctx->s_comp;
    child->s_aux = (odd_lane && !one_lane) ? ctx->lane1_aux : ctx->s_aux;
    child->streams_shared = true;
    rc = ctx_streams(child);                          // (its events)
    if (rc == S2K_OK) rc = ctx_aux_streams(child);
    hipEvent_t done = nullptr;
    if (rc == S2K_OK && hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) {
      snprintf(child->err, sizeof child->err, "hipEventCreate: %s", hipGetErrorString(hipGetLastError()));
      rc = S2K_ERR_HIP;
    }
    if (rc) {
      rc = fail(ctx, rc, "submit: child context: %s", child->err);
      s2k_ctx_destroy(child);                         // (streams_shared: the parent's streams stay)
      return rc;
    }
    sl.ctx = child;
    sl.done = done;
  }
  if (ctx->pipe_timing && !sl.t_begin) {              // timing events on demand; a failure only means no times for this slot
    hipEvent_t e[3] = {nullptr, nullptr, nullptr};
    bool ok = true;
    for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreate(&e[i]) == hipSuccess;
    if (ok) {
      sl.t_begin = e[0];
      sl.t_copied = e[1];
      sl.t_end = e[2];
    } else {
      (void)hipGetLastError();
      for (int i = 0; i < 3; ++i)
        if (e[i]) (void)hipEventDestroy(e[i]);
    }
  }
  sl.timed = ctx->pipe_timing && sl.t_begin && hipEventRecord(sl.t_begin, ctx->s_copy) == hipSuccess;
  // the child verifies with the parent's settings of the moment
  sl.ctx->kg_mode = ctx->kg_mode;
  sl.ctx->parent = ctx;
  sl.ctx->kg_min_group = ctx->kg_min_group;
  sl.ctx->kg_hash_bits = ctx->kg_hash_bits;
  sl.ctx->kg_max_tables = ctx->kg_max_tables;
  sl.ctx->row_max = ctx->row_max;
  sl.ctx->quad_max = ctx->quad_max;
  sl.direct = host_pinned(valid, n);
  if (!sl.direct && sl.h_valid_bytes < n) {
    if (sl.h_valid) (void)hipHostFree(sl.h_valid);
    sl.h_valid = nullptr;
    sl.h_valid_bytes = 0;
    const size_t want = (n + 4095) & ~(size_t)4095;
    HIP_TRY(ctx, hipHostMalloc((void**)&sl.h_valid, want, hipHostMallocDefault));
    sl.h_valid_bytes = want;
  }
  sl.dst = valid;
  sl.n = n;
  *out = &sl;
  return S2K_OK;
}
__attribute__((visibility("hidden"))) void s2k_internal_pipe_issue(s2k_ctx* ctx, s2k_ctx::pipe_slot* sl, s2k_ticket* ticket) {
  // s2k_ctx_ticket_timing: t_begin (s2k_internal_pipe_slot) and t_copied bracket the ticket's transfers on the copy stream; the
  // compute stream WAITS for t_copied before t_end is recorded behind the verdicts, so that first copy start <= last copy end
  // <= verdicts holds on the device's clock however the runtime multiplexes the streams onto hardware queues (eight members on
  // one device: a marker could be processed after another stream's later one).  Costs the ticket nothing
  // while the timing is off.
  if (sl->timed)
    sl->timed = hipEventRecord(sl->t_copied, ctx->s_copy) == hipSuccess && hipStreamWaitEvent(sl->ctx->s_comp, sl->t_copied, 0) == hipSuccess &&
                hipEventRecord(sl->t_end, sl->ctx->s_comp) == hipSuccess;
  if (hipEventRecord(sl->done, sl->ctx->s_comp) != hipSuccess) {    // behind the ticket's last operation (on its lane's stream)
    // no event to wait on: the ticket is made to finish here, so that s2k_wait finds it done (or reports this failure)
    const hipError_t e = hipGetLastError();
    (void)hipStreamSynchronize(sl->ctx->s_comp);
    (void)fail(ctx, S2K_ERR_HIP, "submit: hipEventRecord: %s", hipGetErrorString(e));
  }
  sl->ticket = ctx->pipe_next++;
  *ticket = sl->ticket;
}
__attribute__((visibility("hidden"))) bool s2k_internal_host_pinned(const void* p, size_t bytes) { return host_pinned(p, bytes); }

extern "C" {
int s2k_ecdsa_verify_batch_submit(s2k_ctx* ctx, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r,
                                  const uint8_t* s, uint32_t flags, uint8_t* valid, s2k_ticket* ticket) {
  if (!ctx || !ticket) return fail(ctx, S2K_ERR_ARG, "null argument");
  *ticket = 0;
  if (n && (!pub || !dig || !r || !s || !valid)) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  s2k_ctx::pipe_slot* sl = nullptr;
  int rc = s2k_internal_pipe_slot(ctx, n, valid, &sl);
  if (rc) return rc;
  if (n) {
    rc = verify_batch_enqueue(sl->ctx, n, pub, dig, r, s, flags, sl->direct ? valid : sl->h_valid, /*one_shot=*/true);
    if (rc) return fail(ctx, rc, "%s", sl->ctx->err);
  }
  s2k_internal_pipe_issue(ctx, sl, ticket);
  return S2K_OK;
}

// s2k_ecdsa_verify_batch_keyset without the wait at its end: the ticket's index, digest and signature arrays (100 bytes per
// signature) cross PCIe on the copy stream beside the kernels of the tickets before it; the child context verifies over
// the parent's key set.
int s2k_ecdsa_verify_batch_keyset_submit(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const uint32_t* key_index, const uint8_t* dig,
                                         const uint8_t* r, const uint8_t* s, uint32_t flags, uint8_t* valid, s2k_ticket* ticket) {
  if (!ctx || !ticket) return fail(ctx, S2K_ERR_ARG, "null argument");
  *ticket = 0;
  if (!ks || ks->ctx != ctx || ks->generation != ctx->generation || ks->device != ctx->device)
    return fail(ctx, S2K_ERR_ARG, "key set of another context");
  if (n && (!key_index || !dig || !r || !s || !valid)) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  if (flags & S2K_ECDSA_FORCE_COMPLETE) return fail(ctx, S2K_ERR_ARG, "S2K_ECDSA_FORCE_COMPLETE does not apply to key sets");
  s2k_ctx::pipe_slot* sl = nullptr;
  int rc = s2k_internal_pipe_slot(ctx, n, valid, &sl);
  if (rc) return rc;
  if (n) {
    s2k_ctx* c = sl->ctx;
    uint8_t* h_out = sl->direct ? valid : sl->h_valid;
    auto enqueue = [&]() -> int {
      const size_t sizes[5] = {n * 4, n * 32, n * 32, n * 32, n};
      uint8_t* d[5];
      int rc2 = ctx_stage(c, sizes, 5, d);
      if (rc2) return rc2;
      HIP_TRY(c, hipMemcpyAsync(d[0], key_index, n * 4, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipMemcpyAsync(d[1], dig, n * 32, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipMemcpyAsync(d[2], r, n * 32, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipMemcpyAsync(d[3], s, n * 32, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipEventRecord(c->ev_copied[0], c->s_copy));
      HIP_TRY(c, hipStreamWaitEvent(c->s_comp, c->ev_copied[0], 0));
      rc2 = s2k_ecdsa_verify_batch_keyset_device(c, ks, n, d[0], d[1], d[2], d[3], flags, d[4], c->s_comp);
      if (rc2) return rc2;
      HIP_TRY(c, hipMemcpyAsync(h_out, d[4], n, hipMemcpyDeviceToHost, c->s_comp));
      return S2K_OK;
    };
    rc = enqueue();
    if (rc) {
      s2k_internal_drain(c);
      return fail(ctx, rc, "%s", c->err);
    }
  }
  s2k_internal_pipe_issue(ctx, sl, ticket);
  return S2K_OK;
}

// s2k_schnorr_verify_batch_keyset in the ticket form (fixed-length or offset-delimited messages)
int s2k_schnorr_verify_batch_keyset_submit(s2k_ctx* ctx, const s2k_keyset* ks, size_t n, const uint32_t* key_index, const uint8_t* msgs,
                                           const uint64_t* msg_offsets, size_t msg_len, const uint8_t* sig, uint32_t flags, uint8_t* valid,
                                           s2k_ticket* ticket) {
  if (!ctx || !ticket) return fail(ctx, S2K_ERR_ARG, "null argument");
  *ticket = 0;
  if (!ks || ks->ctx != ctx || ks->generation != ctx->generation || ks->device != ctx->device)
    return fail(ctx, S2K_ERR_ARG, "key set of another context");
  if (n && (!key_index || !sig || !valid || (!msgs && (msg_offsets || msg_len)))) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH || msg_len > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  if (flags) return fail(ctx, S2K_ERR_ARG, "s2k_schnorr_verify_batch_keyset takes no flags");
  s2k_ctx::pipe_slot* sl = nullptr;
  int rc = s2k_internal_pipe_slot(ctx, n, valid, &sl);
  if (rc) return rc;
  if (n) {
    s2k_ctx* c = sl->ctx;
    uint8_t* h_out = sl->direct ? valid : sl->h_valid;
    auto enqueue = [&]() -> int {
      const size_t msg_bytes = msg_offsets ? (size_t)msg_offsets[n] : n * msg_len;
      const size_t sizes[5] = {n * 4, msg_bytes + 16, msg_offsets ? (n + 1) * 8 : 8, n * 64, n};
      uint8_t* d[5];
      int rc2 = ctx_stage(c, sizes, 5, d);
      if (rc2) return rc2;
      HIP_TRY(c, hipMemcpyAsync(d[0], key_index, n * 4, hipMemcpyHostToDevice, c->s_copy));
      if (msg_bytes) HIP_TRY(c, hipMemcpyAsync(d[1], msgs, msg_bytes, hipMemcpyHostToDevice, c->s_copy));
      if (msg_offsets) HIP_TRY(c, hipMemcpyAsync(d[2], msg_offsets, (n + 1) * 8, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipMemcpyAsync(d[3], sig, n * 64, hipMemcpyHostToDevice, c->s_copy));
      HIP_TRY(c, hipEventRecord(c->ev_copied[0], c->s_copy));
      HIP_TRY(c, hipStreamWaitEvent(c->s_comp, c->ev_copied[0], 0));
      rc2 = s2k_schnorr_verify_batch_keyset_device(c, ks, n, d[0], msgs ? d[1] : nullptr, msg_offsets ? d[2] : nullptr, msg_len, d[3], 0, d[4], c->s_comp);
      if (rc2) return rc2;
      HIP_TRY(c, hipMemcpyAsync(h_out, d[4], n, hipMemcpyDeviceToHost, c->s_comp));
      return S2K_OK;
    };
    rc = enqueue();
    if (rc) {
      s2k_internal_drain(c);
      return fail(ctx, rc, "%s", c->err);
    }
  }
  s2k_internal_pipe_issue(ctx, sl, ticket);
  return S2K_OK;
}

int s2k_wait(s2k_ctx* ctx, s2k_ticket ticket) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (ticket == 0 || ticket >= ctx->pipe_next) return fail(ctx, S2K_ERR_ARG, "s2k_wait: ticket %llu was never issued", (unsigned long long)ticket);
  for (s2k_ctx::pipe_slot& sl : ctx->pipe)
    if (sl.ticket == ticket) return s2k_internal_pipe_retire(ctx, sl);
  for (unsigned i = 0; i < 8; ++i)                    // retired by a later submit: its verdicts are delivered (or it failed then)
    if (ctx->pipe_failed[i] == ticket) return fail(ctx, ctx->pipe_failed_rc[i], "ticket %llu failed when it was retired", (unsigned long long)ticket);
  return S2K_OK;
}

// S2K_OK: the ticket's verdicts are delivered (as s2k_wait); S2K_PENDING: still in flight.  Never blocks.
int s2k_poll(s2k_ctx* ctx, s2k_ticket ticket) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (ticket == 0 || ticket >= ctx->pipe_next) return fail(ctx, S2K_ERR_ARG, "s2k_poll: ticket %llu was never issued", (unsigned long long)ticket);
  for (s2k_ctx::pipe_slot& sl : ctx->pipe)
    if (sl.ticket == ticket) {
      HIP_TRY(ctx, hipSetDevice(ctx->device));
      const hipError_t e = hipEventQuery(sl.done);
      if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        return S2K_PENDING;
      }
      return s2k_internal_pipe_retire(ctx, sl);
    }
  return s2k_wait(ctx, ticket);                       // retired earlier
}

int s2k_wait_all(s2k_ctx* ctx) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  int rc = S2K_OK;
  for (;;) {                                          // oldest first
    s2k_ctx::pipe_slot* oldest = nullptr;
    for (s2k_ctx::pipe_slot& sl : ctx->pipe)
      if (sl.ticket && (!oldest || sl.ticket < oldest->ticket)) oldest = &sl;
    if (!oldest) break;
    const int r1 = s2k_internal_pipe_retire(ctx, *oldest);
    if (r1 && !rc) rc = r1;
  }
  return rc;
}

// Batches of up to max_n signatures take the wave-per-signature ladder (k_verify_row); 0 switches it off.
int s2k_ctx_set_small_batch_max(s2k_ctx* ctx, uint32_t max_n) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  ctx->row_max = max_n;
  return S2K_OK;
}
// ECDSA batches above that and up to max_n take the four-lanes-per-signature ladder (k_verify_quad); 0 switches it off.
int s2k_ctx_set_mid_batch_max(s2k_ctx* ctx, uint32_t max_n) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  ctx->quad_max = max_n;
  return S2K_OK;
}
// Per-ticket times on the device's clock (for placement diagnostics, s2k_group_member_stats_ex): enable, submit, wait, ask.
int s2k_ctx_ticket_timing(s2k_ctx* ctx, int enable) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  ctx->pipe_timing = enable != 0;
  return S2K_OK;
}
int s2k_ticket_times(s2k_ctx* ctx, s2k_ticket ticket, double ms[2]) {
  if (!ctx || !ms) return fail(ctx, S2K_ERR_ARG, "null argument");
  for (unsigned i = 0; i < 8; ++i)
    if (ticket && ctx->pipe_times_ticket[i] == ticket) {
      ms[0] = ctx->pipe_times_ms[i][0];
      ms[1] = ctx->pipe_times_ms[i][1];
      return S2K_OK;
    }
  ms[0] = ms[1] = 0;
  return S2K_PENDING;      // not retired yet, timing was off, or more than eight tickets ago
}

int s2k_schnorr_verify_batch(s2k_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msgs,
                             const uint64_t* msg_offsets, size_t msg_len, const uint8_t* sig, uint32_t flags,
                             uint8_t* valid) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!pk || !sig || !valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  size_t total = msg_offsets ? (size_t)msg_offsets[n] : n * msg_len;
  if (total && !msgs) return fail(ctx, S2K_ERR_ARG, "null message buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  const size_t sizes[5] = {n * 32, total ? total : 16, (n + 1) * sizeof(uint64_t), n * 64, n};
  if (s2k_internal_small_call(ctx, n, flags) && total <= ((size_t)1 << 20)) {       // (ctx_small_block: no DMA transfers)
    uint8_t *h[5], *dv[5];
    rc = s2k_internal_small_block(ctx, sizes, 5, h, dv);
    if (rc) return rc;
    memcpy(h[0], pk, n * 32);
    if (total) memcpy(h[1], msgs, total);
    if (msg_offsets) memcpy(h[2], msg_offsets, (n + 1) * sizeof(uint64_t));
    memcpy(h[3], sig, n * 64);
    rc = s2k_schnorr_verify_batch_device(ctx, n, dv[0], dv[1], msg_offsets ? dv[2] : nullptr, msg_len, dv[3], flags, dv[4], ctx->s_comp);
    if (rc) {
      s2k_internal_drain(ctx);
      return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
    memcpy(valid, h[4], n);
    ctx->have_last = false;
    return S2K_OK;
  }
  uint8_t* d[5];
  rc = ctx_stage(ctx, sizes, 5, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc) return rc;
  s2k_phase_guard phase(ctx->device, n * 96 + total);    // (two verifiers on two threads: engine_internal.h)
  HIP_TRY(ctx, hipMemcpyAsync(d[0], pk, n * 32, hipMemcpyHostToDevice, st));
  if (total) HIP_TRY(ctx, hipMemcpyAsync(d[1], msgs, total, hipMemcpyHostToDevice, st));
  if (msg_offsets) HIP_TRY(ctx, hipMemcpyAsync(d[2], msg_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[3], sig, n * 64, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_schnorr_verify_batch_device(ctx, n, d[0], d[1], msg_offsets ? d[2] : nullptr, msg_len, d[3], flags, d[4], st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(valid, d[4], n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

// Point.DoubleScalarMultBasepointVartime (point_mul_glv.go:307), or scalarMultVartimeGLV (:203) when
// u1 == NULL, over a batch of 65-byte records; host pointers.  impl selects the arithmetic (header).
int s2k_double_scalar_mult_basepoint_batch_ex(s2k_ctx* ctx, uint32_t impl, size_t n, const uint8_t* u1, const uint8_t* u2,
                                              const uint8_t* points, uint8_t* out) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (impl != S2K_IMPL_COMPLETE && impl != S2K_IMPL_FAST) return fail(ctx, S2K_ERR_ARG, "unknown implementation selector");
  if (n == 0) return S2K_OK;
  if (!u2 || !points || !out) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  rc = ctx_enter(ctx, st);
  if (rc) return rc;
  rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o_u1 = 0, o_u2 = pad(n * 32), o_pts = o_u2 + pad(n * 32), o_out = o_pts + pad(n * 65), o_pub = o_out + pad(n * 65),
               o_ok = o_pub + pad(n * 64), o_status = o_ok + pad(n);
  rc = ctx_reserve(ctx, &ctx->io, &ctx->io_bytes, o_status + 256);
  if (rc) return rc;
  uint8_t* io = (uint8_t*)ctx->io;
  if (u1)
    HIP_TRY(ctx, hipMemcpyAsync(io + o_u1, u1, n * 32, hipMemcpyHostToDevice, st));
  else
    HIP_TRY(ctx, hipMemsetAsync(io + o_u1, 0, n * 32, st));   // plain scalar multiplication: u1 = 0
  HIP_TRY(ctx, hipMemcpyAsync(io + o_u2, u2, n * 32, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(io + o_pts, points, n * 65, hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemsetAsync(io + o_status, 0, 4, st));
  const uint8_t* d_u1 = io + o_u1;
  const size_t stride = lane_stride(n);
  uint32_t* ws = (uint32_t*)ctx->ws;
  uint32_t* qt = ws + WS_QT * stride;
  uint32_t* fin = ws + WS_FIN * stride;
  uint32_t* prep = ws + WS_PREP * stride;
  uint32_t* wl_count = ws + WS_LANE_WORDS * stride;
  uint32_t* wl = wl_count + 64;
  uint32_t* status = (uint32_t*)(io + o_status);
  if (impl == S2K_IMPL_COMPLETE) {
    k_point_fallback<true><<<blocks_for(n), 256, 0, st>>>(wl_count, wl, (uint32_t)n, d_u1, io + o_u2, io + o_pts, io + o_out,
                                                          ctx->gt_call, qt, stride, status);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    HIP_TRY(ctx, hipMemsetAsync(wl_count, 0, sizeof(uint32_t), st));
    HIP_TRY(ctx, hipMemsetAsync(io + o_out, 0, n * 65, st));   // (the ladder kernel leaves the records of undecided lanes alone)
    k_hot_prep<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, d_u1, io + o_u2, io + o_pts, io + o_pub, prep, stride, wl_count, wl,
                                              status);
    HIP_TRY(ctx, hipGetLastError());
    k_verify_fast<MODE_POINT><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, io + o_pub, nullptr, prep, qt, fin, ctx->gt_call,
                                                             io + o_ok, wl_count, wl, stride, io + o_out, nullptr, key_groups{});
    HIP_TRY(ctx, hipGetLastError());
    const uint32_t T = (uint32_t)((n + FIN_M - 1) / FIN_M);
    k_affine_finish<MODE_RECOVER><<<(T + 63) / 64, 64, 0, st>>>((uint32_t)n, T, nullptr, fin, io + o_ok, stride, io + o_out);
    HIP_TRY(ctx, hipGetLastError());
    k_point_fallback<false><<<fallback_blocks(ctx, n), 256, 0, st>>>(wl_count, wl, (uint32_t)n, d_u1, io + o_u2, io + o_pts,
                                                                     io + o_out, ctx->gt_call, qt, stride, status);
    HIP_TRY(ctx, hipGetLastError());
  }
  uint32_t h_status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(out, io + o_out, n * 65, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipMemcpyAsync(&h_status, status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  if (h_status) return fail(ctx, S2K_ERR_ARG, "malformed point record (not an encoding the reference's Point can hold)");
  return S2K_OK;
}

}  // extern "C"

// ops.hip — the Point / Scalar / Element entry points (the Go types' methods need
// something to call), batched: scalar multiplications on the complete path, Point.Add / Double on
// the 9x29 complete formulas of the multiscalar kernels, SEC1 decoding, field and scalar
// arithmetic, the valid-bitmap packing of the shard exchange, generator-table inspection.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "complete_path.h"
#include "engine_internal.h"
#include "fe29_inv.h"
#include "jacobian29.h"
#include "pt29.h"
#include "pt29q.h"
#include "fe29r.h"
#include "xyzz29.h"
#include "sc26.h"

using namespace s2k;

extern "C" __attribute__((visibility("hidden"))) int s2k_internal_ensure_ws(s2k_ctx* ctx, size_t n);   // engine.hip: the per-lane workspace (table region first)

// ---------------------------------------------------------------------------------------
// Element-wise kernels behind the Point / Scalar / Element entry points
// ---------------------------------------------------------------------------------------
// 65-byte record -> projective point.  Returns false (and the identity) for malformed records.
S2K_DEV bool point_record_load(pt& p, const uint8_t* rec) {
  p = pt_identity();
  if (rec[0] == 0x00) return true;
  if (rec[0] != 0x04) return false;
  apt a;
  load_be32_unaligned(a.x.v, rec + 1);
  load_be32_unaligned(a.y.v, rec + 33);
  if (!fe_is_canonical_raw(a.x.v) || !fe_is_canonical_raw(a.y.v) || !apt_on_curve(a)) return false;
  p = pt_from_affine(a);
  return true;
}
S2K_DEV void point_record_store(uint8_t* rec, const pt& p) {
  apt a;
  bool finite = pt_to_affine(a, p);
  if (!finite) {
    for (int i = 0; i < 65; ++i) rec[i] = 0;
    return;
  }
  rec[0] = 0x04;
  store_be32_unaligned(rec + 1, a.x.v);
  store_be32_unaligned(rec + 33, a.y.v);
}

enum { PK_BASE_MUL = 0, PK_ADD = 3, PK_DOUBLE = 4 };

S2K_DEV pt29 pt29_from_pt(const pt& p) {
  pt29 r;
  r.x = fe29_from_words(p.x.v);
  r.y = fe29_from_words(p.y.v);
  r.z = fe29_from_words(p.z.v);
  return r;
}
S2K_DEV pt pt_from_pt29(const pt29& p) {
  pt r;
  fe29_to_words(r.x.v, fe29_normalize(p.x));
  fe29_to_words(r.y.v, fe29_normalize(p.y));
  fe29_to_words(r.z.v, fe29_normalize(p.z));
  return r;
}

// scalar multiplications with a per-item point live in engine.hip (s2k_double_scalar_mult_basepoint_batch_ex)
__global__ void __launch_bounds__(256)
k_point_op(int op, uint32_t n, const uint8_t* __restrict__ k1, const uint8_t* __restrict__ pa,
           const uint8_t* __restrict__ pb, uint8_t* __restrict__ out, gt_view gt,
           uint32_t* __restrict__ status) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  pt res = pt_identity();
  bool wellformed = true;
  if (op == PK_BASE_MUL) {
    uint32_t raw[8];
    load_be32(raw, k1 + idx * 32);
    sc k = sc_reduce_once(raw);
    res = pt_base_mul(gt, k.v);
  } else if (op == PK_ADD) {
    // Point.Add (point.go:62) through the 9x29 complete formulas the multiscalar kernels use
    pt a, b;
    wellformed = point_record_load(a, pa + idx * 65);
    wellformed = point_record_load(b, pb + idx * 65) && wellformed;
    res = pt_from_pt29(pt29_add(pt29_from_pt(a), pt29_from_pt(b)));
  } else if (op == PK_DOUBLE) {
    pt a;
    wellformed = point_record_load(a, pa + idx * 65);
    res = pt_from_pt29(pt29_double(pt29_from_pt(a)));
  }
  if (!wellformed) atomicOr(status, 1u);   // not a Point the reference could hold (SetUncompressedBytes fails, point_s11n.go:178)
  point_record_store(out + idx * 65, res);
}

__global__ void __launch_bounds__(256)
k_point_decode(uint32_t n, uint32_t enc_len, const uint8_t* __restrict__ enc, uint8_t* __restrict__ out,
               uint8_t* __restrict__ okv) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const uint8_t* e = enc + idx * enc_len;
  uint8_t* o = out + idx * 65;
  bool ok = false;
  apt a;
  a.x = fe_zero();
  a.y = fe_zero();
  if (enc_len == 33) {
    // SetCompressedBytes (point_s11n.go:140-172)
    uint8_t tag = e[0];
    load_be32_unaligned(a.x.v, e + 1);
    ok = (tag == 0x02 || tag == 0x03) && fe_is_canonical_raw(a.x.v);
    fe y;
    bool has = fe_sqrt(y, fe_curve_rhs(a.x));
    ok = ok && has;
    y = fe_normalize(y);
    bool flip = ((y.v[0] & 1u) != (uint32_t)(tag & 1));
    a.y = fe_normalize(fe_select(flip, y, fe_neg(y)));
  } else {
    // SetUncompressedBytes (point_s11n.go:178-209)
    load_be32_unaligned(a.x.v, e + 1);
    load_be32_unaligned(a.y.v, e + 33);
    ok = e[0] == 0x04 && fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v) && apt_on_curve(a);
  }
  okv[idx] = ok ? 1 : 0;
  if (ok) {
    o[0] = 0x04;
    store_be32_unaligned(o + 1, a.x.v);
    store_be32_unaligned(o + 33, a.y.v);
  } else {
    for (int i = 0; i < 65; ++i) o[i] = 0;
  }
}

__global__ void __launch_bounds__(256)
k_fp_op(int op, uint32_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
        uint8_t* __restrict__ out, uint8_t* __restrict__ flag) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  fe x, y = fe_zero(), r;
  load_be32(x.v, a + idx * 32);
  if (b) load_be32(y.v, b + idx * 32);
  uint8_t f = 1;
  switch (op) {
    case S2K_OP_MUL: r = fe_mul(x, y); break;
    case S2K_OP_SQR: r = fe_sqr(x); break;
    case S2K_OP_ADD: r = fe_add(x, y); break;
    case S2K_OP_SUB: r = fe_sub(x, y); break;
    case S2K_OP_NEG: r = fe_neg(x); break;
    case S2K_OP_INV: r = fe_inv(x); break;
    default: f = fe_sqrt(r, x) ? 1 : 0; break;
  }
  r = fe_normalize(r);
  store_be32(out + idx * 32, r.v);
  if (flag) flag[idx] = f;
}

__global__ void __launch_bounds__(256)
k_fn_op(int op, uint32_t n, const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
        uint8_t* __restrict__ out, uint8_t* __restrict__ out2) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  uint32_t raw[8];
  load_be32(raw, a + idx * 32);
  sc x = sc_reduce_once(raw), y = sc_zero(), r = sc_zero();
  if (b) {
    load_be32(raw, b + idx * 32);
    y = sc_reduce_once(raw);
  }
  sc26 one26;   // plain 1: multiplying by it leaves the Montgomery domain
#pragma unroll
  for (int i = 0; i < 10; ++i) one26.n[i] = i == 0 ? 1u : 0u;
  switch (op) {
    case S2K_OP_MUL: r = sc26_to_sc(sc26_mm(sc26_from_sc(x), sc26_to_mont(sc26_from_sc(y)))); break;
    case S2K_OP_SQR: r = sc26_to_sc(sc26_mm(one26, sc26_montsqr(sc26_to_mont(sc26_from_sc(x))))); break;
    case S2K_OP_ADD: r = sc_add(x, y); break;
    case S2K_OP_SUB: r = sc_add(x, sc_neg(y)); break;
    case S2K_OP_NEG: r = sc_neg(x); break;
    case S2K_OP_INV: r = sc26_to_sc(sc26_mm(sc26_mont_inv(sc26_to_mont(sc26_from_sc(x))), one26)); break;
    default: {   // GLV split, un-normalised (k1, k2 canonical mod n)
      sc k1, k2;
      bool n1, n2;
      sc_split_glv(x, k1, n1, k2, n2);
      r = n1 ? sc_neg(k1) : k1;
      sc r2 = n2 ? sc_neg(k2) : k2;
      store_be32(out2 + idx * 32, r2.v);
      break;
    }
  }
  store_be32(out + idx * 32, r.v);
}

// ---------------------------------------------------------------------------------------
// Test access to the arithmetic the verification ladder and the multiscalar kernels actually run
// (s2k_fp_op_batch_ex, S2K_IMPL_FAST): the 9x29 lazy field incl. its fused products, the Jacobian
// doubling / mixed addition of jacobian29.h and the complete formulas of pt29.h, on operands in
// LAZY form — same value, different unreduced limbs (the counterpart of the reference tests'
// "random Z" re-randomisation, point_test.go:359-373).  Lazy code per operand (4 bits each, operand
// j in bits 4j..4j+3): bits 1:0 = multiples of p added limb by limb, bit 2 = borrow-spread (limb
// i + 2^29, limb i+1 - 1 wherever limb i+1 > 0).  The caller keeps the unit budget of fe29.h.
// ---------------------------------------------------------------------------------------
S2K_DEV fe29 fe29_lazy_form(fe29 v, uint32_t code) {
  if (code & 4u) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (v.n[i + 1] > 0) {
        v.n[i + 1] -= 1u;
        v.n[i] += 1u << 29;
      }
    }
  }
  const uint32_t k = code & 3u;
  v.n[0] += k * F29_P0;
  v.n[1] += k * F29_P1;
#pragma unroll
  for (int i = 2; i < 8; ++i) v.n[i] += k * F29_PM;
  v.n[8] += k * F29_P8;
  return v;
}
S2K_DEV uint32_t lazy_units(uint32_t code) { return 1u + (code & 3u) + ((code & 4u) ? 1u : 0u); }

struct hp_args {
  const uint8_t* in[5];
};

__global__ void __launch_bounds__(256)
k_fp29_op(int op, uint32_t lazy, uint32_t n, hp_args args, uint8_t* __restrict__ out, uint8_t* __restrict__ out2,
          uint8_t* __restrict__ flag) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  fe29 v[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    v[j] = fe29_zero();
    if (args.in[j]) {
      uint32_t w[8];
      load_be32(w, args.in[j] + idx * 32);
      v[j] = fe29_lazy_form(fe29_from_words(w), (lazy >> (4 * j)) & 15u);
    }
  }
  const fe29 &a = v[0], &b = v[1], &c = v[2], &d = v[3], &e = v[4];
  fe29 r = fe29_zero(), r2 = fe29_zero();
  uint8_t f = 1;
  switch (op) {
    case S2K_HP_MUL: r = fe29_mul(a, b); break;
    case S2K_HP_SQR: r = fe29_sqr(a); break;
    case S2K_HP_MUL_PLUS: r = fe29_mul_plus(a, b, c); break;
    case S2K_HP_SQR_PLUS: r = fe29_sqr_plus(a, b); break;
    case S2K_HP_MUL_ADD_MUL: r = fe29_mul_add_mul(a, b, c, d); break;
    case S2K_HP_MUL_ADD_SQR: r = fe29_mul_add_sqr(a, b, c); break;
    case S2K_HP_ADD: r = fe29_add(a, b); break;
    case S2K_HP_NEGATE: r = fe29_negate(a, lazy_units(lazy & 15u)); break;
    case S2K_HP_HALF: r = fe29_half(a); break;
    case S2K_HP_NORMALIZE: r = a; f = fe29_is_zero(a) ? 1 : 0; break;
    case S2K_HP_COND_NEGATE1: r = fe29_cond_negate1(a, (b.n[0] & 1u) != 0); break;
    case S2K_HP_INV: r = fe29_inv(a); break;
    case S2K_HP_INV_GCD: r = fe29_inv_gcd(a); break;
    case S2K_HP_SQRT: f = fe29_sqrt(r, a) ? 1 : 0; if (!f) r = fe29_zero(); break;
    case S2K_HP_EQ: f = fe29_eq(a, b) ? 1 : 0; break;
    case S2K_HP_MUL_SMALL21: r = fe29_mul_small_norm(a, 21); break;
    case S2K_HP_NORMALIZE_WEAK: r = fe29_normalize_weak(a); break;
    case S2K_HP_JDBL:
    case S2K_HP_JADD_FULL:
    case S2K_HP_JADD: {
      // P = (a, b) affine, lifted to Jacobian with Z = c (any non-zero value): X = a c^2, Y = b c^3
      // (lazy code of c with bit 3 set: no lift, P = (a, b, 1) with a, b in their lazy forms: x [1], y [<= 2])
      jpt29 p;
      fe29 zz = fe29_sqr(c);
      p.x = fe29_mul(a, zz);
      p.y = fe29_mul(b, fe29_mul(zz, c));
      p.z = fe29_normalize_weak(c);
      if ((lazy >> 8) & 8u) {
        p.x = a;
        p.y = b;
        p.z = fe29_one();
      }
      jpt29 q;
      if (op == S2K_HP_JADD_FULL) {
        jpt29 t;
        fe29 z2 = fe29_sqr(c), z4 = fe29_sqr(z2);
        t.x = fe29_mul(d, z4);
        t.y = fe29_mul(e, fe29_mul(z4, z2));
        t.z = z2;
        q = jpt29_add(p, t);
      } else {
        q = op == S2K_HP_JDBL ? jpt29_double(p) : jpt29_add_affine(p, d, e);
      }
      if (fe29_is_zero(q.z)) {   // exceptional input of the incomplete formulas (or a true infinity)
        f = 0;
      } else {
        fe29 zi = fe29_inv(q.z), zi2 = fe29_sqr(zi);
        r = fe29_mul(q.x, zi2);
        r2 = fe29_mul(fe29_mul(q.y, zi2), zi);
      }
      break;
    }
    case S2K_HP_XYZZ_ADD: {
      // P = (a, b) lifted to XYZZ with ZZ = c^2, ZZZ = c^3 (c non-zero): X = a c^2, Y = b c^3; Q = (d, e) affine; the result
      // goes back through the projective form the bucket pass stores (flag = 0: ZZ3 = 0, an exceptional input)
      xyzz29 p;
      const fe29 c2 = fe29_sqr(c), c3 = fe29_mul(c2, c);
      p.x = fe29_mul(a, c2);
      p.y = fe29_mul(b, c3);
      p.zz = c2;
      p.zzz = c3;
      const pt29 q = xyzz29_to_pt29(xyzz29_add_affine(p, fe29_normalize_weak(d), fe29_normalize_weak(e)));
      if (fe29_is_zero(q.z)) {
        f = 0;
      } else {
        fe29 zi = fe29_inv(q.z);
        r = fe29_mul(q.x, zi);
        r2 = fe29_mul(q.y, zi);
      }
      break;
    }
    case S2K_HP_XYZZ_ROUND: {
      xyzz29 xa;
      const fe29 c2 = fe29_sqr(c), c3 = fe29_mul(c2, c);
      xa.x = fe29_mul(a, c2);
      xa.y = fe29_mul(b, c3);
      xa.zz = c2;
      xa.zzz = c3;
      const fe29 qx = fe29_normalize_weak(d), qy = fe29_normalize_weak(e);
      xa = xyzz29_add_affine(xa, qx, qy);
      jpt29 j = xyzz29_to_jacobian(xa);
      j = jpt29_double(jpt29_double(j));
      xa = xyzz29_from_jacobian(j);
      xa = xyzz29_add_affine(xa, qx, fe29_cond_negate1(qy, false));     // (y in the form the ladder passes it)
      j = xyzz29_to_jacobian(xa);
      if (fe29_is_zero(j.z)) {
        f = 0;
      } else {
        const fe29 zi = fe29_inv(j.z), zi2 = fe29_sqr(zi);
        r = fe29_mul(j.x, zi2);
        r2 = fe29_mul(fe29_mul(j.y, zi2), zi);
      }
      break;
    }
    case S2K_HP_PT29_DBL:
    case S2K_HP_PT29_ADD:
    case S2K_HP_PT29_ADD_MIXED: {
      // P = (a : b : 1) scaled by Z = c (c == 0 on input: the identity (0 : 1 : 0)); Q = (d, e) affine,
      // for PT29_ADD rescaled by the same c
      pt29 p, q;
      bool p_inf = fe29_is_zero(c);
      fe29 cn = fe29_normalize_weak(c);
      p.x = fe29_mul(a, cn);
      p.y = fe29_mul(b, cn);
      p.z = cn;
      if (p_inf) p = pt29_identity();
      if (op == S2K_HP_PT29_DBL) {
        q = pt29_double(p);
      } else if (op == S2K_HP_PT29_ADD_MIXED) {
        q = pt29_add_mixed(p, d, e);
      } else {
        pt29 t;
        t.x = fe29_mul(d, cn);
        t.y = fe29_mul(e, cn);
        t.z = cn;
        if (p_inf) {     // c == 0: add the affine point itself, Z = 1
          t.x = d;
          t.y = e;
          t.z = fe29_one();
        }
        q = pt29_add(p, t);
      }
      if (fe29_is_zero(q.z)) {
        f = 0;          // the identity
      } else {
        fe29 zi = fe29_inv(q.z);
        r = fe29_mul(q.x, zi);
        r2 = fe29_mul(q.y, zi);
      }
      break;
    }
    default: f = 0; break;
  }
  uint32_t w[8];
  fe29_to_words(w, fe29_normalize(r));
  store_be32(out + idx * 32, w);
  if (out2) {
    fe29_to_words(w, fe29_normalize(r2));
    store_be32(out2 + idx * 32, w);
  }
  if (flag) flag[idx] = f;
}

// The quad-spread group law of pt29q.h, four lanes per item: inputs as for PT29_DBL / PT29_ADD above (P = (a : b : 1)
// scaled by Z = c, c == 0: the identity; Q = (d, e) scaled by the same c, or affine when P is the identity); `reps`
// > 1 chains the operation (r = r + Q / r = 2 r again and again) so that outputs are fed back as inputs.
__global__ void __launch_bounds__(256)
k_pt29q_op(int op, uint32_t lazy, uint32_t n, uint32_t reps, hp_args args, uint8_t* __restrict__ out, uint8_t* __restrict__ out2,
           uint8_t* __restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t idx = t >> 2;
  const uint32_t q = (uint32_t)t & 3u;
  if (idx >= n) return;
  fe29 v[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    v[j] = fe29_zero();
    if (args.in[j]) {
      uint32_t w[8];
      load_be32(w, args.in[j] + idx * 32);
      v[j] = fe29_lazy_form(fe29_from_words(w), (lazy >> (4 * j)) & 15u);
    }
  }
  const fe29 &a = v[0], &b = v[1], &c = v[2], &d = v[3], &e = v[4];
  pt29 p, s;
  const bool p_inf = fe29_is_zero(c);
  const fe29 cn = fe29_normalize_weak(c);
  p.x = fe29_mul(a, cn);
  p.y = fe29_mul(b, cn);
  p.z = cn;
  s.x = fe29_mul(d, cn);
  s.y = fe29_mul(e, cn);
  s.z = cn;
  if (p_inf) {
    p = pt29_identity();
    s.x = fe29_normalize_weak(d);
    s.y = fe29_normalize_weak(e);
    s.z = fe29_one();
  }
  fe29 rc = pt29q_from(p, q);
  const fe29 sc_ = pt29q_from(s, q);
#pragma unroll 1
  for (uint32_t i = 0; i < reps; ++i) rc = op == S2K_HP_PT29Q_DBL ? pt29q_double(rc, q) : pt29q_add(rc, sc_, q);
  const pt29 r = pt29q_gather(rc);
  if (q != 0) return;
  fe29 x = fe29_zero(), y = fe29_zero();
  uint8_t f = 1;
  if (fe29_is_zero(r.z)) {
    f = 0;
  } else {
    fe29 zi = fe29_inv(r.z);
    x = fe29_mul(r.x, zi);
    y = fe29_mul(r.y, zi);
  }
  uint32_t w[8];
  fe29_to_words(w, fe29_normalize(x));
  store_be32(out + idx * 32, w);
  if (out2) {
    fe29_to_words(w, fe29_normalize(y));
    store_be32(out2 + idx * 32, w);
  }
  if (flag) flag[idx] = f;
}

// The row-spread field and group law of fe29r.h, ONE WAVE PER ITEM (the item's four rows all hold it; for the field products
// the four rows multiply the operands in four different lazy forms and must agree).  Inputs as for k_pt29q_op.
//   FER_MUL a*b | FER_MUL_PLUS a*b+c | FER_MUL_ADD_MUL a*b+c*d | FER_SMALL 21 a  (out = row 0's result, flag = the rows agree)
//   PT29R_DBL / PT29R_ADD: chained `reps` times like the quad forms
//   FER_SWAPS: out[0..255] = what v_permlane16_swap / v_permlane32_swap make of the lane numbers (n >= 8)
__global__ void __launch_bounds__(256)
k_pt29r_op(int op, uint32_t lazy, uint32_t n, uint32_t reps, hp_args args, uint8_t* __restrict__ out, uint8_t* __restrict__ out2,
           uint8_t* __restrict__ flag) {
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t idx = t >> 6;
  const uint32_t lane = (uint32_t)t & 63u;
  if (idx >= n) return;                       // (whole waves leave together)
  const fer_consts k = fer_setup(lane);
  if (op == S2K_HP_FER_SWAPS) {
    if (idx != 0) return;
    fer e, o, lo, hi;
    fer_pairs(lane, e, o);
    fer_halves(lane, lo, hi);
    out[lane] = (uint8_t)e;
    out[64 + lane] = (uint8_t)o;
    out[128 + lane] = (uint8_t)lo;
    out[192 + lane] = (uint8_t)hi;
    return;
  }
  fe29 v[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    v[j] = fe29_zero();
    if (args.in[j]) {
      uint32_t w[8];
      load_be32(w, args.in[j] + idx * 32);
      v[j] = fe29_from_words(w);
    }
  }
  if (op == S2K_HP_FER_MUL || op == S2K_HP_FER_MUL_PLUS || op == S2K_HP_FER_MUL_ADD_MUL || op == S2K_HP_FER_SMALL) {
    // Row r multiplies the operands in its own lazy forms (codes of fe29_lazy_form: bits 1:0 multiples of p added, bit 2
    // borrow-spread; units = 1 + both), chosen so that every row stays inside the budget of a reduction - units multiplied
    // and summed over the terms at most 7: a, b | a, b, addend c | a, b, c, d.  `lazy` & 1 swaps the roles of the factors.
    constexpr uint32_t PAT[4][4] = {{0, 0, 0, 0}, {1, 0, 0, 1}, {4, 1, 0, 0}, {2, 0, 4, 0}};   // [row][operand]: 2 | 2+2 | 4+1 | 3+2
    constexpr uint32_t PAT_PLUS[4] = {0, 2, 1, 0};                                             // addend c with a, b as above: 1+1 | 2+3 | 4+2 | 3+1
    fer f[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      uint32_t code = j < 4 ? PAT[k.row][(lazy & 1u) ? (j ^ 1) : j] : 0u;
      if (op == S2K_HP_FER_MUL_PLUS && j == 2) code = PAT_PLUS[k.row];
      if (op == S2K_HP_FER_SMALL) code = j == 0 ? (k.row == 3 ? 6u : PAT[k.row][0]) : 0u;       // a small multiple takes up to 4 units
      f[j] = fer_from_fe29(fe29_lazy_form(fe29_normalize_weak(v[j]), code), k);
    }
    fer r;
    if (op == S2K_HP_FER_MUL) r = fer_mul(f[0], f[1], k);
    else if (op == S2K_HP_FER_MUL_PLUS) r = fer_mul_plus(f[0], f[1], f[2], k);
    else if (op == S2K_HP_FER_MUL_ADD_MUL) r = fer_mul_add_mul(f[0], f[1], f[2], f[3], k);
    else r = fer_small_norm(f[0], 21u, k);
    const fe29 rr = fe29_normalize(fer_to_fe29(r));        // this row's result, canonical, in every lane of the row
    uint32_t agree = 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) agree &= (uint32_t)(__builtin_amdgcn_readlane((int)rr.n[i], 0) == __builtin_amdgcn_readlane((int)rr.n[i], 16) &&
                                                    __builtin_amdgcn_readlane((int)rr.n[i], 0) == __builtin_amdgcn_readlane((int)rr.n[i], 32) &&
                                                    __builtin_amdgcn_readlane((int)rr.n[i], 0) == __builtin_amdgcn_readlane((int)rr.n[i], 48));
    if (lane != 0) return;
    uint32_t w[8];
    fe29_to_words(w, rr);
    store_be32(out + idx * 32, w);
    if (flag) flag[idx] = (uint8_t)agree;
    return;
  }
  const fe29 &a = v[0], &b = v[1], &c = v[2], &d = v[3], &e = v[4];
  pt29 p, s;
  const bool p_inf = fe29_is_zero(c);
  const fe29 cn = fe29_normalize_weak(c);
  p.x = fe29_mul(a, cn);
  p.y = fe29_mul(b, cn);
  p.z = cn;
  s.x = fe29_mul(d, cn);
  s.y = fe29_mul(e, cn);
  s.z = cn;
  if (p_inf) {
    p = pt29_identity();
    s.x = fe29_normalize_weak(d);
    s.y = fe29_normalize_weak(e);
    s.z = fe29_one();
  }
  p.y = fe29_lazy_form(p.y, lazy & 1u);        // y may come with two units
  pt29r rc = pt29r_from(p, k);
  const pt29r sc_ = pt29r_from(s, k);
#pragma unroll 1
  for (uint32_t i = 0; i < reps; ++i) rc = op == S2K_HP_PT29R_DBL ? pt29r_double(rc, k) : pt29r_add(rc, sc_, k);
  const pt29 r = pt29r_gather(rc, k);
  // the four rows must hold the same point
  uint32_t agree = 1;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const fe29* cs[3] = {&r.x, &r.y, &r.z};
#pragma unroll
    for (int cc = 0; cc < 3; ++cc) {
      const int v0 = __builtin_amdgcn_readlane((int)cs[cc]->n[i], 0);
      agree &= (uint32_t)(v0 == __builtin_amdgcn_readlane((int)cs[cc]->n[i], 16) && v0 == __builtin_amdgcn_readlane((int)cs[cc]->n[i], 32) &&
                          v0 == __builtin_amdgcn_readlane((int)cs[cc]->n[i], 48));
    }
  }
  if (lane != 0) return;
  fe29 x = fe29_zero(), y = fe29_zero();
  uint8_t f = 1;
  if (fe29_is_zero(r.z)) {
    f = 0;
  } else {
    fe29 zi = fe29_inv(r.z);
    x = fe29_mul(r.x, zi);
    y = fe29_mul(r.y, zi);
  }
  if (!agree) f = 2;
  uint32_t w[8];
  fe29_to_words(w, fe29_normalize(x));
  store_be32(out + idx * 32, w);
  if (out2) {
    fe29_to_words(w, fe29_normalize(y));
    store_be32(out2 + idx * 32, w);
  }
  if (flag) flag[idx] = f;
}

// odd GLV split of the hot path (sc_split_glv_odd): magnitudes (129 bits) and sign bits
__global__ void __launch_bounds__(256)
k_split_glv_odd(uint32_t n, const uint8_t* __restrict__ k, uint8_t* __restrict__ k1o, uint8_t* __restrict__ k2o,
                uint8_t* __restrict__ signs) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  uint32_t raw[8];
  load_be32(raw, k + idx * 32);
  sc k1, k2;
  bool n1, n2;
  sc_split_glv_odd(sc_reduce_once(raw), k1, n1, k2, n2);
  store_be32(k1o + idx * 32, k1.v);
  store_be32(k2o + idx * 32, k2.v);
  signs[idx] = (n1 ? 1 : 0) | (n2 ? 2 : 0);
}

// valid bytes -> bitmap (bit i of byte i/8, LSB first) + number of valid items.  One lane per 32 items (four bitmap
// bytes); the count is summed per wave, then per workgroup in LDS, and ONE atomic per workgroup reaches the counter:
// atomics on one address cost about 12 ns each on MI355X, and one per wave (2048 for 2^20 items) made this kernel 27 us.
constexpr int PACK_BYTES = 4;   // bitmap bytes per lane
__global__ void __launch_bounds__(256)
k_pack_valid(uint32_t n, const uint8_t* __restrict__ valid, uint8_t* __restrict__ bitmap,
             unsigned long long* __restrict__ count) {
  __shared__ uint32_t wsum[4];
  const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t c = 0;
#pragma unroll
  for (int g = 0; g < PACK_BYTES; ++g) {
    const size_t byte = t * PACK_BYTES + g, base = byte * 8;
    uint32_t bits = 0;
    if (base + 8 <= n) {
      uint2 v = *reinterpret_cast<const uint2*>(valid + base);
      uint32_t lo = v.x & 0x01010101u, hi = v.y & 0x01010101u;
      // gather bit 0 of each byte: multiply trick
      bits = ((lo * 0x10204080u) >> 28) | (((hi * 0x10204080u) >> 28) << 4);
    } else if (base < n) {
      for (uint32_t j = 0; base + j < n; ++j) bits |= (uint32_t)(valid[base + j] & 1u) << j;
    }
    if (base < n) bitmap[byte] = (uint8_t)bits;
    c += __popc(bits);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    if (s) atomicAdd(count, (unsigned long long)s);
  }
}

__global__ void k_gtable_entry(gt_view gt, uint32_t window, uint32_t digit, uint8_t* out64) {
  apt a = gt_load(gt, window, digit);
  store_be32(out64, a.x.v);
  store_be32(out64 + 32, a.y.v);
}

extern "C" {

static int point_op(s2k_ctx* ctx, int op, size_t n, const uint8_t* k1, const uint8_t* pa, const uint8_t* pb, uint8_t* out) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!out) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());   // these entry points run on the NULL stream with their own buffers
  dev_buf dk1, dpa, dpb, dout, dst;
  if (k1) HIP_TRY(ctx, dk1.upload(k1, n * 32));
  if (pa) HIP_TRY(ctx, dpa.upload(pa, n * 65));
  if (pb) HIP_TRY(ctx, dpb.upload(pb, n * 65));
  HIP_TRY(ctx, dout.alloc(n * 65));
  HIP_TRY(ctx, dst.alloc(16));
  HIP_TRY(ctx, hipMemset(dst.p, 0, 16));
  k_point_op<<<blocks_for(n), 256>>>(op, (uint32_t)n, (const uint8_t*)dk1.p, (const uint8_t*)dpa.p, (const uint8_t*)dpb.p,
                                     (uint8_t*)dout.p, s2k_internal_gt_load(ctx), (uint32_t*)dst.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  uint32_t h_status = 0;
  HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 65, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(&h_status, dst.p, 4, hipMemcpyDeviceToHost));
  if (h_status) return fail(ctx, S2K_ERR_ARG, "malformed point record (not an encoding the reference's Point can hold)");
  return S2K_OK;
}

int s2k_scalar_base_mult_batch(s2k_ctx* ctx, size_t n, const uint8_t* k, uint8_t* out) {
  if (n && !k) return fail(ctx, S2K_ERR_ARG, "null scalar buffer");
  return point_op(ctx, PK_BASE_MUL, n, k, nullptr, nullptr, out);
}
int s2k_scalar_mult_batch(s2k_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* points, uint8_t* out) {
  if (n && (!k || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return s2k_double_scalar_mult_basepoint_batch_ex(ctx, S2K_IMPL_FAST, n, nullptr, k, points, out);
}
int s2k_double_scalar_mult_basepoint_batch(s2k_ctx* ctx, size_t n, const uint8_t* u1, const uint8_t* u2,
                                           const uint8_t* points, uint8_t* out) {
  if (n && (!u1 || !u2 || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return s2k_double_scalar_mult_basepoint_batch_ex(ctx, S2K_IMPL_FAST, n, u1, u2, points, out);
}
int s2k_point_add_batch(s2k_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (n && (!a || !b)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_ADD, n, nullptr, a, b, out);
}
int s2k_point_double_batch(s2k_ctx* ctx, size_t n, const uint8_t* a, uint8_t* out) {
  if (n && !a) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_DOUBLE, n, nullptr, a, nullptr, out);
}

int s2k_point_decode_batch(s2k_ctx* ctx, size_t n, size_t enc_len, const uint8_t* enc, uint8_t* out, uint8_t* ok) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (enc_len != 33 && enc_len != 65) return fail(ctx, S2K_ERR_ARG, "enc_len must be 33 or 65");
  if (n == 0) return S2K_OK;
  if (!enc || !out || !ok) return fail(ctx, S2K_ERR_ARG, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf de, dout, dok;
  HIP_TRY(ctx, de.upload(enc, n * enc_len));
  HIP_TRY(ctx, dout.alloc(n * 65));
  HIP_TRY(ctx, dok.alloc(n));
  k_point_decode<<<blocks_for(n), 256>>>((uint32_t)n, (uint32_t)enc_len, (const uint8_t*)de.p, (uint8_t*)dout.p,
                                         (uint8_t*)dok.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 65, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(ok, dok.p, n, hipMemcpyDeviceToHost));
  return S2K_OK;
}

static int field_op(s2k_ctx* ctx, bool is_fp, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out,
                    uint8_t* out2, size_t out2_bytes) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!a || !out) return fail(ctx, S2K_ERR_ARG, "null buffer");
  bool binary = (op == S2K_OP_MUL || op == S2K_OP_ADD || op == S2K_OP_SUB);
  if (binary && !b) return fail(ctx, S2K_ERR_ARG, "binary op needs b");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf da, db, dout, dout2;
  HIP_TRY(ctx, da.upload(a, n * 32));
  if (binary) HIP_TRY(ctx, db.upload(b, n * 32));
  HIP_TRY(ctx, dout.alloc(n * 32));
  if (out2) HIP_TRY(ctx, dout2.alloc(n * out2_bytes));
  if (is_fp)
    k_fp_op<<<blocks_for(n), 256>>>(op, (uint32_t)n, (const uint8_t*)da.p, binary ? (const uint8_t*)db.p : nullptr,
                                    (uint8_t*)dout.p, (uint8_t*)dout2.p);
  else
    k_fn_op<<<blocks_for(n), 256>>>(op, (uint32_t)n, (const uint8_t*)da.p, binary ? (const uint8_t*)db.p : nullptr,
                                    (uint8_t*)dout.p, (uint8_t*)dout2.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 32, hipMemcpyDeviceToHost));
  if (out2) HIP_TRY(ctx, hipMemcpy(out2, dout2.p, n * out2_bytes, hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_fp_op_batch(s2k_ctx* ctx, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* flag) {
  if (op < S2K_OP_MUL || op > S2K_OP_SQRT) return fail(ctx, S2K_ERR_ARG, "bad op");
  return field_op(ctx, true, op, n, a, b, out, flag, 1);
}
int s2k_fn_op_batch(s2k_ctx* ctx, int op, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out, uint8_t* flag) {
  if (op < S2K_OP_MUL || op > S2K_OP_INV) return fail(ctx, S2K_ERR_ARG, "bad op");
  if (flag) memset(flag, 1, n);
  return field_op(ctx, false, op, n, a, b, out, nullptr, 0);
}
int s2k_fn_split_glv_batch(s2k_ctx* ctx, size_t n, const uint8_t* k, uint8_t* k1, uint8_t* k2) {
  if (n && !k2) return fail(ctx, S2K_ERR_ARG, "null buffer");
  return field_op(ctx, false, 100, n, k, nullptr, k1, k2, 32);
}

int s2k_fp_op_batch_ex(s2k_ctx* ctx, uint32_t impl, int op, uint32_t lazy, size_t n, const uint8_t* const in[5],
                       uint8_t* out, uint8_t* out2, uint8_t* flag) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (impl != S2K_IMPL_FAST) return fail(ctx, S2K_ERR_ARG, "s2k_fp_op_batch_ex serves S2K_IMPL_FAST only (8x32: s2k_fp_op_batch)");
  const uint32_t reps = lazy >> 20;           // quad operations: bits 20.. of `lazy` = how often the operation is chained (0: once)
  lazy &= 0xfffffu;
  if (op < 0 || op > S2K_HP_FER_SWAPS) return fail(ctx, S2K_ERR_ARG, "bad op");
  if (op == S2K_HP_FER_SWAPS && n < 8) return fail(ctx, S2K_ERR_ARG, "S2K_HP_FER_SWAPS writes 256 bytes: n >= 8");
  if (n == 0) return S2K_OK;
  if (!in || !in[0] || !out) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  dev_buf din[5], dout, dout2, dflag;
  hp_args args;
  for (int j = 0; j < 5; ++j) {
    args.in[j] = nullptr;
    if (in[j]) {
      HIP_TRY(ctx, din[j].upload(in[j], n * 32));
      args.in[j] = (const uint8_t*)din[j].p;
    }
  }
  HIP_TRY(ctx, dout.alloc(n * 32));
  if (out2) HIP_TRY(ctx, dout2.alloc(n * 32));
  if (flag) HIP_TRY(ctx, dflag.alloc(n));
  if (op >= S2K_HP_FER_MUL && op <= S2K_HP_FER_SWAPS)
    k_pt29r_op<<<blocks_for(64 * n), 256>>>(op, lazy, (uint32_t)n, reps ? reps : 1u, args, (uint8_t*)dout.p, (uint8_t*)dout2.p, (uint8_t*)dflag.p);
  else if (op == S2K_HP_PT29Q_DBL || op == S2K_HP_PT29Q_ADD)
    k_pt29q_op<<<blocks_for(4 * n), 256>>>(op, lazy, (uint32_t)n, reps ? reps : 1u, args, (uint8_t*)dout.p, (uint8_t*)dout2.p, (uint8_t*)dflag.p);
  else
    k_fp29_op<<<blocks_for(n), 256>>>(op, lazy, (uint32_t)n, args, (uint8_t*)dout.p, (uint8_t*)dout2.p, (uint8_t*)dflag.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 32, hipMemcpyDeviceToHost));
  if (out2) HIP_TRY(ctx, hipMemcpy(out2, dout2.p, n * 32, hipMemcpyDeviceToHost));
  if (flag) HIP_TRY(ctx, hipMemcpy(flag, dflag.p, n, hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_fn_split_glv_batch_ex(s2k_ctx* ctx, uint32_t impl, size_t n, const uint8_t* k, uint8_t* k1, uint8_t* k2, uint8_t* signs) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (impl != S2K_IMPL_FAST) return fail(ctx, S2K_ERR_ARG, "s2k_fn_split_glv_batch_ex serves S2K_IMPL_FAST only");
  if (n == 0) return S2K_OK;
  if (!k || !k1 || !k2 || !signs) return fail(ctx, S2K_ERR_ARG, "null buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  dev_buf dk, d1, d2, ds;
  HIP_TRY(ctx, dk.upload(k, n * 32));
  HIP_TRY(ctx, d1.alloc(n * 32));
  HIP_TRY(ctx, d2.alloc(n * 32));
  HIP_TRY(ctx, ds.alloc(n));
  k_split_glv_odd<<<blocks_for(n), 256>>>((uint32_t)n, (const uint8_t*)dk.p, (uint8_t*)d1.p, (uint8_t*)d2.p, (uint8_t*)ds.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(k1, d1.p, n * 32, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(k2, d2.p, n * 32, hipMemcpyDeviceToHost));
  HIP_TRY(ctx, hipMemcpy(signs, ds.p, n, hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_pack_valid_device(s2k_ctx* ctx, size_t n, const void* d_valid, void* d_bitmap, void* d_count,
                          void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!d_count) return fail(ctx, S2K_ERR_ARG, "null count buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 8, st));
  if (n == 0) return S2K_OK;
  if (!d_valid || !d_bitmap) return fail(ctx, S2K_ERR_ARG, "null buffer");
  k_pack_valid<<<blocks_for((n + 8 * PACK_BYTES - 1) / (8 * PACK_BYTES)), 256, 0, st>>>((uint32_t)n, (const uint8_t*)d_valid, (uint8_t*)d_bitmap,
                                                        (unsigned long long*)d_count);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}

// the window width automatic contexts aim for (a build-time constant; what a context uses at a given moment: s2k_ctx_gt_info)
int s2k_generator_window_bits(void) { return GT_BITS_TARGET; }

int s2k_debug_gtable_entry(s2k_ctx* ctx, unsigned i, unsigned d, uint8_t* out64) {
  if (!ctx || !out64) return fail(ctx, S2K_ERR_ARG, "null argument");
  const gt_view gt = s2k_internal_gt_load(ctx);
  if (i >= gt.windows || d >= (1u << gt.bits)) return fail(ctx, S2K_ERR_ARG, "index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf o;
  HIP_TRY(ctx, o.alloc(64));
  k_gtable_entry<<<1, 1>>>(gt, i, d, (uint8_t*)o.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out64, o.p, 64, hipMemcpyDeviceToHost));
  return S2K_OK;
}

}  // extern "C"

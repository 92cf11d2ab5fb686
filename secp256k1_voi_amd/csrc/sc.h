// sc.h — arithmetic modulo the group order n on gfx950, one scalar per lane.
//
// Device replacement for the reference's Scalar (scalar.go:46-206), the fiat Fn code
// (internal/fiat/secp256k1montgomeryscalar/secp256k1montgomeryscalar.go:87) and the GLV
// split (point_mul_glv.go:59-189).  8 x 32-bit limbs, little-endian, always fully reduced
// (0 <= v < n).  n has no special form, so products use word-by-word Montgomery
// reduction with R = 2^256 (the same algorithm fiat generates, on 32-bit words).
// A value is "plain" or "Montgomery" (x*R mod n) as the function comments say;
// sc_montmul(plain, mont) yields a plain product.
#pragma once
#include "fe.h"

namespace s2k {

struct sc {
  uint32_t v[8];
};

__device__ static const uint32_t SC_N[8] = {0xd0364141u, 0xbfd25e8cu, 0xaf48a03bu, 0xbaaedce6u,
                                            0xfffffffeu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
constexpr uint32_t SC_N0INV = 0x5588b13fu;   // -n^-1 mod 2^32 (low word of fiat's 0x4b0dff665588b13f)
__device__ static const uint32_t SC_R2[8] = {0x67d7d140u, 0x896cf214u, 0x0e7cf878u, 0x741496c2u,
                                             0x5bcd07c6u, 0xe697f5e4u, 0x81c69bc5u, 0x9d671cd5u};
__device__ static const uint32_t SC_ONE_M[8] = {0x2fc9bebfu, 0x402da173u, 0x50b75fc4u, 0x45512319u,
                                                0x00000001u, 0, 0, 0};   // R mod n
__device__ static const uint32_t SC_HALF_N[8] = {0x681b20a0u, 0xdfe92f46u, 0x57a4501du, 0x5d576e73u,
                                                 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x7fffffffu};
// g1, g2 of splitGLV (point_mul_glv.go:59-117): round(2^384 b2 / n), round(2^384 (-b1) / n)
__device__ static const uint32_t SC_G1[8] = {0x45dbb031u, 0xe893209au, 0x71e8ca7fu, 0x3daa8a14u,
                                             0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u};
__device__ static const uint32_t SC_G2[8] = {0x8ac47f71u, 0x1571b4aeu, 0x9df506c6u, 0x221208acu,
                                             0x0abfe4c4u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u};

S2K_DEV sc sc_from_limbs(const uint32_t* p) {
  sc r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = p[i];
  return r;
}
S2K_DEV sc sc_zero() {
  sc r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = 0;
  return r;
}
S2K_DEV bool sc_is_zero(const sc& a) { return u256_is_zero(a.v); }                 // scalar.go:181
S2K_DEV bool sc_eq(const sc& a, const sc& b) { return u256_eq(a.v, b.v); }         // scalar.go:176
S2K_DEV bool sc_is_canonical_raw(const uint32_t a[8]) { return u256_lt(a, SC_N); } // scalar.go:136
// IsGreaterThanHalfN on a plain value (scalar.go:190-206)
S2K_DEV bool sc_is_gt_half_n(const sc& a) { return u256_lt(SC_HALF_N, a.v); }

// raw 256-bit value (< 2n) -> reduced (SetBytes, scalar.go:123; reduceSaturated)
S2K_DEV sc sc_reduce_once(const uint32_t a[8]) {
  sc r;
  uint32_t t[8];
  uint32_t borrow = u256_sub(t, a, SC_N);
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = borrow ? a[i] : t[i];
  return r;
}
S2K_DEV sc sc_add(const sc& a, const sc& b) {   // scalar.go:66
  uint32_t s[8], t[8];
  uint32_t carry = u256_add(s, a.v, b.v);
  uint32_t borrow = u256_sub(t, s, SC_N);
  sc r;
  bool use_t = carry || !borrow;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = use_t ? t[i] : s[i];
  return r;
}
S2K_DEV sc sc_neg(const sc& a) {   // scalar.go:78
  uint32_t t[8];
  u256_sub(t, SC_N, a.v);
  bool z = sc_is_zero(a);
  sc r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = z ? 0u : t[i];
  return r;
}

// a * b * R^-1 mod n  (fiat Mul, secp256k1montgomeryscalar.go:87, on 32-bit words)
__device__ __noinline__ sc sc_montmul(sc a, sc b) {
  uint32_t t[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      uint64_t x = (uint64_t)a.v[i] * b.v[j] + t[j] + carry;
      t[j] = (uint32_t)x;
      carry = (uint32_t)(x >> 32);
    }
    uint64_t x = (uint64_t)t[8] + carry;
    t[8] = (uint32_t)x;
    t[9] = (uint32_t)(x >> 32);
    uint32_t q = t[0] * SC_N0INV;
    x = (uint64_t)q * SC_N[0] + t[0];
    carry = (uint32_t)(x >> 32);
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      x = (uint64_t)q * SC_N[j] + t[j] + carry;
      t[j - 1] = (uint32_t)x;
      carry = (uint32_t)(x >> 32);
    }
    x = (uint64_t)t[8] + carry;
    t[7] = (uint32_t)x;
    t[8] = t[9] + (uint32_t)(x >> 32);
  }
  uint32_t d[8];
  uint32_t borrow = u256_sub(d, t, SC_N);
  bool use_d = t[8] != 0 || !borrow;
  sc r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = use_d ? d[i] : t[i];
  return r;
}
S2K_DEV sc sc_to_mont(const sc& a) { return sc_montmul(a, sc_from_limbs(SC_R2)); }   // fiat ToMontgomery
S2K_DEV sc sc_montsqr_n(sc a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = sc_montmul(a, a);
  return a;
}

// x^(n-2) in the Montgomery domain (in: x*R, out: x^-1*R); chain of scalar_invert.go:11-303
// (253 S + 40 M), generic over the limb representation.  Invert(0) = 0.
template <class T, class O>
S2K_DEV T sc_inv_chain(const T& x) {
  T t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14;
  t0 = O::mul(x, x);
  t1 = O::mul(x, t0);
  t2 = O::mul(t0, t1);
  t3 = O::mul(t0, t2);
  t4 = O::mul(t0, t3);
  t5 = O::mul(t0, t4);
  t0 = O::mul(t0, t5);
  t6 = O::sqn(t0, 2);
  t6 = O::mul(t5, t6);
  t7 = O::mul(t6, t6);
  t7 = O::mul(x, t7);
  t8 = O::mul(t7, t7);
  t8 = O::mul(x, t8);
  t9 = O::sqn(t8, 3);
  t10 = O::sqn(t9, 2);
  t11 = O::mul(t10, t10);
  t12 = O::mul(t11, t11);
  t13 = O::sqn(t12, 7);
  t11 = O::mul(t11, t13);
  t11 = O::sqn(t11, 9);
  t12 = O::mul(t12, t11);
  t11 = O::sqn(t12, 6);
  t10 = O::mul(t10, t11);
  t10 = O::sqn(t10, 26);
  t12 = O::mul(t12, t10);
  t10 = O::sqn(t12, 4);
  t9 = O::mul(t9, t10);
  t9 = O::sqn(t9, 60);
  t12 = O::mul(t12, t9);
  t7 = O::mul(t7, t12);
  t7 = O::sqn(t7, 5);
  t7 = O::mul(t5, t7);
  t7 = O::sqn(t7, 3);
  t7 = O::mul(t2, t7);
  t7 = O::sqn(t7, 4);
  t7 = O::mul(t2, t7);
  t7 = O::sqn(t7, 4);
  t7 = O::mul(t3, t7);
  t7 = O::sqn(t7, 5);
  t7 = O::mul(t0, t7);
  t7 = O::sqn(t7, 2);
  t7 = O::mul(t1, t7);
  t7 = O::sqn(t7, 5);
  t7 = O::mul(t3, t7);
  t7 = O::sqn(t7, 6);
  t7 = O::mul(t0, t7);
  t7 = O::sqn(t7, 5);
  t7 = O::mul(t5, t7);
  t7 = O::sqn(t7, 4);
  t7 = O::mul(t0, t7);
  t7 = O::sqn(t7, 3);
  t7 = O::mul(x, t7);
  t7 = O::sqn(t7, 6);
  t2 = O::mul(t2, t7);
  t2 = O::sqn(t2, 10);
  t2 = O::mul(t3, t2);
  t2 = O::sqn(t2, 4);
  t3 = O::mul(t3, t2);
  t3 = O::sqn(t3, 9);
  t8 = O::mul(t8, t3);
  t8 = O::sqn(t8, 5);
  t8 = O::mul(t4, t8);
  t8 = O::sqn(t8, 6);
  t5 = O::mul(t5, t8);
  t5 = O::sqn(t5, 4);
  t5 = O::mul(t0, t5);
  t5 = O::sqn(t5, 5);
  t1 = O::mul(t1, t5);
  t1 = O::sqn(t1, 6);
  t1 = O::mul(t0, t1);
  t1 = O::sqn(t1, 10);
  t0 = O::mul(t0, t1);
  t0 = O::sqn(t0, 4);
  t4 = O::mul(t4, t0);
  t4 = O::sqn(t4, 6);
  t14 = O::mul(x, t4);
  t14 = O::sqn(t14, 8);
  return O::mul(t6, t14);
}
__device__ __noinline__ sc sc_mont_inv(sc x) {
  struct ops {
    static __device__ __forceinline__ sc mul(const sc& a, const sc& b) { return sc_montmul(a, b); }
    static __device__ __forceinline__ sc sqn(const sc& a, int n) { return sc_montsqr_n(a, n); }
  };
  return sc_inv_chain<sc, ops>(x);
}

// round(k * g / 2^384) for plain k, g (mulGFlooredDiv, point_mul_glv.go:119-189): < 2^128
S2K_DEV sc sc_mul_g_floored_div(const sc& k, const uint32_t g[8]) {
  uint32_t t[16];
  u256_mul_wide(t, k.v, g);
  uint32_t add = t[11] >> 31;          // bit 383
  sc r = sc_zero();
  unsigned c = 0;
  r.v[0] = __builtin_addc(t[12], add, c, &c);
  r.v[1] = __builtin_addc(t[13], 0u, c, &c);
  r.v[2] = __builtin_addc(t[14], 0u, c, &c);
  r.v[3] = __builtin_addc(t[15], 0u, c, &c);
  return r;
}
// splitGLV (point_mul_glv.go:59-117) followed by the sign normalisation of
// scalarMultVartimeGLV (:212-220): k = (-1)^neg1*k1 + (-1)^neg2*k2*lambda (mod n), with
// k1, k2 < 2^128 (limbs 4..7 zero).
// out[0..7] -= / += a[0..3] * b[0..4] (a 128-bit times a 129-bit value, b[4] is 0 or 1), modulo 2^256
S2K_DEV void u256_mac_128(uint32_t out[8], const uint32_t a[4], const uint32_t b[5], bool subtract) {
  uint32_t p[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) p[i] = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t carry = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t t = (uint64_t)a[i] * b[j] + p[i + j] + carry;    // < 2^64: no overflow
      p[i + j] = (uint32_t)t;
      carry = (uint32_t)(t >> 32);
    }
    p[i + 4] = carry;
  }
  {   // + a * b[4] * 2^128
    const uint32_t m = 0u - b[4];
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) p[4 + i] = __builtin_addc(p[4 + i], a[i] & m, c, &c);
  }
  unsigned c = 0;
  if (subtract) {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = __builtin_subc(out[i], p[i], c, &c);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = __builtin_addc(out[i], p[i], c, &c);
  }
}
// The reference computes k2 = c1 (-b1) + c2 (-b2) and k1 = k - k2 lambda with three products mod n.  With the lattice
// basis (a1, b1), (a2, b2) behind c1, c2 (a_i + b_i lambda = 0 mod n) the same two values are k1 = k - c1 a1 - c2 a2 and
// k2 = -c1 b1 - c2 b2 as INTEGERS, both below 2^128 in magnitude: four 128-bit products modulo 2^256, no reduction, and
// the sign is the top bit (a quarter of the instructions; same k1, k2 and signs for every k: tests/test_glv_odd_model.py,
// and on the device tests/test_gpu_parity.py test_fn_ops against the reference's boundary scalars).
__device__ static const uint32_t GLV_LAT_A1[5] = {0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u, 0u};    // a1 = b2
__device__ static const uint32_t GLV_LAT_NB1[5] = {0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u, 0u};   // -b1
__device__ static const uint32_t GLV_LAT_A2[5] = {0x9d44cfd8u, 0x57c1108du, 0xa8e2f3f6u, 0x14ca50f7u, 1u};    // a2
S2K_DEV void sc_split_glv(const sc& k, sc& k1, bool& neg1, sc& k2, bool& neg2) {
  const sc c1 = sc_mul_g_floored_div(k, SC_G1);
  const sc c2 = sc_mul_g_floored_div(k, SC_G2);
  uint32_t d1[8], d2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    d1[i] = k.v[i];
    d2[i] = 0;
  }
  u256_mac_128(d1, c1.v, GLV_LAT_A1, true);      // k - c1 a1
  u256_mac_128(d1, c2.v, GLV_LAT_A2, true);      //   - c2 a2
  u256_mac_128(d2, c1.v, GLV_LAT_NB1, false);    // c1 (-b1)
  u256_mac_128(d2, c2.v, GLV_LAT_A1, true);      //   - c2 b2
  neg1 = (d1[7] >> 31) != 0;
  neg2 = (d2[7] >> 31) != 0;
  // magnitudes: two's complement negation where negative (|.| < 2^128: the upper words end up zero)
  const uint32_t m1 = 0u - (uint32_t)neg1, m2 = 0u - (uint32_t)neg2;
  unsigned cy1 = neg1 ? 1u : 0u, cy2 = neg2 ? 1u : 0u;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    k1.v[i] = __builtin_addc(d1[i] ^ m1, 0u, cy1, &cy1);
    k2.v[i] = __builtin_addc(d2[i] ^ m2, 0u, cy2, &cy2);
  }
}


// ---- odd halves ------------------------------------------------------------------------------
// The signed-odd-digit ladder of k_verify_fast wants both half-scalars odd.  Instead of forcing
// the low bit and taking the extra +-Q back out with two more point additions, a short vector of
// the GLV lattice {(a, b): a + b*lambda = 0 (mod n)} is added to (k1, k2): v1 = (a1, b1) has both
// components odd, v2 = (a2, b2) is (even, odd), v2 - v1 is (odd, even).  The long component is
// always added against the sign of the half it could overflow, so |k1|, |k2| < 2^129 (limb 4 of
// the results is 0 or 1).  Model and bounds: tests/test_glv_odd_model.py.
__device__ static const uint32_t GLV_A1[5] = {0x9284eb15u, 0xe86c90e4u, 0xa7d46bcdu, 0x3086d221u, 0u};
__device__ static const uint32_t GLV_NB1[5] = {0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u, 0u};   // -b1
__device__ static const uint32_t GLV_A2[5] = {0x9d44cfd8u, 0x57c1108du, 0xa8e2f3f6u, 0x14ca50f7u, 1u};
__device__ static const uint32_t GLV_A2_A1[5] = {0x0abfe4c3u, 0x6f547fa9u, 0x010e8828u, 0xe4437ed6u, 0u};   // a2 - a1
__device__ static const uint32_t GLV_B2_B1[5] = {0x9d44cfd8u, 0x57c1108du, 0xa8e2f3f6u, 0x14ca50f7u, 1u};   // b2 - b1 = a1 - b1

// (neg, mag) += (cneg, c) on sign-magnitude 160-bit values
S2K_DEV void sm160_add(uint32_t mag[5], bool& neg, const uint32_t c[5], bool cneg) {
  uint32_t sum[5], d1[5], d2[5];
  unsigned cy = 0, bw1 = 0, bw2 = 0;
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    sum[i] = __builtin_addc(mag[i], c[i], cy, &cy);
    d1[i] = __builtin_subc(mag[i], c[i], bw1, &bw1);     // mag - c
    d2[i] = __builtin_subc(c[i], mag[i], bw2, &bw2);     // c - mag
  }
  const bool same = neg == cneg;
  const bool mag_lt = bw1 != 0;                           // mag < c
#pragma unroll
  for (int i = 0; i < 5; ++i) mag[i] = same ? sum[i] : (mag_lt ? d2[i] : d1[i]);
  neg = same ? neg : (mag_lt ? cneg : neg);
}

// splitGLV with both halves odd: k = (-1)^neg1 k1 + (-1)^neg2 k2 lambda (mod n), k1, k2 odd,
// < 2^129 (limbs 0..4, limbs 5..7 zero)
S2K_DEV void sc_split_glv_odd(const sc& k, sc& k1, bool& neg1, sc& k2, bool& neg2) {
  sc_split_glv(k, k1, neg1, k2, neg2);
  uint32_t m1[5] = {k1.v[0], k1.v[1], k1.v[2], k1.v[3], 0u}, m2[5] = {k2.v[0], k2.v[1], k2.v[2], k2.v[3], 0u};
  const bool odd1 = m1[0] & 1u, odd2 = m2[0] & 1u;
  if (!odd1 && !odd2) {            // + v1 = (a1, b1), b1 < 0
    // The natural halves (k1, k2), kept for the rule below (magnitude in n2, sign in n2neg)
    const bool k1_zero = (m1[0] | m1[1] | m1[2] | m1[3]) == 0u;
    const uint32_t n2[4] = {m2[0], m2[1], m2[2], m2[3]};
    const bool n2neg = neg2;
    sm160_add(m1, neg1, GLV_A1, false);
    sm160_add(m2, neg2, GLV_NB1, true);
    if (k1_zero) {
      // k = k2 * lambda with a SMALL even k2: the first half has collapsed and the halves the ladders see are (a1, k2 + b1).
      // If k2 = 2 s d w - s the sign of the second half, d its signed digit at the position a ladder adds LAST, w that
      // position's weight (digit 0, weight 1: the general ladder; digit 28, weight 16^28: the ladder over per-key tables) -
      // then everything but the last addend sums to the last addend, the last table addition of the ladder is P + P, and
      // the lane ends on the complete-formula worklist.  One value of k2 does that for each ladder (-26 and -26 * 16^28:
      // found by simulating the ladders on integers, tests/test_glv_odd_model.py), and anyone can put it into every lane of
      // a batch (any r, s = r / (k2 lambda)): 3.4 x a step.  - v1 is as good a vector as + v1 (same bound), has other
      // digits, and does not collide (same test): take it when + v1 would.
      bool hit = false;
#pragma unroll
      for (int form = 0; form < 2; ++form) {
        const uint32_t nib = form ? (m2[3] >> 17) & 15u : (m2[0] >> 1) & 15u;       // digit 28 / digit 0: bits 4i+1 .. 4i+4
        const int d = 2 * (int)nib - 15, sd = neg2 ? -d : d;                          // the addend's sign times its digit
        const uint32_t mag = 2u * (uint32_t)(sd < 0 ? -sd : sd);
        const bool same_mag = form ? (n2[3] == (mag << 16) && (n2[0] | n2[1] | n2[2]) == 0u)
                                   : (n2[0] == mag && (n2[1] | n2[2] | n2[3]) == 0u);
        hit = hit || (same_mag && n2neg == (sd < 0));
      }
      if (hit) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          m1[i] = 0u;
          m2[i] = n2[i];
        }
        m1[4] = m2[4] = 0u;
        neg1 = false;
        neg2 = n2neg;
        sm160_add(m1, neg1, GLV_A1, true);      // - v1
        sm160_add(m2, neg2, GLV_NB1, false);
      }
    }
  } else if (odd1 && !odd2) {      // -+ v2 = (a2, b2), b2 = a1: against the sign of k1
    const bool cneg = !neg1;
    sm160_add(m1, neg1, GLV_A2, cneg);
    sm160_add(m2, neg2, GLV_A1, cneg);
  } else if (!odd1 && odd2) {      // -+ (v2 - v1): against the sign of k2
    const bool cneg = !neg2;
    sm160_add(m1, neg1, GLV_A2_A1, cneg);
    sm160_add(m2, neg2, GLV_B2_B1, cneg);
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    k1.v[i] = m1[i];
    k2.v[i] = m2[i];
  }
}

}  // namespace s2k

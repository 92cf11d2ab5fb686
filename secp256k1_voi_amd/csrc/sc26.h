// sc26.h — products modulo the group order n on 10 x 26-bit limbs (gfx950), for the scalar
// preparation of the batch kernels: s^-1 by Montgomery's trick, u1 = e/s, u2 = r/s
// (ecdsa.go:206-228, scalar.go:91-100, scalar_invert.go:11-303).
//
// Same idea as fe26.h: 26-bit limbs leave room to sum twenty 52-bit products in a 64-bit
// v_mad_u64_u32 chain with no carry handling.  n has no special form, so this is Montgomery
// multiplication with R = 2^260 (product scanning, tools/gen_sc26_mul.py).  Values in a chain
// are "lazy": < 2n, limbs < 2^26; inputs < 2n give outputs < 2n because R > 16n.  A value is
// "plain" (x) or "Montgomery" (x*R mod n); sc26_montmul(plain, mont) is a plain product.
// sc.h (8 x 32, always canonical) stays the storage / comparison / GLV type.
#pragma once
#include "fe26.h"
#include "modinv30.h"
#include "sc.h"

namespace s2k {

struct sc26 {
  uint32_t n[10];
};

#include "sc26_mul_gen.h"

S2K_DEV sc26 sc26_from_limbs(const uint32_t* p) {
  sc26 r;
#pragma unroll
  for (int i = 0; i < 10; ++i) r.n[i] = p[i];
  return r;
}
// canonical 8 x 32 scalar -> limbs (same value)
S2K_DEV sc26 sc26_from_sc(const sc& a) {
  fe26 t = fe26_from_words(a.v);
  return sc26_from_limbs(t.n);
}
// lazy (< 2n) limbs -> canonical 8 x 32 scalar
S2K_DEV sc sc26_to_sc(const sc26& a) {
  fe26 t;
#pragma unroll
  for (int i = 0; i < 10; ++i) t.n[i] = a.n[i];
  uint32_t top = t.n[9] >> 22;   // bit 256
  t.n[9] &= 0x3FFFFFu;
  uint32_t w[8], d[8];
  fe26_to_words(w, t);
  uint32_t borrow = u256_sub(d, w, SC_N);
  bool use_d = top != 0 || !borrow;
  sc r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r.v[i] = use_d ? d[i] : w[i];
  return r;
}

__device__ __noinline__ sc26 sc26_mm(sc26 a, sc26 b) { return sc26_montmul(a, b); }
__device__ __noinline__ sc26 sc26_sqr_n(sc26 a, int n) {
#pragma unroll 1
  for (int i = 0; i < n; ++i) a = sc26_montsqr(a);
  return a;
}
S2K_DEV sc26 sc26_to_mont(const sc26& a) { return sc26_mm(a, sc26_from_limbs(SC26_R2)); }

// x^-1 in the Montgomery domain (in: x*R, out: x^-1*R): the plain inverse of x*R by the safegcd
// division steps (modinv30.h, ~10 k instructions against ~68 k for the Fermat chain of
// scalar_invert.go:11-303), then * R^3 * R^-1 to land on x^-1 * R.  Invert(0) = 0.
#ifndef S2K_SC_INV_FERMAT
#define S2K_SC_INV_FERMAT 0   // 1: the Fermat addition chain on the 10x26 form (A/B runs)
#endif
__device__ __noinline__ sc26 sc26_mont_inv(sc26 x) {
#if S2K_SC_INV_FERMAT
  struct ops {
    static __device__ __forceinline__ sc26 mul(const sc26& a, const sc26& b) { return sc26_mm(a, b); }
    static __device__ __forceinline__ sc26 sqn(const sc26& a, int n) { return sc26_sqr_n(a, n); }
  };
  return sc_inv_chain<sc26, ops>(x);
#else
  return sc26_mm(sc26_from_sc(sc_modinv(sc26_to_sc(x))), sc26_from_limbs(SC26_R3));
#endif
}

}  // namespace s2k

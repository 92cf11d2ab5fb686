// modinv30.h — x^-1 mod n (and mod p) by the Bernstein–Yang "safegcd" division steps (eprint 2019/266), on
// 9 signed limbs of 30 bits, one inversion per lane, constant instruction flow (no lane diverges).
//
// Replaces the Fermat chain of Scalar.Invert (scalar_invert.go:11-303: 253 squarings + 40
// products, ~68 k VALU instructions on the 10x26 Montgomery form) in the batched scalar
// preparation: 20 rounds of 30 division steps on the low words, each round followed by one 2x2
// matrix update of (f, g) and of (d, e) mod n, ~10 k instructions.  Same value as the reference
// for every input (0 -> 0); the algorithm and its 600-step bound for 256-bit moduli are the
// published ones (also used by libsecp256k1's modinv32).  tests: fn INV through s2k_fn_op_batch;
// integer model of exactly this limb schedule: tests/test_modinv_model.py.
#pragma once
#include "sc.h"

namespace s2k {

struct s30 {
  int32_t v[9];
};

constexpr int32_t MI_M30 = 0x3FFFFFFF;
__device__ static const int32_t MI_N[9] = {0x10364141, 0x3f497a33, 0x348a03bb, 0x2bb739ab, 0x3ffffeba,
                                           0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff};
constexpr uint32_t MI_NINV30 = 0x2a774ec1u;   // n^-1 mod 2^30
// the same for the field prime p = 2^256 - 2^32 - 977 (k_key_chain / k_key_finish: per-key tables, keyed.hip)
__device__ static const int32_t MI_P[9] = {0x3ffffc2f, 0x3ffffffb, 0x3fffffff, 0x3fffffff, 0x3fffffff,
                                           0x3fffffff, 0x3fffffff, 0x3fffffff, 0xffff};
constexpr uint32_t MI_PINV30 = 0x2ddacacfu;   // p^-1 mod 2^30

// acc += a * b (signed 32 x 32 -> 64), as one v_mad_i64_i32 (inline asm: see pt29.h on why no
// C-level 64-bit multiply is left to the compiler)
S2K_DEV void mi_mad(int64_t& acc, int32_t a, int32_t b) {
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
}

// 30 division steps on the low words; t = (u, v, q, r), the transition matrix scaled by 2^30
S2K_DEV int32_t mi_divsteps30(int32_t zeta, uint32_t f, uint32_t g, int32_t t[4]) {
  uint32_t u = 1, v = 0, q = 0, r = 1;
#pragma unroll 1
  for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);          // all ones when zeta < 0
    uint32_t c2 = 0u - (g & 1u);
    uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;
    g += x & c2;
    q += y & c2;
    r += z & c2;
    c1 &= c2;
    zeta = (int32_t)(((uint32_t)zeta ^ c1) - 1u);
    f += g & c1;
    u += q & c1;
    v += r & c1;
    g >>= 1;
    u <<= 1;
    v <<= 1;
  }
  t[0] = (int32_t)u;
  t[1] = (int32_t)v;
  t[2] = (int32_t)q;
  t[3] = (int32_t)r;
  return zeta;
}

// (f, g) <- (u f + v g, q f + r g) / 2^30   (exact: the low 30 bits cancel)
S2K_DEV void mi_update_fg(s30& f, s30& g, const int32_t t[4]) {
  int64_t cf = 0, cg = 0;
  mi_mad(cf, t[0], f.v[0]);
  mi_mad(cf, t[1], g.v[0]);
  mi_mad(cg, t[2], f.v[0]);
  mi_mad(cg, t[3], g.v[0]);
  cf >>= 30;
  cg >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int32_t fi = f.v[i], gi = g.v[i];
    mi_mad(cf, t[0], fi);
    mi_mad(cf, t[1], gi);
    mi_mad(cg, t[2], fi);
    mi_mad(cg, t[3], gi);
    f.v[i - 1] = (int32_t)cf & MI_M30;
    cf >>= 30;
    g.v[i - 1] = (int32_t)cg & MI_M30;
    cg >>= 30;
  }
  f.v[8] = (int32_t)cf;
  g.v[8] = (int32_t)cg;
}

// (d, e) <- (u d + v e, q d + r e) / 2^30 mod m, kept in (-2m, m); m = p (FP) or n
template <bool FP>
S2K_DEV void mi_update_de(s30& d, s30& e, const int32_t t[4]) {
  const int32_t* MI_N = FP ? MI_P : s2k::MI_N;
  const uint32_t MI_NINV30 = FP ? MI_PINV30 : s2k::MI_NINV30;
  const int32_t u = t[0], v = t[1], q = t[2], r = t[3];
  const int32_t sd = d.v[8] >> 31, se = e.v[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
  int64_t cd = 0, ce = 0;
  mi_mad(cd, u, d.v[0]);
  mi_mad(cd, v, e.v[0]);
  mi_mad(ce, q, d.v[0]);
  mi_mad(ce, r, e.v[0]);
  // multiples of n that make the low 30 bits vanish
  md -= (int32_t)((MI_NINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)MI_M30);
  me -= (int32_t)((MI_NINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)MI_M30);
  mi_mad(cd, MI_N[0], md);
  mi_mad(ce, MI_N[0], me);
  cd >>= 30;
  ce >>= 30;
#pragma unroll
  for (int i = 1; i < 9; ++i) {
    const int32_t di = d.v[i], ei = e.v[i];
    mi_mad(cd, u, di);
    mi_mad(cd, v, ei);
    mi_mad(ce, q, di);
    mi_mad(ce, r, ei);
    mi_mad(cd, MI_N[i], md);
    mi_mad(ce, MI_N[i], me);
    d.v[i - 1] = (int32_t)cd & MI_M30;
    cd >>= 30;
    e.v[i - 1] = (int32_t)ce & MI_M30;
    ce >>= 30;
  }
  d.v[8] = (int32_t)cd;
  e.v[8] = (int32_t)ce;
}

// r in (-2m, m) -> (+-r) in [0, m); sign < 0 negates
template <bool FP>
S2K_DEV void mi_normalize(s30& r, int32_t sign) {
  const int32_t* MI_N = FP ? MI_P : s2k::MI_N;
  int32_t c = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] += MI_N[i] & c;
  c = sign >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] = (r.v[i] ^ c) - c;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= MI_M30;
  }
  c = r.v[8] >> 31;
#pragma unroll
  for (int i = 0; i < 9; ++i) r.v[i] += MI_N[i] & c;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    r.v[i + 1] += r.v[i] >> 30;
    r.v[i] &= MI_M30;
  }
}

// x^-1 mod m for a canonical x held in 8 words (0 -> 0)
template <bool FP>
S2K_DEV void mi_modinv_words(uint32_t out[8], const uint32_t x[8]) {
  const int32_t* M = FP ? MI_P : MI_N;
  s30 d, e, f, g;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    d.v[i] = 0;
    e.v[i] = i == 0 ? 1 : 0;
    f.v[i] = M[i];
  }
  // 8 x 32 -> 9 x 30
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int lo = 30 * i, w = lo >> 5, sh = lo & 31;
    uint32_t limb = x[w] >> sh;
    if (sh > 2 && w + 1 < 8) limb |= x[w + 1] << (32 - sh);
    g.v[i] = (int32_t)(limb & (uint32_t)MI_M30);
  }
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; ++it) {
    int32_t t[4];
    zeta = mi_divsteps30(zeta, (uint32_t)f.v[0], (uint32_t)g.v[0], t);
    mi_update_de<FP>(d, e, t);
    mi_update_fg(f, g, t);
  }
  mi_normalize<FP>(d, f.v[8]);
  // 9 x 30 -> 8 x 32
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int lo = 32 * j, i = lo / 30, sh = lo - 30 * i;      // word j starts at bit `sh` of limb i
    uint32_t wv = (uint32_t)d.v[i] >> sh;
    wv |= (uint32_t)d.v[i + 1] << (30 - sh);
    if (30 - sh + 30 < 32 && i + 2 < 9) wv |= (uint32_t)d.v[i + 2] << (60 - sh);
    out[j] = wv;
  }
}

// x^-1 mod n for a canonical x (Scalar.Invert, scalar_invert.go:11; 0 -> 0)
__device__ __noinline__ sc sc_modinv(sc x) {
  sc r;
  mi_modinv_words<false>(r.v, x.v);
  return r;
}

}  // namespace s2k

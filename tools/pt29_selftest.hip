// Device self-test of pt29.h against point.h (8x32 complete formulas), step by step.
//   hipcc -O3 --offload-arch=gfx950 -I secp256k1_voi_amd/csrc tools/pt29_selftest.hip -o /tmp/pt29_selftest
#include <hip/hip_runtime.h>
#include <cstdio>
#include "engine_internal.h"
#include "fe.h"
#include "point.h"
#include "pt29.h"
using namespace s2k;

__device__ bool same(const fe29& a, const fe& b) {
  uint32_t w[8];
  fe29_to_words(w, fe29_normalize(a));
  fe bn = fe_normalize(b);
  return u256_eq(w, bn.v);
}
#define CHECK(name, a29, afe) if (!same(a29, afe)) { printf("MISMATCH %s\n", name); }

__global__ void k_test() {
  // P = 2G (projective, from doubling), Q = 3G = 2G + G
  apt g;
  g.x = fe_from_limbs(FE_GX);
  g.y = fe_from_limbs(FE_GY);
  pt G = pt_from_affine(g);
  pt P = pt_double_complete(G);
  pt Q = pt_add_complete(P, G);
  pt29 p, q;
  p.x = fe29_from_words(fe_normalize(P.x).v); p.y = fe29_from_words(fe_normalize(P.y).v); p.z = fe29_from_words(fe_normalize(P.z).v);
  q.x = fe29_from_words(fe_normalize(Q.x).v); q.y = fe29_from_words(fe_normalize(Q.y).v); q.z = fe29_from_words(fe_normalize(Q.z).v);
  CHECK("conv px", p.x, P.x);
  // primitives
  CHECK("mul", fe29_mul(p.x, q.x), fe_mul(P.x, Q.x));
  CHECK("small21", fe29_mul_small_norm(p.z, 21), fe_mul_small(P.z, 21));
  CHECK("small63", fe29_mul_small_norm(p.z, 63), fe_mul_small(P.z, 63));
  CHECK("triple", fe29_triple_norm(p.y), fe_mul_small(P.y, 3));
  CHECK("mul_plus", fe29_mul_plus(p.x, q.y, p.z), fe_add(fe_mul(P.x, Q.y), P.z));
  CHECK("mul_plus_neg", fe29_mul_plus(fe29_add(p.x, p.y), fe29_add(q.x, q.y), fe29_negate(fe29_add(p.z, q.z), 2)),
        fe_sub(fe_mul(fe_add(P.x, P.y), fe_add(Q.x, Q.y)), fe_add(P.z, Q.z)));
  CHECK("mul_add_mul", fe29_mul_add_mul(p.x, q.y, fe29_negate(p.z, 1), q.z), fe_sub(fe_mul(P.x, Q.y), fe_mul(P.z, Q.z)));
  pt R = pt_add_complete(P, Q);
  pt29 r = pt29_add(p, q);
  // projective equality: cross-multiply
  {
    uint32_t w[8];
    fe29_to_words(w, fe29_normalize(r.x)); fe rx = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(r.y)); fe ry = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(r.z)); fe rz = fe_from_limbs(w);
    if (!fe_eq(fe_mul(rx, R.z), fe_mul(R.x, rz))) printf("MISMATCH add x\n");
    if (!fe_eq(fe_mul(ry, R.z), fe_mul(R.y, rz))) printf("MISMATCH add y\n");
    pt D = pt_double_complete(P);
    pt29 d = pt29_double(p);
    fe29_to_words(w, fe29_normalize(d.x)); rx = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(d.y)); ry = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(d.z)); rz = fe_from_limbs(w);
    if (!fe_eq(fe_mul(rx, D.z), fe_mul(D.x, rz))) printf("MISMATCH dbl x\n");
    if (!fe_eq(fe_mul(ry, D.z), fe_mul(D.y, rz))) printf("MISMATCH dbl y\n");
    pt M = pt_add_complete(P, G);
    pt29 m = pt29_add_mixed(p, fe29_from_words(g.x.v), fe29_from_words(g.y.v));
    fe29_to_words(w, fe29_normalize(m.x)); rx = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(m.y)); ry = fe_from_limbs(w);
    fe29_to_words(w, fe29_normalize(m.z)); rz = fe_from_limbs(w);
    if (!fe_eq(fe_mul(rx, M.z), fe_mul(M.x, rz))) printf("MISMATCH mixed x\n");
    if (!fe_eq(fe_mul(ry, M.z), fe_mul(M.y, rz))) printf("MISMATCH mixed y\n");
  }
  {
    // mixed addition step by step
    fe29 qx = fe29_from_words(g.x.v), qy = fe29_from_words(g.y.v);
    fe29 t0 = fe29_mul(p.x, qx), t1 = fe29_mul(p.y, qy);
    fe T0 = fe_mul(P.x, g.x), T1 = fe_mul(P.y, g.y);
    CHECK("m t0", t0, T0); CHECK("m t1", t1, T1);
    fe29 t3 = fe29_mul_plus(fe29_add(qx, qy), fe29_add(p.x, p.y), fe29_negate(fe29_add(t0, t1), 2));
    fe T3 = fe_sub(fe_mul(fe_add(g.x, g.y), fe_add(P.x, P.y)), fe_add(T0, T1));
    CHECK("m t3", t3, T3);
    fe29 t4 = fe29_mul_plus(qy, p.z, p.y);
    fe T4 = fe_add(fe_mul(g.y, P.z), P.y);
    CHECK("m t4", t4, T4);
    fe29 y3 = fe29_mul_small_norm(fe29_mul_plus(qx, p.z, p.x), 21);
    fe Y3 = fe_mul_small(fe_add(fe_mul(g.x, P.z), P.x), 21);
    CHECK("m y3", y3, Y3);
    {
      fe29 w = fe29_mul_plus(qx, p.z, p.x);
      CHECK("m w", w, fe_add(fe_mul(g.x, P.z), P.x));
      uint32_t ww[8];
      fe29_to_words(ww, fe29_normalize(w));
      fe Wn = fe_from_limbs(ww);
      CHECK("m w21", fe29_mul_small_norm(w, 21), fe_mul_small(Wn, 21));
      CHECK("m wn21", fe29_mul_small_norm(fe29_normalize(w), 21), fe_mul_small(Wn, 21));
      fe29 r = fe29_mul_small_norm(w, 21);
      printf("w : %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", w.n[0], w.n[1], w.n[2], w.n[3], w.n[4], w.n[5], w.n[6], w.n[7], w.n[8]);
      printf("r : %08x %08x %08x %08x %08x %08x %08x %08x %08x\n", r.n[0], r.n[1], r.n[2], r.n[3], r.n[4], r.n[5], r.n[6], r.n[7], r.n[8]);
      fe e = fe_mul_small(Wn, 21);
      printf("e : %08x %08x %08x %08x %08x %08x %08x %08x\n", e.v[0], e.v[1], e.v[2], e.v[3], e.v[4], e.v[5], e.v[6], e.v[7]);
    }
    fe29 t2 = fe29_mul_small_norm(p.z, 21);
    fe T2 = fe_mul_small(P.z, 21);
    CHECK("m t2", t2, T2);
    fe29 t0n = fe29_triple_norm(t0);
    fe T0n = fe_mul_small(T0, 3);
    CHECK("m t0n", t0n, T0n);
    fe29 z3 = fe29_add(t1, t2);
    fe Z3 = fe_add(T1, T2);
    CHECK("m z3", z3, Z3);
    fe29 t1m = fe29_add(t1, fe29_negate(t2, 1));
    fe T1m = fe_sub(T1, T2);
    CHECK("m t1m", t1m, T1m);
    CHECK("m rx", fe29_mul_add_mul(t3, t1m, fe29_negate(t4, 1), y3), fe_sub(fe_mul(T3, T1m), fe_mul(T4, Y3)));
    CHECK("m ry", fe29_mul_add_mul(t1m, z3, y3, t0n), fe_add(fe_mul(T1m, Z3), fe_mul(Y3, T0n)));
    CHECK("m rz", fe29_mul_add_mul(z3, t4, t0n, t3), fe_add(fe_mul(Z3, T4), fe_mul(T0n, T3)));
    pt29 m = pt29_add_mixed(p, qx, qy);
    CHECK("m fx", m.x, fe_sub(fe_mul(T3, T1m), fe_mul(T4, Y3)));
    CHECK("m fy", m.y, fe_add(fe_mul(T1m, Z3), fe_mul(Y3, T0n)));
    CHECK("m fz", m.z, fe_add(fe_mul(Z3, T4), fe_mul(T0n, T3)));
    pt M = pt_add_complete(P, G);
    CHECK("M x", m.x, M.x); CHECK("M y", m.y, M.y); CHECK("M z", m.z, M.z);
  }
  printf("done\n");
}
int main() {
  k_test<<<1, 1>>>();
  hipDeviceSynchronize();
  return 0;
}

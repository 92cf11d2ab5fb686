// Reproducer of a hipcc 7.2 (ROCm 7.2.0, gfx950, -O3) miscompilation met in round 2: six variants of the first
// form of k_point_fallback (engine.hip).  On MI355X: kA, kC, kE, kF give per-lane garbage for k = 1 on P = G;
// kB (no generator part) and kD (no worklist indirection) are right.  The shipped kernel avoids the shape.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 miscompile_point_fallback.hip -o /tmp/mpf  (the kernel text
// was cut from engine.hip at commit "fast-ladder and 9x29 test-access entry points").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include "../../secp256k1_voi_amd/csrc/complete_path.h"
using namespace s2k;
__global__ void __launch_bounds__(256)
kA(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
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
    uint32_t raw[8];
    load_be32(raw, u2 + idx * 32);
    pt res = pt_select(!finite, pt_mul_glv(sc_reduce_once(raw), a, qt, stride, idx), pt_identity());
    if (u1) {
      load_be32(raw, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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

__global__ void __launch_bounds__(256)
kB(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
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
    uint32_t raw[8];
    load_be32(raw, u2 + idx * 32);
    pt res = pt_select(!finite, pt_mul_glv(sc_reduce_once(raw), a, qt, stride, idx), pt_identity());
    if (false) {
      load_be32(raw, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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

__global__ void __launch_bounds__(256)
kC(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
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
    uint32_t raw[8];
    load_be32(raw, u2 + idx * 32);
    sc kk = sc_reduce_once(raw);
    pt res = pt_mul_glv(kk, a, qt, stride, idx);
    res = pt_select(!finite, res, pt_identity());
    if (u1) {
      load_be32(raw, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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

__global__ void __launch_bounds__(256)
kD(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = w;
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
    uint32_t raw[8];
    load_be32(raw, u2 + idx * 32);
    pt res = pt_select(!finite, pt_mul_glv(sc_reduce_once(raw), a, qt, stride, idx), pt_identity());
    if (u1) {
      load_be32(raw, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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

__global__ void __launch_bounds__(256)
kE(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
    const uint8_t* rec = pts65 + idx * 65;
    apt a;
    load_be32_unaligned(a.x.v, rec + 1);
    load_be32_unaligned(a.y.v, rec + 33);
    bool finite = rec[0] == 0x04 && fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v) && apt_on_curve(a);
    
    if (!finite) {   // keep the arithmetic on the curve; the term is masked below
      a.x = fe_from_limbs(FE_GX);
      a.y = fe_from_limbs(FE_GY);
    }
    uint32_t raw[8];
    load_be32(raw, u2 + idx * 32);
    pt res = pt_select(!finite, pt_mul_glv(sc_reduce_once(raw), a, qt, stride, idx), pt_identity());
    if (u1) {
      load_be32(raw, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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

__global__ void __launch_bounds__(256)
kF(const uint32_t* __restrict__ wl_count, const uint32_t* __restrict__ wl, uint32_t all_n,
                 const uint8_t* __restrict__ u1, const uint8_t* __restrict__ u2, const uint8_t* __restrict__ pts65,
                 uint8_t* __restrict__ out65, gt_view gt, uint32_t* __restrict__ qt, size_t stride,
                 uint32_t* __restrict__ status) {
  uint32_t count = all_n ? all_n : *wl_count;
  for (uint32_t w = blockIdx.x * 256 + threadIdx.x; w < count; w += gridDim.x * 256) {
    size_t idx = all_n ? w : wl[w];
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
    uint32_t raw[8], raw1[8];
    load_be32(raw, u2 + idx * 32);
    pt res = pt_select(!finite, pt_mul_glv(sc_reduce_once(raw), a, qt, stride, idx), pt_identity());
    if (u1) {
      load_be32(raw1, u1 + idx * 32);
      sc v1 = sc_reduce_once(raw1);
      res = pt_add_complete(pt_base_mul(gt, v1.v), res);   // point_mul_glv.go:316
    }
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


typedef void (*kfn)(const uint32_t*, const uint32_t*, uint32_t, const uint8_t*, const uint8_t*, const uint8_t*, uint8_t*, gt_view, uint32_t*, size_t, uint32_t*);
int main() {
  const int n = 8;
  uint8_t hk[n * 32], hp[n * 65], ho[n * 65];
  memset(hk, 0, sizeof hk);
  for (int i = 0; i < n; ++i) hk[i * 32 + 31] = 1;
  static const uint32_t gx[8] = {0x16f81798u, 0x59f2815bu, 0x2dce28d9u, 0x029bfcdbu, 0xce870b07u, 0x55a06295u, 0xf9dcbbacu, 0x79be667eu};
  static const uint32_t gy[8] = {0xfb10d4b8u, 0x9c47d08fu, 0xa6855419u, 0xfd17b448u, 0x0e1108a8u, 0x5da4fbfcu, 0x26a3c465u, 0x483ada77u};
  for (int i = 0; i < n; ++i) {
    hp[i * 65] = 4;
    for (int w = 0; w < 8; ++w)
      for (int b = 0; b < 4; ++b) {
        hp[i * 65 + 1 + (7 - w) * 4 + b] = (uint8_t)(gx[w] >> (24 - 8 * b));
        hp[i * 65 + 33 + (7 - w) * 4 + b] = (uint8_t)(gy[w] >> (24 - 8 * b));
      }
  }
  uint8_t *dk, *dp, *dout; uint32_t *qt, *st, *gt, *wl;
  hipMalloc(&dk, sizeof hk); hipMalloc(&dp, sizeof hp + 64); hipMalloc(&dout, sizeof ho + 64); hipMalloc(&qt, 192 * 64 * 4 * 4); hipMalloc(&st, 64);
  hipMalloc(&gt, 1 << 20); hipMalloc(&wl, 1 << 12);
  hipMemcpy(dk, hk, sizeof hk, hipMemcpyHostToDevice); hipMemcpy(dp, hp, sizeof hp, hipMemcpyHostToDevice); hipMemset(st, 0, 64);
  kfn ks[] = {kA,kB,kC,kD,kE,kF};
  const char* names[] = {"kA","kB","kC","kD","kE","kF"};
  for (int v = 0; v < (int)(sizeof ks / sizeof ks[0]); ++v) {
    hipMemset(dout, 0xEE, sizeof ho);
    ks[v]<<<1, 256>>>(wl, wl + 64, n, nullptr, dk, dp, dout, gt_view{gt, 8u, 32u}, qt, 64, st);   // (a table of 8-bit windows fits the megabyte; the generator part is not taken: u1 is null)
    hipDeviceSynchronize();
    hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i) bad += memcmp(ho + i * 65, hp + i * 65, 65) != 0;
    printf("%s: %d of %d wrong\n", names[v], bad, n);
  }
  return 0;
}

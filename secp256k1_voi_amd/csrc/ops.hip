// ops.hip — the Point / Scalar / Element entry points (the Go types' methods need
// something to call), batched: scalar multiplications on the complete path, Point.Add / Double on
// the 9x29 complete formulas of the multiscalar kernels, SEC1 decoding, field and scalar
// arithmetic, the valid-bitmap packing of the shard exchange, generator-table inspection.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "complete_path.h"
#include "engine_internal.h"
#include "pt29.h"
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

enum { PK_BASE_MUL = 0, PK_MUL = 1, PK_DOUBLE_MUL = 2, PK_ADD = 3, PK_DOUBLE = 4 };

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

__global__ void __launch_bounds__(256)
k_point_op(int op, uint32_t n, const uint8_t* __restrict__ k1, const uint8_t* __restrict__ k2,
           const uint8_t* __restrict__ pa, const uint8_t* __restrict__ pb, uint8_t* __restrict__ out,
           const uint32_t* __restrict__ gt, uint32_t* __restrict__ qt, size_t stride) {
  size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  pt res = pt_identity();
  if (op == PK_BASE_MUL) {
    uint32_t raw[8];
    load_be32(raw, k1 + idx * 32);
    sc k = sc_reduce_once(raw);
    res = pt_base_mul(gt, k.v);
  } else if (op == PK_MUL || op == PK_DOUBLE_MUL) {
    uint32_t raw[8];
    pt p;
    point_record_load(p, pa + idx * 65);
    bool p_inf = pt_is_identity(p);
    apt a;
    a.x = p.x;
    a.y = p.y;                    // Z = 1 for finite records
    if (p_inf) {                   // keep the arithmetic on the curve; result is masked below
      a.x = fe_from_limbs(FE_GX);
      a.y = fe_from_limbs(FE_GY);
    }
    load_be32(raw, (op == PK_MUL ? k1 : k2) + idx * 32);
    sc k = sc_reduce_once(raw);
    pt rq = pt_mul_glv(k, a, qt, stride, idx);
    rq = pt_select(p_inf, rq, pt_identity());
    if (op == PK_DOUBLE_MUL) {
      load_be32(raw, k1 + idx * 32);
      sc u1 = sc_reduce_once(raw);
      res = pt_add_complete(pt_base_mul(gt, u1.v), rq);
    } else {
      res = rq;
    }
  } else if (op == PK_ADD) {
    // Point.Add (point.go:62) through the 9x29 complete formulas the multiscalar kernels use
    pt a, b;
    point_record_load(a, pa + idx * 65);
    point_record_load(b, pb + idx * 65);
    res = pt_from_pt29(pt29_add(pt29_from_pt(a), pt29_from_pt(b)));
  } else if (op == PK_DOUBLE) {
    pt a;
    point_record_load(a, pa + idx * 65);
    res = pt_from_pt29(pt29_double(pt29_from_pt(a)));
  }
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

// valid bytes -> bitmap (bit i of byte i/8, LSB first) + number of valid items.  One lane
// per 8 items; the count is reduced per wave with a ballot-free popcount sum and one atomic.
__global__ void __launch_bounds__(256)
k_pack_valid(uint32_t n, const uint8_t* __restrict__ valid, uint8_t* __restrict__ bitmap,
             unsigned long long* __restrict__ count) {
  size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
  size_t base = t * 8;
  uint32_t bits = 0;
  if (base + 8 <= n) {
    uint2 v = *reinterpret_cast<const uint2*>(valid + base);
    uint32_t lo = v.x & 0x01010101u, hi = v.y & 0x01010101u;
    // gather bit 0 of each byte: multiply trick
    bits = ((lo * 0x10204080u) >> 28) | (((hi * 0x10204080u) >> 28) << 4);
  } else if (base < n) {
    for (uint32_t j = 0; base + j < n; ++j) bits |= (uint32_t)(valid[base + j] & 1u) << j;
  }
  if (base < n) bitmap[t] = (uint8_t)bits;
  uint32_t c = __popc(bits);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) c += __shfl_down(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

__global__ void k_gtable_entry(const uint32_t* __restrict__ gt, uint32_t window, uint32_t digit, uint8_t* out64) {
  apt a = gt_load(gt, window, digit);
  store_be32(out64, a.x.v);
  store_be32(out64 + 32, a.y.v);
}

extern "C" {

static int point_op(s2k_ctx* ctx, int op, size_t n, const uint8_t* k1, const uint8_t* k2, const uint8_t* pa,
                    const uint8_t* pb, uint8_t* out) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!out) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf dk1, dk2, dpa, dpb, dout;
  if (k1) HIP_TRY(ctx, dk1.upload(k1, n * 32));
  if (k2) HIP_TRY(ctx, dk2.upload(k2, n * 32));
  if (pa) HIP_TRY(ctx, dpa.upload(pa, n * 65));
  if (pb) HIP_TRY(ctx, dpb.upload(pb, n * 65));
  HIP_TRY(ctx, dout.alloc(n * 65));
  int rc = s2k_internal_ensure_ws(ctx, n);
  if (rc) return rc;
  k_point_op<<<blocks_for(n), 256>>>(op, (uint32_t)n, (const uint8_t*)dk1.p, (const uint8_t*)dk2.p,
                                     (const uint8_t*)dpa.p, (const uint8_t*)dpb.p, (uint8_t*)dout.p, ctx->gtable,
                                     (uint32_t*)ctx->ws, lane_stride(n));   // the table region starts the workspace
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out, dout.p, n * 65, hipMemcpyDeviceToHost));
  return S2K_OK;
}

int s2k_scalar_base_mult_batch(s2k_ctx* ctx, size_t n, const uint8_t* k, uint8_t* out) {
  if (n && !k) return fail(ctx, S2K_ERR_ARG, "null scalar buffer");
  return point_op(ctx, PK_BASE_MUL, n, k, nullptr, nullptr, nullptr, out);
}
int s2k_scalar_mult_batch(s2k_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* points, uint8_t* out) {
  if (n && (!k || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_MUL, n, k, nullptr, points, nullptr, out);
}
int s2k_double_scalar_mult_basepoint_batch(s2k_ctx* ctx, size_t n, const uint8_t* u1, const uint8_t* u2,
                                           const uint8_t* points, uint8_t* out) {
  if (n && (!u1 || !u2 || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_DOUBLE_MUL, n, u1, u2, points, nullptr, out);
}
int s2k_point_add_batch(s2k_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out) {
  if (n && (!a || !b)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_ADD, n, nullptr, nullptr, a, b, out);
}
int s2k_point_double_batch(s2k_ctx* ctx, size_t n, const uint8_t* a, uint8_t* out) {
  if (n && !a) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  return point_op(ctx, PK_DOUBLE, n, nullptr, nullptr, a, nullptr, out);
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

int s2k_pack_valid_device(s2k_ctx* ctx, size_t n, const void* d_valid, void* d_bitmap, void* d_count,
                          void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!d_count) return fail(ctx, S2K_ERR_ARG, "null count buffer");
  hipStream_t st = (hipStream_t)hip_stream;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipMemsetAsync(d_count, 0, 8, st));
  if (n == 0) return S2K_OK;
  if (!d_valid || !d_bitmap) return fail(ctx, S2K_ERR_ARG, "null buffer");
  k_pack_valid<<<blocks_for((n + 7) / 8), 256, 0, st>>>((uint32_t)n, (const uint8_t*)d_valid, (uint8_t*)d_bitmap,
                                                        (unsigned long long*)d_count);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}

int s2k_generator_window_bits(void) { return GT_BITS; }

int s2k_debug_gtable_entry(s2k_ctx* ctx, unsigned i, unsigned d, uint8_t* out64) {
  if (!ctx || !out64) return fail(ctx, S2K_ERR_ARG, "null argument");
  if (i >= (unsigned)GT_WINDOWS || d >= (1u << GT_BITS)) return fail(ctx, S2K_ERR_ARG, "index out of range");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf o;
  HIP_TRY(ctx, o.alloc(64));
  k_gtable_entry<<<1, 1>>>(ctx->gtable, i, d, (uint8_t*)o.p);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out64, o.p, 64, hipMemcpyDeviceToHost));
  return S2K_OK;
}

}  // extern "C"

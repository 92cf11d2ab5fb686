// msm.hip — multi-scalar multiplication  sum_i k_i * P_i  on gfx950.
//
// Serves Point.MultiScalarMult / MultiScalarMultVartime (point_mul_multi.go:25,73).  The
// reference implements Straus and notes that Pippenger's bucket method is the better
// algorithm for large batches (point_mul_multi.go:16-18, README.md:91); this is that method,
// laid out for a GPU.  Parity is on the resulting group element (canonical bytes).
//
//   1. k_msm_prepare   one lane per term: reduce the scalar, parse the point, histogram of
//                      (window, digit) keys with atomics
//   2. k_msm_scan      exclusive prefix sum of the histogram (one workgroup)
//   3. k_msm_scatter   counting sort: term indices grouped by key
//   4. k_msm_accumulate one lane per bucket: complete mixed additions of its points
//   5. k_msm_reduce    one lane per chunk of 32 buckets: sum_b b*B_b by running sums, plus the
//                      chunk offset by double-and-add
//   6. k_msm_tree      one workgroup per window: tree-sum of the chunk results
//   7. k_msm_final     Horner over the windows, affine result
// All additions use the complete formulas (pt26.h): buckets receive arbitrary points
// (duplicates, inverses, the same point many times), so there is no exceptional case to
// detect and no fallback.  Window width c is 8 bits for small inputs and 16 bits for large
// ones (2^20 terms: 16 windows x 65535 buckets, ~16 points per bucket).
#include "engine_internal.h"
#include "pt26.h"
#include "sc.h"

namespace {

constexpr int CHUNK = 32;   // buckets per reduction chunk

struct msm_geom {
  uint32_t c;        // window bits
  uint32_t nw;       // windows = ceil(256 / c)
  uint32_t nb;       // keys per window = 2^c (key 0 unused)
  uint32_t nchunk;   // nb / CHUNK
};

S2K_DEV uint32_t msm_digit(const uint32_t* __restrict__ scw, size_t n_stride, size_t i, uint32_t w, uint32_t c) {
  uint32_t bit = w * c, word = bit >> 5, sh = bit & 31;
  uint32_t lo = scw[(size_t)word * n_stride + i];
  uint32_t hi = word + 1 < 8 ? scw[(size_t)(word + 1) * n_stride + i] : 0u;
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}

// pt26 in planes [word][slot]
S2K_DEV void pt_store(uint32_t* __restrict__ base, size_t stride, size_t slot, const pt26& p) {
#pragma unroll
  for (int w = 0; w < 10; ++w) base[(size_t)w * stride + slot] = p.x.n[w];
#pragma unroll
  for (int w = 0; w < 10; ++w) base[(size_t)(10 + w) * stride + slot] = p.y.n[w];
#pragma unroll
  for (int w = 0; w < 10; ++w) base[(size_t)(20 + w) * stride + slot] = p.z.n[w];
}
S2K_DEV pt26 pt_load(const uint32_t* __restrict__ base, size_t stride, size_t slot) {
  pt26 p;
#pragma unroll
  for (int w = 0; w < 10; ++w) p.x.n[w] = base[(size_t)w * stride + slot];
#pragma unroll
  for (int w = 0; w < 10; ++w) p.y.n[w] = base[(size_t)(10 + w) * stride + slot];
#pragma unroll
  for (int w = 0; w < 10; ++w) p.z.n[w] = base[(size_t)(20 + w) * stride + slot];
  return p;
}
// keep magnitudes at the pt26 invariant after a select etc.
S2K_DEV pt26 pt_select(bool pick_b, const pt26& a, const pt26& b) {
  pt26 r;
  r.x = fe26_select(pick_b, a.x, b.x);
  r.y = fe26_select(pick_b, a.y, b.y);
  r.z = fe26_select(pick_b, a.z, b.z);
  return r;
}

__global__ void __launch_bounds__(256)
k_msm_prepare(uint32_t n, msm_geom g, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points,
              uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, uint8_t* __restrict__ flag,
              uint32_t* __restrict__ count, uint32_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t raw[8];
  load_be32(raw, scalars + i * 32);
  sc k = sc_reduce_once(raw);                     // SetBytes semantics (scalar.go:123)
#pragma unroll
  for (int w = 0; w < 8; ++w) scw[(size_t)w * n + i] = k.v[w];
  const uint8_t* rec = points + i * 65;
  uint8_t f = 0;                                  // 0 identity, 1 finite, 2 malformed
  apt a;
  a.x = fe_zero();
  a.y = fe_zero();
  if (rec[0] == 0x04) {
    load_be32_unaligned(a.x.v, rec + 1);
    load_be32_unaligned(a.y.v, rec + 33);
    f = (fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v) && apt_on_curve(a)) ? 1 : 2;
  } else if (rec[0] != 0x00) {
    f = 2;
  }
  if (f == 2) atomicOr(status, 1u);
  flag[i] = f;
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    ptw[(size_t)w * n + i] = a.x.v[w];
    ptw[(size_t)(8 + w) * n + i] = a.y.v[w];
  }
  if (f != 1) return;
  for (uint32_t w = 0; w < g.nw; ++w) {
    uint32_t d = msm_digit(scw, n, i, w, g.c);
    if (d) atomicAdd(&count[(size_t)w * g.nb + d], 1u);
  }
}

// exclusive scan of `total` counters (total a multiple of 1024) by one 1024-thread workgroup
__global__ void __launch_bounds__(1024) k_msm_scan(const uint32_t* __restrict__ count, uint32_t* __restrict__ offset,
                                                   uint32_t total) {
  __shared__ uint32_t part[1024];
  uint32_t per = total / 1024, t = threadIdx.x;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; ++j) sum += count[(size_t)t * per + j];
  part[t] = sum;
  __syncthreads();
  for (uint32_t s = 1; s < 1024; s <<= 1) {
    uint32_t v = t >= s ? part[t - s] : 0;
    __syncthreads();
    part[t] += v;
    __syncthreads();
  }
  uint32_t run = t ? part[t - 1] : 0;
  for (uint32_t j = 0; j < per; ++j) {
    offset[(size_t)t * per + j] = run;
    run += count[(size_t)t * per + j];
  }
  if (t == 1023) offset[total] = run;
}

__global__ void __launch_bounds__(256)
k_msm_scatter(uint32_t n, msm_geom g, const uint32_t* __restrict__ scw, const uint8_t* __restrict__ flag,
              const uint32_t* __restrict__ offset, uint32_t* __restrict__ cursor, uint32_t* __restrict__ list) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n || flag[i] != 1) return;
  for (uint32_t w = 0; w < g.nw; ++w) {
    uint32_t d = msm_digit(scw, n, i, w, g.c);
    if (!d) continue;
    size_t key = (size_t)w * g.nb + d;
    uint32_t pos = atomicAdd(&cursor[key], 1u);
    list[offset[key] + pos] = (uint32_t)i;
  }
}

__global__ void __launch_bounds__(256)
k_msm_accumulate(uint32_t nkeys, uint32_t n, const uint32_t* __restrict__ offset, const uint32_t* __restrict__ list,
                 const uint32_t* __restrict__ ptw, uint32_t* __restrict__ buckets) {
  size_t key = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (key >= nkeys) return;
  uint32_t lo = offset[key], hi = offset[key + 1];
  pt26 acc = pt26_identity();
#pragma unroll 1
  for (uint32_t j = lo; j < hi; ++j) {
    size_t i = list[j];
    uint32_t xw[8], yw[8];
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      xw[w] = ptw[(size_t)w * n + i];
      yw[w] = ptw[(size_t)(8 + w) * n + i];
    }
    acc = pt26_add_mixed(acc, fe26_from_words(xw), fe26_from_words(yw));
  }
  pt_store(buckets, nkeys, key, acc);
}

// chunk (w, j): buckets b in [32j, 32j+32) of window w -> sum_b b * B_b
__global__ void __launch_bounds__(64)
k_msm_reduce(msm_geom g, const uint32_t* __restrict__ buckets, uint32_t* __restrict__ partial) {
  size_t id = (size_t)blockIdx.x * 64 + threadIdx.x;
  size_t nslots = (size_t)g.nw * g.nchunk;
  if (id >= nslots) return;
  uint32_t w = (uint32_t)(id / g.nchunk), j = (uint32_t)(id % g.nchunk);
  size_t nkeys = (size_t)g.nw * g.nb;
  size_t base = (size_t)w * g.nb + (size_t)j * CHUNK;
  pt26 run = pt26_identity(), tot = pt26_identity();
#pragma unroll 1
  for (int b = CHUNK - 1; b >= 1; --b) {
    run = pt26_add(run, pt_load(buckets, nkeys, base + b));
    tot = pt26_add(tot, run);
  }
  run = pt26_add(run, pt_load(buckets, nkeys, base));       // S_j (bucket 32j has coefficient 0 in tot)
  // tot += (32 j) * S_j :  j * S_j by double-and-add (j < 2^11), then 5 doublings
  pt26 m = pt26_identity();
#pragma unroll 1
  for (int bit = 10; bit >= 0; --bit) {
    m = pt26_double(m);
    pt26 s = pt26_add(m, run);
    m = pt_select((j >> bit) & 1u, m, s);
  }
#pragma unroll 1
  for (int t = 0; t < 5; ++t) m = pt26_double(m);
  tot = pt26_add(tot, m);
  pt_store(partial, nslots, id, tot);
}

// one workgroup per window: partial[w][0..nchunk) -> partial[w][0]
__global__ void __launch_bounds__(1024) k_msm_tree(msm_geom g, uint32_t* __restrict__ partial) {
  size_t nslots = (size_t)g.nw * g.nchunk;
  size_t base = (size_t)blockIdx.x * g.nchunk;
  for (uint32_t half = g.nchunk >> 1; half >= 1; half >>= 1) {
    for (uint32_t t = threadIdx.x; t < half; t += 1024) {
      pt26 a = pt_load(partial, nslots, base + t), b = pt_load(partial, nslots, base + t + half);
      pt_store(partial, nslots, base + t, pt26_add(a, b));
    }
    __syncthreads();
  }
}

// Horner over the window sums, then the 65-byte record
__global__ void k_msm_final(msm_geom g, const uint32_t* __restrict__ partial, uint8_t* __restrict__ out65) {
  size_t nslots = (size_t)g.nw * g.nchunk;
  pt26 acc = pt_load(partial, nslots, (size_t)(g.nw - 1) * g.nchunk);
#pragma unroll 1
  for (int w = (int)g.nw - 2; w >= 0; --w) {
#pragma unroll 1
    for (uint32_t t = 0; t < g.c; ++t) acc = pt26_double(acc);
    acc = pt26_add(acc, pt_load(partial, nslots, (size_t)w * g.nchunk));
  }
  if (fe26_is_zero(acc.z)) {
    for (int i = 0; i < 65; ++i) out65[i] = 0;
    return;
  }
  fe26 zi = fe26_inv(fe26_normalize_weak(acc.z));
  fe26 x = fe26_normalize(fe26_mul(acc.x, zi)), y = fe26_normalize(fe26_mul(acc.y, zi));
  // 10x26 -> 8x32 words -> big-endian bytes
  auto words = [](const fe26& a, uint32_t w[8]) {
    w[0] = a.n[0] | (a.n[1] << 26);
    w[1] = (a.n[1] >> 6) | (a.n[2] << 20);
    w[2] = (a.n[2] >> 12) | (a.n[3] << 14);
    w[3] = (a.n[3] >> 18) | (a.n[4] << 8);
    w[4] = (a.n[4] >> 24) | (a.n[5] << 2) | (a.n[6] << 28);
    w[5] = (a.n[6] >> 4) | (a.n[7] << 22);
    w[6] = (a.n[7] >> 10) | (a.n[8] << 16);
    w[7] = (a.n[8] >> 16) | (a.n[9] << 10);
  };
  uint32_t xw[8], yw[8];
  words(x, xw);
  words(y, yw);
  out65[0] = 0x04;
  store_be32_unaligned(out65 + 1, xw);
  store_be32_unaligned(out65 + 33, yw);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace

extern "C" {

int s2k_multi_scalar_mult_device(s2k_ctx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out65,
                                 void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!d_out65) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n && (!d_scalars || !d_points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  if (n > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  if (n == 0) {   // l == 0: identity (point_mul_multi.go:37 v.Identity())
    HIP_TRY(ctx, hipMemsetAsync(d_out65, 0, 65, st));
    return S2K_OK;
  }
  msm_geom g;
  g.c = n >= (1u << 14) ? 16 : (n >= 256 ? 12 : 8);
  g.nw = (256 + g.c - 1) / g.c;
  g.nb = 1u << g.c;
  g.nchunk = g.nb / CHUNK;
  const size_t nkeys = (size_t)g.nw * g.nb;              // multiple of 1024 for every c used
  const size_t nslots = (size_t)g.nw * g.nchunk;
  // carve the workspace
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  size_t o_status = carve(256), o_count = carve((nkeys + 1) * 4), o_cursor = carve(nkeys * 4),
         o_offset = carve((nkeys + 1) * 4), o_scw = carve(n * 8 * 4), o_ptw = carve(n * 16 * 4), o_flag = carve(n),
         o_list = carve(n * (size_t)g.nw * 4), o_buckets = carve(nkeys * 30 * 4), o_partial = carve(nslots * 30 * 4);
  int rc = ctx_reserve(ctx, &ctx->msm_ws, &ctx->msm_ws_bytes, off);
  if (rc) return rc;
  uint8_t* ws = (uint8_t*)ctx->msm_ws;
  uint32_t* status = (uint32_t*)(ws + o_status);
  uint32_t* count = (uint32_t*)(ws + o_count);
  uint32_t* cursor = (uint32_t*)(ws + o_cursor);
  uint32_t* offset = (uint32_t*)(ws + o_offset);
  uint32_t* scw = (uint32_t*)(ws + o_scw);
  uint32_t* ptw = (uint32_t*)(ws + o_ptw);
  uint8_t* flag = ws + o_flag;
  uint32_t* list = (uint32_t*)(ws + o_list);
  uint32_t* buckets = (uint32_t*)(ws + o_buckets);
  uint32_t* partial = (uint32_t*)(ws + o_partial);
  HIP_TRY(ctx, hipMemsetAsync(ws, 0, o_offset, st));      // status, count, cursor
  k_msm_prepare<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, g, (const uint8_t*)d_scalars, (const uint8_t*)d_points, scw,
                                               ptw, flag, count, status);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_scan<<<1, 1024, 0, st>>>(count, offset, (uint32_t)nkeys);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_scatter<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, g, scw, flag, offset, cursor, list);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_accumulate<<<blocks_for(nkeys), 256, 0, st>>>((uint32_t)nkeys, (uint32_t)n, offset, list, ptw, buckets);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_reduce<<<(unsigned)((nslots + 63) / 64), 64, 0, st>>>(g, buckets, partial);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_tree<<<g.nw, 1024, 0, st>>>(g, partial);
  HIP_TRY(ctx, hipGetLastError());
  k_msm_final<<<1, 1, 0, st>>>(g, partial, (uint8_t*)d_out65);
  HIP_TRY(ctx, hipGetLastError());
  // malformed point records are a caller error (the reference cannot even construct such Points)
  uint32_t h_status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&h_status, status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  if (h_status) return fail(ctx, S2K_ERR_ARG, "malformed point record in multi-scalar multiplication input");
  return S2K_OK;
}

int s2k_multi_scalar_mult(s2k_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out65) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!out65) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n && (!scalars || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  dev_buf dk, dp, dout;
  HIP_TRY(ctx, dk.upload(scalars, n * 32));
  HIP_TRY(ctx, dp.upload(points, n * 65));
  HIP_TRY(ctx, dout.alloc(80));
  int rc = s2k_multi_scalar_mult_device(ctx, n, dk.p, dp.p, dout.p, nullptr);
  if (rc) return rc;
  HIP_TRY(ctx, hipDeviceSynchronize());
  HIP_TRY(ctx, hipMemcpy(out65, dout.p, 65, hipMemcpyDeviceToHost));
  return S2K_OK;
}

}  // extern "C"

// msm.hip — multi-scalar multiplication  sum_i k_i * P_i  on gfx950.
//
// Serves Point.MultiScalarMult / MultiScalarMultVartime (point_mul_multi.go:25,73).  The
// reference implements Straus and notes that Pippenger's bucket method is the better
// algorithm for large batches (point_mul_multi.go:16-18, README.md:91); this is that method,
// laid out for a GPU.  Parity is on the resulting group element (canonical bytes).
//
//   1. k_msm_parse     one lane per term: reduce the scalar, parse and check the point, split with the endomorphism
//   2. sort of the (window, signed digit) keys, two levels, all counting done in LDS:
//      k_msm_coarse_count / scan / k_msm_coarse_scatter  partition the (key, term) pairs by the key's upper bits
//                      (256 keys per coarse bucket): each workgroup counts its slice of terms in LDS and publishes
//                      one row of a [coarse][workgroup] matrix, whose exclusive scan gives every workgroup a private
//                      output range per coarse bucket (no global atomics)
//      k_msm_fine_sort one workgroup per coarse bucket: LDS histogram of its 256 keys, scan, the sorted piece of the
//                      list assembled in LDS and written out in order; emits the per-key offsets
//   3. k_msm_accumulate the bucket pass: the sorted list cut into equal ranges, one lane per range, complete mixed
//                      additions; k_msm_stitch puts together the buckets that cross range borders
//   4. k_msm_reduce    four lanes per chunk of 8 buckets (pt29q.h): sum_b b*B_b by running sums, plus the chunk offset
//                      by double-and-add
//   5. k_msm_tree      tree-sum of the chunk results per window (two launches, a quad per addition)
//   6. k_msm_final     Horner over the windows, affine result
// All additions use the complete formulas (pt29.h, pt29q.h): buckets receive arbitrary points (duplicates, inverses,
// the same point many times), so there is no exceptional case to detect and no fallback.  Every term is first split
// with the curve endomorphism into two 128-bit terms (below); the digits are signed (half the buckets).  Window width
// c is 8 bits for small inputs and 16 bits for large ones (2^20 inputs: 2^21 terms, 9 slots x 32768 buckets, ~57
// points per bucket).
#include <sys/random.h>

#include <cstdlib>

#include <vector>

#include "engine_internal.h"
#include "fe29_inv.h"
#include "pt29.h"
#include "pt29q.h"
#include "fe29r.h"
#include "xyzz29.h"
#include "sc.h"
#include "sha256.h"

namespace {

#ifndef S2K_MSM_WAVES
#define S2K_MSM_WAVES 4   // waves per SIMD the bucket pass is built for (127 VGPRs) and sized to fill once
#endif
#ifndef S2K_MSM_SPLIT_DEFAULT
#define S2K_MSM_SPLIT_DEFAULT 0   // windows in the lower part of the two-part bucket pass (msm_core; 0: one part - measured: DESIGN.md section 6)
#endif
#ifndef S2K_MSM_CHUNK_LOG2
#define S2K_MSM_CHUNK_LOG2 3   // default of the buckets per reduction chunk (log2); S2K_MSM_CHUNK_LOG2 in the environment overrides it
#endif
// Every input term k*P is split with the curve endomorphism (splitGLV, point_mul_glv.go:59) into
// |k1| * (+-P) + |k2| * (+-lambda P), |k1|, |k2| < 2^128: twice the terms, half the windows.  The
// bucket additions stay the same (2n * 8 instead of n * 16 at c = 16) but the serial tail -
// Horner over the windows, 256 - c doublings on one lane - and the bucket reductions halve.
constexpr int SCW_WORDS = 5;      // scalar planes per term: 128-bit magnitude + one zero word
constexpr uint32_t SCALAR_BITS = 128;

struct msm_geom {
  uint32_t c;        // window bits
  uint32_t nw;       // windows = ceil(128 / c)
  uint32_t nb;       // keys per slot = 2^(c-1): the magnitudes 1 .. 2^(c-1) of a signed digit (key = magnitude - 1)
  uint32_t nslot;    // nw + 1 slots of nb keys: one per window, the top window (unsigned, up to 2^c) takes two
  uint32_t chunk_log2, nchunk;   // buckets per reduction chunk (log2), chunks per slot = nb >> chunk_log2
};

// window w (c bits) of a 128-bit magnitude held in k[0..3] (k[4] = 0)
S2K_DEV uint32_t msm_digit(const uint32_t k[SCW_WORDS], uint32_t w, uint32_t c) {
  const uint32_t bit = w * c, word = bit >> 5, sh = bit & 31;
  uint32_t lo = 0, hi = 0;
#pragma unroll
  for (int j = 0; j < SCW_WORDS; ++j) {      // (selects, not indexed loads: k[] stays in registers)
    lo = word == (uint32_t)j ? k[j] : lo;
    hi = word + 1 == (uint32_t)j ? k[j] : hi;
  }
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (uint32_t)(v >> sh) & ((1u << c) - 1u);
}
// Signed digits: k = sum_w d_w 2^(cw) with |d_w| <= 2^(c-1) below the top window (a digit above 2^(c-1) becomes
// digit - 2^c and carries one into the next window); the top window stays unsigned, 0 .. 2^c, and its
// magnitudes above 2^(c-1) simply continue into the extra slot.  Half the buckets of unsigned digits: the bucket
// of magnitude m receives +P for digit m and -P for digit -m.  f(key, negative) for every non-zero digit of term i,
// key = window * nb + magnitude - 1.
constexpr uint32_t TERM_NEG = 0x80000000u;    // sign flag next to a term index (term counts stay below 2^30)
template <class F>
S2K_DEV void msm_for_digits(const uint32_t* __restrict__ scw, size_t n_stride, size_t i, const msm_geom& g, F f) {
  uint32_t k[SCW_WORDS];
#pragma unroll
  for (int j = 0; j < SCW_WORDS - 1; ++j) k[j] = scw[(size_t)j * n_stride + i];     // the magnitude, loaded once
  k[SCW_WORDS - 1] = 0;
  uint32_t carry = 0;
  const uint32_t half = g.nb, full = g.nb << 1;
  for (uint32_t w = 0; w < g.nw; ++w) {
    uint32_t v = msm_digit(k, w, g.c) + carry;
    const bool neg = w + 1 < g.nw && v > half;
    carry = neg ? 1u : 0u;
    const uint32_t m = neg ? full - v : v;
    if (m) f(w * g.nb + m - 1u, neg);
  }
}

// pt29 in planes [word][slot], PT_WORDS words per point
constexpr int PT_WORDS = 27;
S2K_DEV void pt_store(uint32_t* __restrict__ base, size_t stride, size_t slot, const pt29& p) {
#pragma unroll
  for (int w = 0; w < 9; ++w) base[(size_t)w * stride + slot] = p.x.n[w];
#pragma unroll
  for (int w = 0; w < 9; ++w) base[(size_t)(9 + w) * stride + slot] = p.y.n[w];
#pragma unroll
  for (int w = 0; w < 9; ++w) base[(size_t)(18 + w) * stride + slot] = p.z.n[w];
}
S2K_DEV pt29 pt_load(const uint32_t* __restrict__ base, size_t stride, size_t slot) {
  pt29 p;
#pragma unroll
  for (int w = 0; w < 9; ++w) p.x.n[w] = base[(size_t)w * stride + slot];
#pragma unroll
  for (int w = 0; w < 9; ++w) p.y.n[w] = base[(size_t)(9 + w) * stride + slot];
#pragma unroll
  for (int w = 0; w < 9; ++w) p.z.n[w] = base[(size_t)(18 + w) * stride + slot];
  return p;
}
// keep magnitudes at the pt29 invariant after a select etc.
S2K_DEV pt29 pt_select(bool pick_b, const pt29& a, const pt29& b) {
  pt29 r;
  r.x = fe29_select(pick_b, a.x, b.x);
  r.y = fe29_select(pick_b, a.y, b.y);
  r.z = fe29_select(pick_b, a.z, b.z);
  return r;
}

// one coordinate (cc = 0, 1, 2: x, y, z) of a stored point: the quad-spread group law (pt29q.h) keeps a point in four lanes
S2K_DEV fe29 ptq_load(const uint32_t* __restrict__ base, size_t stride, size_t slot, uint32_t cc) {
  fe29 r;
#pragma unroll
  for (int w = 0; w < 9; ++w) r.n[w] = base[(size_t)(9 * cc + w) * stride + slot];
  return r;
}
S2K_DEV void ptq_store(uint32_t* __restrict__ base, size_t stride, size_t slot, uint32_t cc, const fe29& v) {
#pragma unroll
  for (int w = 0; w < 9; ++w) base[(size_t)(9 * cc + w) * stride + slot] = v.n[w];
}

// term `t` of a term array with plane stride N: magnitude k (< 2^128), point (x, +-y)
S2K_DEV void msm_store_term(uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, size_t N, size_t t, const sc& k,
                            const uint32_t x[8], const uint32_t y[8], bool neg) {
#pragma unroll
  for (int w = 0; w < 4; ++w) scw[(size_t)w * N + t] = k.v[w];
  scw[(size_t)4 * N + t] = 0;
  uint32_t ny[8];
  u256_sub(ny, FE_P, y);                                   // y != 0 on this curve
  // points as 64-byte records [t][16 words]: the bucket pass gathers them by index, one
  // contiguous record per lane (planes would cost 16 sparse sectors per gathered point)
  uint4* rec4 = reinterpret_cast<uint4*>(ptw + t * 16);
  rec4[0] = make_uint4(x[0], x[1], x[2], x[3]);
  rec4[1] = make_uint4(x[4], x[5], x[6], x[7]);
  rec4[2] = neg ? make_uint4(ny[0], ny[1], ny[2], ny[3]) : make_uint4(y[0], y[1], y[2], y[3]);
  rec4[3] = neg ? make_uint4(ny[4], ny[5], ny[6], ny[7]) : make_uint4(y[4], y[5], y[6], y[7]);
}
// k * (x, y) -> terms t1, t2:  |k1| * (x, +-y) and |k2| * (beta x, +-y)   (mulBeta, point_mul_glv.go:191)
S2K_DEV void msm_store_split(uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, size_t N, size_t t1, size_t t2,
                             const sc& k, const uint32_t x[8], const uint32_t y[8]) {
  sc k1, k2;
  bool neg1, neg2;
  sc_split_glv(k, k1, neg1, k2, neg2);
  msm_store_term(scw, ptw, N, t1, k1, x, y, neg1);
  uint32_t bx[8];
  fe29_to_words(bx, fe29_normalize(fe29_mul(fe29_from_words(x), fe29_from_words(FE_BETA))));
  msm_store_term(scw, ptw, N, t2, k2, bx, y, neg2);
}

// bytes -> terms: scalar reduced mod n and split, affine point words; flag per term
// (0 identity, 1 finite, 2 malformed).  Input i becomes terms i and n + i.
__global__ void __launch_bounds__(256)
k_msm_parse(uint32_t n, const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ points,
            uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, uint8_t* __restrict__ flag,
            uint32_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t raw[8];
  load_be32(raw, scalars + i * 32);
  sc k = sc_reduce_once(raw);                     // SetBytes semantics (scalar.go:123)
  const uint8_t* rec = points + i * 65;
  uint8_t f = 0;
  apt a;
  a.x = fe_zero();
  a.y = fe_zero();
  if (rec[0] == 0x04) {
    load_be32_unaligned(a.x.v, rec + 1);
    load_be32_unaligned(a.y.v, rec + 33);
    bool on = fe_is_canonical_raw(a.x.v) && fe_is_canonical_raw(a.y.v);
    if (on) {   // y^2 == x^3 + 7 on the 9x29 field (a third of the instructions of the 8x32 form; xyOnCurve, point_s11n.go:298-307)
      const fe29 x = fe29_from_words(a.x.v), y = fe29_from_words(a.y.v);
      fe29 rhs = fe29_mul(fe29_sqr(x), x);
      rhs.n[0] += 7;
      on = fe29_eq(fe29_sqr(y), rhs);
    }
    f = on ? 1 : 2;
  } else if (rec[0] != 0x00) {
    f = 2;
  }
  if (f == 2) atomicOr(status, 1u);
  flag[i] = f;
  flag[(size_t)n + i] = f;
  if (f == 1) msm_store_split(scw, ptw, 2 * (size_t)n, i, (size_t)n + i, k, a.x.v, a.y.v);
}

// exclusive scan of `total` counters (total a multiple of 1024), three small launches:
// per-block sums -> scan of the block sums (one workgroup) -> per-block scan + base
__global__ void __launch_bounds__(256) k_msm_scan_blocks(const uint32_t* __restrict__ count, uint32_t* __restrict__ bsum) {
  __shared__ uint32_t part[256];
  const uint4 v = reinterpret_cast<const uint4*>(count)[(size_t)blockIdx.x * 256 + threadIdx.x];
  part[threadIdx.x] = v.x + v.y + v.z + v.w;
  __syncthreads();
  for (uint32_t s = 128; s >= 1; s >>= 1) {
    if (threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) bsum[blockIdx.x] = part[0];
}
// nblocks <= 1024 * 8
__global__ void __launch_bounds__(1024) k_msm_scan_top(uint32_t* __restrict__ bsum, uint32_t nblocks,
                                                       uint32_t* __restrict__ total_out) {
  __shared__ uint32_t part[1024];
  uint32_t per = (nblocks + 1023) / 1024, t = threadIdx.x;
  uint32_t sum = 0;
  for (uint32_t j = 0; j < per; ++j) {
    uint32_t i = t * per + j;
    if (i < nblocks) sum += bsum[i];
  }
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
    uint32_t i = t * per + j;
    if (i < nblocks) {
      uint32_t v = bsum[i];
      bsum[i] = run;
      run += v;
    }
  }
  if (t == 1023) *total_out = part[1023];
}
// (count and offset may be the same array: each thread reads its four counters before writing)
__global__ void __launch_bounds__(256) k_msm_scan_apply(const uint32_t* count, const uint32_t* __restrict__ bsum,
                                                        uint32_t* offset) {
  __shared__ uint32_t part[256];
  const size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x;
  const uint4 v = reinterpret_cast<const uint4*>(count)[i4];
  uint32_t mine = v.x + v.y + v.z + v.w;
  part[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t s = 1; s < 256; s <<= 1) {
    uint32_t a = threadIdx.x >= s ? part[threadIdx.x - s] : 0;
    __syncthreads();
    part[threadIdx.x] += a;
    __syncthreads();
  }
  uint32_t base = bsum[blockIdx.x] + part[threadIdx.x] - mine;
  reinterpret_cast<uint4*>(offset)[i4] = make_uint4(base, base + v.x, base + v.x + v.y, base + v.x + v.y + v.z);
}

// ---------------------------------------------------------------------------------------
// Two-level counting sort of the (key, term) pairs, key = window * 2^(c-1) + |digit| - 1 (zero digits skipped).
// FINE keys per coarse bucket; a sort workgroup owns SORT_TERMS consecutive terms.  A pair is ONE 32-bit word while the
// term index fits 23 bits (fine key << 24 | sign << 23 | term: up to 2^22 inputs), two words beyond (WIDE).
// ---------------------------------------------------------------------------------------
constexpr uint32_t FINE_BITS = 8, FINE = 1u << FINE_BITS;
constexpr uint32_t SORT_THREADS = 1024;
// terms per workgroup of the coarse kernels (a run-time value, msm_setup): 2048 where the workgroup's pairs fit the LDS
// stage of k_msm_coarse_scatter_staged, 8192 otherwise
constexpr uint32_t SORT_TERMS_STAGED = 2048, SORT_TERMS_DIRECT = 8192, STAGE_PAIRS = 16384;
constexpr uint32_t MAX_COARSE = 1152;     // nkeys / FINE, at most 9 * 2^15 / 2^8 (c = 16, the largest geometry)
constexpr uint32_t NARROW_TERM_BITS = 23;
constexpr uint32_t FS_THREADS = 512, FS_STAGE = 19456;   // fine sort: threads, list entries staged in LDS (76 KiB: two workgroups per CU)

template <bool WIDE> struct msm_pair;
template <> struct msm_pair<false> {
  typedef uint32_t type;
  static S2K_DEV type make(uint32_t fine, uint32_t term, bool neg) { return fine << 24 | (neg ? 1u << NARROW_TERM_BITS : 0u) | term; }
  static S2K_DEV uint32_t fine(type p) { return p >> 24; }
  static S2K_DEV uint32_t entry(type p) { return (p & ((1u << NARROW_TERM_BITS) - 1u)) | ((p >> NARROW_TERM_BITS & 1u) << 31); }
};
template <> struct msm_pair<true> {
  typedef uint2 type;
  static S2K_DEV type make(uint32_t fine, uint32_t term, bool neg) { return make_uint2(fine, term | (neg ? TERM_NEG : 0u)); }
  static S2K_DEV uint32_t fine(type p) { return p.x; }
  static S2K_DEV uint32_t entry(type p) { return p.y; }
};

// matrix[coarse * nblk_pad + block] = pairs of this workgroup's terms falling into `coarse`
__global__ void __launch_bounds__(SORT_THREADS)
k_msm_coarse_count(uint32_t n, size_t nstride, msm_geom g, const uint32_t* __restrict__ scw, const uint8_t* __restrict__ flag,
                   uint32_t ncoarse, uint32_t nblk_pad, uint32_t* __restrict__ matrix, uint32_t sort_terms) {
  __shared__ uint32_t h[MAX_COARSE];
  for (uint32_t t = threadIdx.x; t < ncoarse; t += SORT_THREADS) h[t] = 0;
  __syncthreads();
  size_t base = (size_t)blockIdx.x * sort_terms;
  for (uint32_t t = threadIdx.x; t < sort_terms; t += SORT_THREADS) {
    size_t i = base + t;
    if (i >= n || flag[i] != 1) continue;
    msm_for_digits(scw, nstride, i, g, [&](uint32_t key, bool) { atomicAdd(&h[key >> FINE_BITS], 1u); });
  }
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < ncoarse; t += SORT_THREADS) matrix[(size_t)t * nblk_pad + blockIdx.x] = h[t];
}
// pairs[pos] = (key & (FINE - 1), term, sign), grouped by coarse bucket; `mbase` is the scanned matrix
template <bool WIDE>
__global__ void __launch_bounds__(SORT_THREADS)
k_msm_coarse_scatter(uint32_t n, size_t nstride, msm_geom g, const uint32_t* __restrict__ scw, const uint8_t* __restrict__ flag,
                     uint32_t ncoarse, uint32_t nblk_pad, const uint32_t* __restrict__ mbase,
                     typename msm_pair<WIDE>::type* __restrict__ pairs, uint32_t sort_terms) {
  __shared__ uint32_t cur[MAX_COARSE];
  for (uint32_t t = threadIdx.x; t < ncoarse; t += SORT_THREADS) cur[t] = mbase[(size_t)t * nblk_pad + blockIdx.x];
  __syncthreads();
  size_t base = (size_t)blockIdx.x * sort_terms;
  for (uint32_t t = threadIdx.x; t < sort_terms; t += SORT_THREADS) {
    size_t i = base + t;
    if (i >= n || flag[i] != 1) continue;
    msm_for_digits(scw, nstride, i, g, [&](uint32_t key, bool neg) {
      uint32_t pos = atomicAdd(&cur[key >> FINE_BITS], 1u);
      pairs[pos] = msm_pair<WIDE>::make(key & (FINE - 1), (uint32_t)i, neg);
    });
  }
}
// The same with the workgroup's pairs put in order in LDS first and written out run by run (one-word pairs, at most
// STAGE_PAIRS of them per workgroup).  Scattered straight to memory, a workgroup's 4-byte stores went to 1152 places at
// once, 32 workgroups per XCD kept more partly written lines open than the 4 MiB L2 holds, and the lines left it a few
// words at a time: 358 MB of write traffic for 65 MB of pairs.  The counts of this workgroup are the differences of
// neighbouring entries of the scanned matrix (its row is [coarse][workgroup]: the next entry is the next workgroup's
// base in the same coarse bucket, or the next bucket's first).
__global__ void __launch_bounds__(SORT_THREADS)
k_msm_coarse_scatter_staged(uint32_t n, size_t nstride, msm_geom g, const uint32_t* __restrict__ scw, const uint8_t* __restrict__ flag,
                            uint32_t ncoarse, uint32_t nblk_pad, const uint32_t* __restrict__ mbase, uint32_t* __restrict__ pairs,
                            uint32_t sort_terms) {
  __shared__ uint32_t gbase[MAX_COARSE], lofs[MAX_COARSE], lcur[MAX_COARSE], part[SORT_THREADS], stage[STAGE_PAIRS];
  __shared__ uint16_t sbin[STAGE_PAIRS];
  const uint32_t t = threadIdx.x;
  // this workgroup's count per coarse bucket, two buckets per thread, and their exclusive scan
  uint32_t c0 = 0, c1 = 0;
  {
    const uint32_t b0 = 2 * t, b1 = 2 * t + 1;
    if (b0 < ncoarse) {
      const size_t at = (size_t)b0 * nblk_pad + blockIdx.x;
      gbase[b0] = mbase[at];
      c0 = mbase[at + 1] - mbase[at];
    }
    if (b1 < ncoarse) {
      const size_t at = (size_t)b1 * nblk_pad + blockIdx.x;
      gbase[b1] = mbase[at];
      c1 = mbase[at + 1] - mbase[at];
    }
  }
  part[t] = c0 + c1;
  __syncthreads();
  for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
    const uint32_t a = t >= d ? part[t - d] : 0;
    __syncthreads();
    part[t] += a;
    __syncthreads();
  }
  {
    const uint32_t ex = part[t] - (c0 + c1);
    if (2 * t < ncoarse) lofs[2 * t] = lcur[2 * t] = ex;
    if (2 * t + 1 < ncoarse) lofs[2 * t + 1] = lcur[2 * t + 1] = ex + c0;
  }
  const uint32_t total = part[SORT_THREADS - 1];
  __syncthreads();
  const size_t base = (size_t)blockIdx.x * sort_terms;
  for (uint32_t k = t; k < sort_terms; k += SORT_THREADS) {
    const size_t i = base + k;
    if (i >= n || flag[i] != 1) continue;
    msm_for_digits(scw, nstride, i, g, [&](uint32_t key, bool neg) {
      const uint32_t b = key >> FINE_BITS, lp = atomicAdd(&lcur[b], 1u);
      stage[lp] = msm_pair<false>::make(key & (FINE - 1), (uint32_t)i, neg);
      sbin[lp] = (uint16_t)b;
    });
  }
  __syncthreads();
  for (uint32_t k = t; k < total; k += SORT_THREADS) {
    const uint32_t b = sbin[k];
    pairs[gbase[b] + (k - lofs[b])] = stage[k];
  }
}
// The bucket pass runs in one or two PARTS (msm_core): part B = the windows below `split_key / nb` (the list positions below
// offset[split_key]), part A = the windows from there up, each cut into its own ranges - L_B entries per lane for the lanes
// [0, nlanesB), L_A for the lanes [nlanesB, nlanes) - so that either part fills the chip by itself.  One part: split_key = 0,
// nlanesB = 0, everything is part A.
struct msm_parts {
  uint32_t split_key, nlanesB, nlanes, L_A, L_B;
};
// one workgroup per coarse bucket: pairs -> list (term index | sign << 31, grouped by key), offset[key], lanekey[range];
// the last workgroup also writes offset[nkeys] = total.  The sorted piece of the list is put together in LDS and written out in
// order (scattered 4-byte stores straight to memory cost six times the list's size in write traffic); a coarse
// bucket too large for that - only engineered inputs - is scattered directly.
template <bool WIDE>
__global__ void __launch_bounds__(FS_THREADS)
k_msm_fine_sort(uint32_t ncoarse, uint32_t nblk_pad, const uint32_t* __restrict__ mbase, uint32_t total_slot,
                const typename msm_pair<WIDE>::type* __restrict__ pairs, uint32_t* __restrict__ offset,
                uint32_t* __restrict__ list, msm_parts P, uint32_t* __restrict__ lanekey) {
  typedef msm_pair<WIDE> PR;
  __shared__ uint32_t h[FINE], part[FINE], stage[FS_STAGE];
  const uint32_t b = blockIdx.x, t = threadIdx.x;
  const uint32_t lo = mbase[(size_t)b * nblk_pad];
  const uint32_t hi = b + 1 < ncoarse ? mbase[(size_t)(b + 1) * nblk_pad] : mbase[total_slot];
  if (t < FINE) h[t] = 0;
  __syncthreads();
  // four loads in flight per thread (one pair per trip made this kernel latency bound: a global load, then an LDS
  // atomic that waits for it, dozens of times in a row)
  constexpr uint32_t U = 4;
  for (uint32_t j = lo + t; j < hi; j += FS_THREADS * U) {
    uint32_t k[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) k[u] = j + u * FS_THREADS < hi ? PR::fine(pairs[j + u * FS_THREADS]) : FINE;
#pragma unroll
    for (uint32_t u = 0; u < U; ++u)
      if (k[u] < FINE) atomicAdd(&h[k[u]], 1u);
  }
  __syncthreads();
  const uint32_t mine = t < FINE ? h[t] : 0u;
  if (t < FINE) part[t] = mine;
  __syncthreads();
  for (uint32_t s = 1; s < FINE; s <<= 1) {
    uint32_t a = (t < FINE && t >= s) ? part[t - s] : 0;
    __syncthreads();
    if (t < FINE) part[t] += a;
    __syncthreads();
  }
  const bool staged = hi - lo <= FS_STAGE;
  if (t < FINE) {
    const uint32_t off = lo + part[t] - mine;
    const size_t key = (size_t)b * FINE + t;
    offset[key] = off;
    if (b + 1 == ncoarse && t == FINE - 1) offset[key + 1] = hi;
    h[t] = staged ? off - lo : off;                      // running cursor of key t
    // the bucket pass cuts each part of the list into ranges of L entries: the key of every range start that falls into this
    // bucket (saves each of its lanes a binary search of offset[], 19 dependent loads with the whole chip waiting).  Part A
    // starts where the coarse bucket of split_key starts (split_key is a multiple of FINE: the scanned matrix has the position).
    const bool in_b = key < P.split_key;
    const uint32_t start = in_b || !P.split_key ? 0u : mbase[(size_t)(P.split_key >> FINE_BITS) * nblk_pad];
    const uint32_t L = in_b ? P.L_B : P.L_A, lane0 = in_b ? 0u : P.nlanesB, rel = off - start;
    for (uint32_t k = (rel + L - 1) / L; (uint64_t)k * L < (uint64_t)rel + mine; ++k) lanekey[lane0 + k] = (uint32_t)key;
  }
  __syncthreads();
  for (uint32_t j = lo + t; j < hi; j += FS_THREADS * U) {
    typename PR::type e[U];
    bool ok[U];
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      ok[u] = j + u * FS_THREADS < hi;
      e[u] = pairs[ok[u] ? j + u * FS_THREADS : lo];
    }
#pragma unroll
    for (uint32_t u = 0; u < U; ++u) {
      if (!ok[u]) continue;
      const uint32_t pos = atomicAdd(&h[PR::fine(e[u])], 1u);
      if (staged) stage[pos] = PR::entry(e[u]); else list[pos] = PR::entry(e[u]);
    }
  }
  if (!staged) return;
  __syncthreads();
  for (uint32_t j = t; j < hi - lo; j += FS_THREADS) list[lo + j] = stage[j];
}

// ---------------------------------------------------------------------------------------
// The bucket pass.  The sorted list is cut into RANGES of L consecutive entries, one lane per range, whatever
// buckets the entries belong to: every lane does exactly L additions (the last one fewer), so all the waves of
// the launch finish together and - L chosen so that the lanes fill the chip once - every SIMD keeps its
// waves from start to end.  (One lane per bucket, the buckets ordered by size, was the round-2 design: bucket sizes
// are Poisson distributed, a SIMD's waves ran out one after the other and the longest ended alone at half the issue
// rate: 0.79-0.85 of the issue slots.)
// A lane walks its range; at every bucket border (offset[] of the sort) it flushes its accumulator and restarts
// from the identity (the complete formulas take the identity as they take any point).  A bucket that lies inside one
// range is written to its final place sums[key]; a piece of a bucket that continues from the previous range goes to
// sums[nkeys + lane] ("left edge"), a piece that continues into the next range to sums[nkeys + nlanes + lane] ("right
// edge"); a range that lies inside one bucket altogether is a left edge.  k_msm_stitch then completes every bucket
// that crosses a range border: right edge of its first range + left edges of the following ones (one addition for an
// ordinary bucket; buckets spread over more than STITCH_SERIAL ranges - only engineered inputs make those - are
// queued for k_msm_stitch_big, one workgroup per bucket, a tree over the pieces).
// ---------------------------------------------------------------------------------------
constexpr uint32_t MSM_LANES = 256u * 4u * 64u * S2K_MSM_WAVES;      // lanes that fill an MI355X once (256 CU x 4 SIMD x waves x 64)
constexpr uint32_t MSM_L_MIN = 8;             // shortest range
constexpr uint32_t STITCH_SERIAL = 8, STITCH_BIG_CAP = 4096;

// A list entry is a term index with the digit's sign: a negative digit adds -P = (x, -y), negated on the limbs
// (msm_point_of).  The record (a random 64-byte read from the term array) is fetched one addition
// ahead, into registers (fetching it in two halves to save registers cost more than it saved: the second half's line
// had left the L2 by the time it was asked for, 5.4 GB fetched per 2^20-term call instead of 3.0).
struct msm_rec {
  uint4 a, b, c, d;
};
S2K_DEV msm_rec msm_load_rec(const uint32_t* __restrict__ ptw, uint32_t entry) {
  const uint4* rec4 = reinterpret_cast<const uint4*>(ptw + (size_t)(entry & ~TERM_NEG) * 16);
  msm_rec r;
  r.a = rec4[0]; r.b = rec4[1]; r.c = rec4[2]; r.d = rec4[3];
  return r;
}
S2K_DEV void msm_point_of(const msm_rec& r, uint32_t entry, fe29& x, fe29& y) {
  const uint32_t xw[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
  const uint32_t yw[8] = {r.c.x, r.c.y, r.c.z, r.c.w, r.d.x, r.d.y, r.d.z, r.d.w};
  // a negative entry adds (x, -y): negated on the limbs (2p - y limb by limb, nine instructions, the result at two units like
  // the addends of the verification ladders) - on the 32-bit words it was an eight-word add-with-carry chain per addition
  x = fe29_from_words(xw);
  y = fe29_cond_negate1(fe29_from_words(yw), (entry >> 31) != 0);
}
// The additions are the incomplete XYZZ mixed additions of xyzz29.h: a piece STARTS as its first point, and it is flushed
// as it is - four coordinates, 36 words, one 144-byte record of `xsum` per slot: the hot loop pays stores only (some lane of a wave
// flushes in nearly every trip, so whatever the flush does, the whole wave waits for).  k_msm_stitch, one lane per key and
// outside the hot loop, turns the pieces into projective points; a piece whose ZZ is 0 - it met P + P or P - P, or it
// really sums to the identity - is walked again there with the complete formulas (msm_piece).
constexpr int XZ_WORDS = 36;
S2K_DEV void xz_store(uint32_t* __restrict__ base, size_t slot, const xyzz29& p) {
  // some lane of a wave flushes in two trips of three, and the whole wave steps through the flush: 36 stores with 36 addresses
  // were a hundred instructions of every such trip; a record is nine stores behind one address
  uint4* r = reinterpret_cast<uint4*>(base + slot * XZ_WORDS);
  r[0] = make_uint4(p.x.n[0], p.x.n[1], p.x.n[2], p.x.n[3]);
  r[1] = make_uint4(p.x.n[4], p.x.n[5], p.x.n[6], p.x.n[7]);
  r[2] = make_uint4(p.x.n[8], p.y.n[0], p.y.n[1], p.y.n[2]);
  r[3] = make_uint4(p.y.n[3], p.y.n[4], p.y.n[5], p.y.n[6]);
  r[4] = make_uint4(p.y.n[7], p.y.n[8], p.zz.n[0], p.zz.n[1]);
  r[5] = make_uint4(p.zz.n[2], p.zz.n[3], p.zz.n[4], p.zz.n[5]);
  r[6] = make_uint4(p.zz.n[6], p.zz.n[7], p.zz.n[8], p.zzz.n[0]);
  r[7] = make_uint4(p.zzz.n[1], p.zzz.n[2], p.zzz.n[3], p.zzz.n[4]);
  r[8] = make_uint4(p.zzz.n[5], p.zzz.n[6], p.zzz.n[7], p.zzz.n[8]);
}
S2K_DEV xyzz29 xz_load(const uint32_t* __restrict__ base, size_t slot) {
  xyzz29 p;
  const uint4* r = reinterpret_cast<const uint4*>(base + slot * XZ_WORDS);
  uint32_t v[XZ_WORDS];
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const uint4 t = r[q];
    v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
  }
#pragma unroll
  for (int w = 0; w < 9; ++w) {
    p.x.n[w] = v[w];
    p.y.n[w] = v[9 + w];
    p.zz.n[w] = v[18 + w];
    p.zzz.n[w] = v[27 + w];
  }
  return p;
}
template <int PAD_KIB>
__global__ void __launch_bounds__(256, S2K_MSM_WAVES)
k_msm_accumulate(msm_parts P, uint32_t which, uint32_t nkeys, const uint32_t* __restrict__ offset,
                 const uint32_t* __restrict__ lanekey, const uint32_t* __restrict__ list, const uint32_t* __restrict__ ptw,
                 uint32_t* __restrict__ xsum) {
  // which = 0: part A (the list from offset[split_key] on; everything when there is one part), 1: part B (the list below it).
  // The split is a bucket border, so no piece is cut by it.  PAD_KIB: a block of LDS the kernel owns and never uses - 48 KiB
  // let three workgroups into a CU's 160 KiB and keep a fourth out, 72 KiB two: msm_core launches part B that way, so that
  // every SIMD keeps wave slots and registers free for the tail of part A, which runs beside it.
  if constexpr (PAD_KIB > 0) {
    __shared__ uint32_t pad[PAD_KIB * 256];
    if (which == 0xffffffffu) {                            // (never: the block only has to exist)
      pad[threadIdx.x] = nkeys;
      __syncthreads();
      xsum[0] = pad[(threadIdx.x + 1) & 255];
    }
  }
  const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
  const uint32_t split_pos = offset[P.split_key];          // (0 when there is one part)
  uint32_t L, limit, gl;
  uint64_t lo64;
  if (which) {
    if (lane >= P.nlanesB) return;
    L = P.L_B;
    lo64 = (uint64_t)lane * L;
    limit = split_pos;
    gl = lane;
  } else {
    if (lane >= P.nlanes - P.nlanesB) return;
    L = P.L_A;
    lo64 = (uint64_t)split_pos + (uint64_t)lane * L;
    limit = offset[nkeys];
    gl = P.nlanesB + lane;
  }
  if (lo64 >= limit) return;
  const uint32_t lo = (uint32_t)lo64, hi = limit - lo > L ? lo + L : limit;
  uint32_t key = lanekey[gl];                              // offset[key] <= lo < offset[key + 1] (k_msm_fine_sort)
  const uint32_t nlanes = P.nlanes;
  const uint32_t lane_slot = gl;
  uint32_t border = offset[key + 1];                       // > lo
  // the border AFTER that one, fetched ahead: some lane of a wave meets a border in two
  // trips of three, and the load of the next border was a dependent one the whole wave waited for - 11.6 % of the kernel's
  // wave cycles were spent waiting for memory (profiles/r06_msm_attempts.txt).  It is re-fetched every trip beside the next
  // record, so that the flush itself issues nothing but stores.
  uint32_t border2 = offset[key + 2 <= nkeys ? key + 2 : nkeys];
  bool open_left = offset[key] < lo;
  bool fresh = true;                                       // the next point starts a piece
  xyzz29 acc = xyzz29_from_affine(fe29_zero(), fe29_zero());
  uint32_t e_cur = list[lo], e_nxt = lo + 1 < hi ? list[lo + 1] : 0u;
  msm_rec r_cur = msm_load_rec(ptw, e_cur);
#pragma unroll 1
  for (uint32_t j = lo; j < hi; ++j) {
    // Order of a trip: the record fetched during the previous addition is unpacked FIRST,
    // then - at a border - the border after the next is asked for and the piece is flushed, then the next record is asked
    // for, then the addition runs.  The memory counter of this hardware completes in order and the compiler must assume
    // the shortest path at a merge: with the flush in front of the unpacking (rounds 3-5) a wave that met a border waited
    // for its 36 stores to reach memory before it could touch the record it had fetched a whole addition earlier.
    fe29 qx, qy;
    msm_point_of(r_cur, e_cur, qx, qy);
#pragma unroll
    for (int w = 0; w < 9; ++w) asm volatile("" : "+v"(qx.n[w]), "+v"(qy.n[w]));   // (pins the unpacking HERE: left alone, the compiler sinks it below the flush)
    if (j == border) {                                     // a bucket ends here: next non-empty bucket, flush
      const size_t slot = open_left ? (size_t)nkeys + lane_slot : (size_t)key;
      ++key;
      border = border2;                                    // (fetched in an earlier trip: no load, no wait in front of the stores)
      if (border <= j) {                                   // (an empty bucket: rare - these loads are waited for)
        do {
          ++key;
          border = offset[key + 1];
        } while (border <= j);
      }
      xz_store(xsum, slot, acc);
      open_left = false;
      fresh = true;
    }
    const msm_rec r_nxt = msm_load_rec(ptw, e_nxt);        // entry 0 when past the end: a valid address
    const uint32_t e_nn = j + 2 < hi ? list[j + 2] : 0u;
    border2 = offset[key + 2 <= nkeys ? key + 2 : nkeys];  // the border after the next, EVERY trip, beside the record's fetch (one cached word)
    if (fresh) {                                           // (a few lanes of the wave at a time: the others wait out 36 moves)
      acc = xyzz29_from_affine(qx, qy);
      fresh = false;
    } else {
      acc = xyzz29_add_affine(acc, qx, qy);
    }
    r_cur = r_nxt;
    e_cur = e_nxt;
    e_nxt = e_nn;
  }
  // the last piece: left edge if it came in from the previous range (then it may go on as well: a range inside one
  // bucket), right edge if it goes on into the next range, else a whole bucket
  const bool open_right = border > hi;
  xz_store(xsum, open_left ? (size_t)nkeys + lane_slot : (open_right ? (size_t)nkeys + nlanes + lane_slot : (size_t)key), acc);
}
// the piece in `slot` (the list entries [first, end) of one bucket) as a projective point; ZZ = 0: walked again, complete formulas
S2K_DEV pt29 msm_piece(const uint32_t* __restrict__ xsum, size_t slot, uint32_t first, uint32_t end,
                       const uint32_t* __restrict__ list, const uint32_t* __restrict__ ptw) {
  const xyzz29 x = xz_load(xsum, slot);
  if (!fe29_is_zero(x.zz)) return xyzz29_to_pt29(x);
  pt29 acc = pt29_identity();
#pragma unroll 1
  for (uint32_t j = first; j < end; ++j) {
    const uint32_t e = list[j];
    fe29 qx, qy;
    msm_point_of(msm_load_rec(ptw, e), e, qx, qy);
    acc = pt29_add_mixed(acc, qx, fe29_normalize_weak(qy));      // (the complete formulas take an addend of one unit)
  }
  return acc;
}
// the ranges of the part a key belongs to: first lane, list position of its first range, entries per range
struct msm_part_of {
  uint32_t lane0, start, L;
};
S2K_DEV msm_part_of msm_part_for(const msm_parts& P, uint32_t key, const uint32_t* __restrict__ offset) {
  msm_part_of r;
  const bool in_b = key < P.split_key;
  r.lane0 = in_b ? 0u : P.nlanesB;
  r.start = in_b ? 0u : offset[P.split_key];
  r.L = in_b ? P.L_B : P.L_A;
  return r;
}
// piece of the bucket [b, e) in range k (a global lane number)
S2K_DEV pt29 msm_piece_of(const uint32_t* __restrict__ xsum, uint32_t nkeys, uint32_t nlanes, const msm_part_of& pp, uint32_t b,
                          uint32_t e, uint32_t k, uint32_t k_lo, const uint32_t* __restrict__ list, const uint32_t* __restrict__ ptw) {
  const uint64_t r0 = (uint64_t)pp.start + (uint64_t)(k - pp.lane0) * pp.L, r1 = r0 + pp.L;
  const uint32_t first = b > r0 ? b : (uint32_t)r0, end = e < r1 ? e : (uint32_t)r1;
  // the first range's piece is a right edge (the bucket starts in it or at its border and goes on), the others' left edges
  return msm_piece(xsum, k == k_lo ? (size_t)nkeys + nlanes + k : (size_t)nkeys + k, first, end, list, ptw);
}
// one lane per key: sums[key] = the bucket as a projective point - the identity for an empty bucket, the converted piece
// for a bucket inside one range, the sum of its pieces for a bucket that crosses range borders (one addition for an ordinary
// bucket; buckets spread over more than STITCH_SERIAL ranges are queued for k_msm_stitch_big)
// (WAVES = 3: 166 registers, nothing spilled - 66 us for the whole key range; WAVES = 4: 128 registers, 29 of them spilled, 76 us,
// but a wave fits beside the lower part's bucket pass: what the two-part flow launches on its second stream)
template <int WAVES>
__global__ void __launch_bounds__(256, WAVES)
k_msm_stitch(msm_parts P, uint32_t nkeys, size_t stride, const uint32_t* __restrict__ offset,
             const uint32_t* __restrict__ xsum, const uint32_t* __restrict__ list, const uint32_t* __restrict__ ptw,
             uint32_t* __restrict__ sums, uint32_t* __restrict__ big /* [0] count, [1 ..] keys */, uint32_t key_lo, uint32_t key_hi) {
  __builtin_amdgcn_s_setprio(3);   // (beside the lower part's bucket pass these waves issue first: they are few and the call waits for them)
  const uint32_t key = key_lo + blockIdx.x * 256 + threadIdx.x;      // the keys [key_lo, key_hi) of this launch
  if (key >= key_hi) return;
  const uint32_t b = offset[key], e = offset[key + 1];
  if (b == e) {
    pt_store(sums, stride, key, pt29_identity());
    return;
  }
  const msm_part_of pp = msm_part_for(P, key, offset);
  const uint32_t k_lo = pp.lane0 + (b - pp.start) / pp.L, k_hi = pp.lane0 + (e - 1 - pp.start) / pp.L;
  if (k_lo == k_hi) {
    pt_store(sums, stride, key, msm_piece(xsum, key, b, e, list, ptw));
    return;
  }
  if (k_hi - k_lo > STITCH_SERIAL) {
    const uint32_t pos = atomicAdd(&big[0], 1u);
    if (pos < STITCH_BIG_CAP) {
      big[1 + pos] = key;
      return;
    }                                       // more oversized buckets than the queue holds: serial after all
  }
  pt29 r = msm_piece_of(xsum, nkeys, P.nlanes, pp, b, e, k_lo, k_lo, list, ptw);
#pragma unroll 1
  for (uint32_t k = k_lo + 1; k <= k_hi; ++k) r = pt29_add(r, msm_piece_of(xsum, nkeys, P.nlanes, pp, b, e, k, k_lo, list, ptw));
  pt_store(sums, stride, key, r);
}
// one workgroup per queued bucket: the threads take the pieces round robin, then a tree in LDS
__global__ void __launch_bounds__(256)
k_msm_stitch_big(msm_parts P, uint32_t nkeys, size_t stride, const uint32_t* __restrict__ offset,
                 const uint32_t* __restrict__ xsum, const uint32_t* __restrict__ list, const uint32_t* __restrict__ ptw,
                 uint32_t* __restrict__ sums, const uint32_t* __restrict__ big) {
  __shared__ uint32_t sh[PT_WORDS][128];
  const uint32_t nbig = big[0] < STITCH_BIG_CAP ? big[0] : STITCH_BIG_CAP;
  for (uint32_t q = blockIdx.x; q < nbig; q += gridDim.x) {
    const uint32_t key = big[1 + q];
    const uint32_t b = offset[key], e = offset[key + 1];
    const msm_part_of pp = msm_part_for(P, key, offset);
    const uint32_t k_lo = pp.lane0 + (b - pp.start) / pp.L, k_hi = pp.lane0 + (e - 1 - pp.start) / pp.L;
    pt29 r = pt29_identity();
#pragma unroll 1
    for (uint32_t k = k_lo + threadIdx.x; k <= k_hi; k += 256)
      r = pt29_add(r, msm_piece_of(xsum, nkeys, P.nlanes, pp, b, e, k, k_lo, list, ptw));
    for (uint32_t half = 128; half >= 1; half >>= 1) {
      __syncthreads();
      if (threadIdx.x >= half && threadIdx.x < 2 * half) pt_store(&sh[0][0], 128, threadIdx.x - half, r);
      __syncthreads();
      if (threadIdx.x < half) r = pt29_add(r, pt_load(&sh[0][0], 128, threadIdx.x));
    }
    if (threadIdx.x == 0) pt_store(sums, stride, key, r);
    __syncthreads();
  }
}

// chunk (s, j): keys [CHUNK j, CHUNK j + CHUNK) of slot s, i.e. the magnitudes m = off + CHUNK j + t + 1, t = 0 .. CHUNK - 1
// (off = nb for the extra slot of the top window, else 0)  ->  sum_t m * B_t = sum_t (t + 1) B_t + (off + CHUNK j) * S.
// This kernel, the tree and the Horner tail below are serial chains of group operations on a chip that has nothing else
// to do: they run the quad-spread formulas (pt29q.h), FOUR LANES PER CHUNK, each holding one coordinate.
__global__ void __launch_bounds__(256)
k_msm_reduce(msm_geom g, const uint32_t* __restrict__ sums, size_t stride, uint32_t* __restrict__ partial, uint32_t slot_lo, uint32_t slot_hi) {
  const size_t th = (size_t)blockIdx.x * 256 + threadIdx.x, id = (size_t)slot_lo * g.nchunk + (th >> 2);   // the slots [slot_lo, slot_hi)
  const uint32_t q = (uint32_t)th & 3u, cc = q < 2 ? q : 2u;
  const size_t nslots = (size_t)g.nslot * g.nchunk;
  if (id >= (size_t)slot_hi * g.nchunk) return;               // (whole quads leave together)
  const uint32_t sl = (uint32_t)(id / g.nchunk), j = (uint32_t)(id % g.nchunk);
  const int CHUNK_LOG2 = (int)g.chunk_log2, CHUNK = 1 << CHUNK_LOG2;
  const size_t base = (size_t)sl * g.nb + (size_t)j * CHUNK;
  fe29 run = pt29q_identity(q), tot = pt29q_identity(q);
  fe29 nxt = ptq_load(sums, stride, base + CHUNK - 1, cc);
#pragma unroll 1
  for (int t = CHUNK - 1; t >= 0; --t) {
    const fe29 cur = nxt;
    if (t > 0) nxt = ptq_load(sums, stride, base + t - 1, cc);     // in flight during the two additions
    run = pt29q_add(run, cur, q);
    tot = pt29q_add(tot, run, q);
  }
  // tot += (off + CHUNK j) * S :  jj * S by double-and-add, then CHUNK_LOG2 doublings
  const uint32_t jj = j + (sl == g.nw ? g.nchunk : 0u);
  fe29 m = pt29q_identity(q);
#pragma unroll 1
  for (int bit = (int)g.c - CHUNK_LOG2 - 1; bit >= 0; --bit) {   // jj < 2 nchunk = 2^(c - CHUNK_LOG2)
    m = pt29q_double(m, q);
    const fe29 s = pt29q_add(m, run, q);
    m = fe29_pick((jj >> bit) & 1u, m, s);
  }
#pragma unroll 1
  for (int t = 0; t < CHUNK_LOG2; ++t) m = pt29q_double(m, q);
  tot = pt29q_add(tot, m, q);
  if (q < 3) ptq_store(partial, nslots + 1, id, cc, tot);     // (plane stride nslots + 1: the Horner tail's carry slot follows)
}

// tree sum of a slot's chunk results, partial[s][0..nchunk) -> partial[s][0], in two launches of 256-thread workgroups:
// `span` consecutive slots are folded into the first one by each workgroup, a quad per addition
__global__ void __launch_bounds__(256) k_msm_tree(uint32_t nslots_total, uint32_t span, uint32_t stride_slots,
                                                  uint32_t* __restrict__ partial, uint32_t block_lo) {
  // workgroup b folds slots [b * span * stride_slots, ...) taken every stride_slots (b counted from block_lo)
  const size_t base = (size_t)(block_lo + blockIdx.x) * span * stride_slots;
  const uint32_t q = threadIdx.x & 3u, cc = q < 2 ? q : 2u, pair = threadIdx.x >> 2;
  for (uint32_t half = span >> 1; half >= 1; half >>= 1) {
    for (uint32_t t = pair; t < half; t += 64) {
      const size_t ia = base + (size_t)t * stride_slots, ib = base + (size_t)(t + half) * stride_slots;
      const fe29 r = pt29q_add(ptq_load(partial, nslots_total, ia, cc), ptq_load(partial, nslots_total, ib, cc), q);
      if (q < 3) ptq_store(partial, nslots_total, ia, cc, r);
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------
// Horner over the window sums, then the 65-byte record: 128 - c doublings, a serial chain on one wave (its sixteen
// quads all run the same recurrence), lane 0 writes the result.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_msm_final(msm_geom g, uint32_t* __restrict__ partial, uint8_t* __restrict__ out65, int affine,
                                                  uint32_t w_from, uint32_t w_to) {
  // out65: the 65-byte record of the sum (all zero for the identity); affine == 0: only identity or not is wanted
  // (first byte 0x00 / 0x04, coordinates left zero): the whole-batch BIP-340 verdict skips the inversion.
  // The recurrence runs over the windows w_from - 1 down to w_to.  w_from == nw: from the top (the top window's two slots
  // first); otherwise it continues from the point an earlier launch has left in the carry slot.  w_to > 0: the point goes
  // to the carry slot for the launch that continues (two-part flow, msm_core); w_to == 0: it is the result.
  const uint32_t q = threadIdx.x & 3u, cc = q < 2 ? q : 2u;
  const size_t nslots = (size_t)g.nslot * g.nchunk, carry = nslots;          // (one slot beyond the chunk results)
  fe29 accq;
  int w;
  if (w_from == g.nw) {   // the two slots of the top window carry the same weight
    accq = pt29q_add(ptq_load(partial, nslots + 1, (size_t)g.nw * g.nchunk, cc), ptq_load(partial, nslots + 1, (size_t)(g.nw - 1) * g.nchunk, cc), q);
    w = (int)g.nw - 2;
  } else {
    accq = ptq_load(partial, nslots + 1, carry, cc);
    w = (int)w_from - 1;
  }
#pragma unroll 1
  for (; w >= (int)w_to; --w) {
    const fe29 add = ptq_load(partial, nslots + 1, (size_t)w * g.nchunk, cc);
#pragma unroll 1
    for (uint32_t t = 0; t < g.c; ++t) accq = pt29q_double(accq, q);
    accq = pt29q_add(accq, add, q);
  }
  if (w_to > 0) {
    if (threadIdx.x < 3) ptq_store(partial, nslots + 1, carry, cc, accq);
    return;
  }
  const pt29 acc = pt29q_gather(accq);
  if (threadIdx.x != 0) return;
  if (fe29_is_zero(acc.z)) {
    for (int i = 0; i < 65; ++i) out65[i] = 0;
    return;
  }
  if (!affine) {
    out65[0] = 0x04;
    for (int i = 1; i < 65; ++i) out65[i] = 0;
    return;
  }
  fe29 zi = fe29_inv_gcd(fe29_normalize_weak(acc.z));   // one lane, serial: safegcd (fe29_inv.h) is 0.04 ms where the Fermat chain was 0.15
  fe29 x = fe29_normalize(fe29_mul(acc.x, zi)), y = fe29_normalize(fe29_mul(acc.y, zi));
  uint32_t xw[8], yw[8];
  fe29_to_words(xw, x);
  fe29_to_words(yw, y);
  out65[0] = 0x04;
  store_be32_unaligned(out65 + 1, xw);
  store_be32_unaligned(out65 + 33, yw);
}

// ---------------------------------------------------------------------------------------
// The bucket reduction for 16-bit windows since round 5: no running sums, no offset multiplications.
// The 9 x 32768 keys are 1152 ROWS of 256 consecutive keys; the 128 rows of a window form a grid, magnitude of key (h, l)
// = 256 h + l + 1 (the top window's second slot: rows 128..255 of its grid), so
//     S_w = sum_key m B = 256 * sum_h h Row_h  +  sum_l (l + 1) Col_l ,   Row_h = sum_l B[h][l],  Col_l = sum_h B[h][l]:
// every bucket goes into ONE row sum and ONE column sum (2 additions per bucket, all of them leaves of trees: no addition
// waits for a running sum), and what is left are weighted sums of 128 or 256 points, taken bit plane by bit plane
// (sum_i w_i E_i = sum_b 2^b sum_{i : bit b of w_i} E_i: plain sums again, then eight doublings).
//   k_msm_fold         level 1, one lane per addition (pt29_add), trees through LDS: 1152 row sums and 9 x 256 column sums
//                      (per block of 128 rows) -> lvl[0 .. 3456)
//   k_msm_planes       level 2, ONE WAVE per addition (fe29r.h: the operations are few and wait for each other): a workgroup
//                      of 16 waves per (weighted sum, bit): 8 members per wave, then a tree through LDS -> lvl[PL ..)
//   k_msm_plane_horner one wave per weighted sum: sum_b 2^b plane_b -> lvl[WS ..)
//   k_msm_final16      S_w = 256 RW_w + CW_w on eight waves side by side, then Horner over the windows (112 doublings, 7 additions)
//                      on one wave, row arithmetic
// ---------------------------------------------------------------------------------------
constexpr uint32_t FOLD_NROWS = 1152, FOLD_NBLK = 9, FOLD_COL0 = FOLD_NROWS, FOLD_PL0 = FOLD_COL0 + FOLD_NBLK * 256;   // 3456
constexpr uint32_t FOLD_NWS = 17, FOLD_BITS = 9, FOLD_WS0 = FOLD_PL0 + FOLD_NWS * FOLD_BITS, FOLD_END = FOLD_WS0 + FOLD_NWS;
constexpr uint32_t FOLD_ROW_BLOCKS = FOLD_NROWS / 2, FOLD_COL_BLOCKS = FOLD_NBLK * 64;

// The narrow levels of a workgroup's trees: `groups` independent sums whose partials lie in LDS, partial i of group g in slot
// g * GS + i * IS.  From 64 additions per level down a lane per addition would leave three quarters of the workgroup idle
// and pay a lone lane's 1800 instructions per level: here a QUAD does each addition (pt29q.h: 820).
S2K_DEV void fold_quad_tree(uint32_t* sh, uint32_t groups, uint32_t GS, uint32_t IS, uint32_t half0, uint32_t t) {
  const uint32_t q = t & 3u, cc = q < 2 ? q : 2u, qd = t >> 2;
  for (uint32_t half = half0; half >= 1; half >>= 1) {
    __syncthreads();
    if (qd < groups * half) {                           // (whole quads)
      const uint32_t g = qd / half, i = qd % half, sa = g * GS + i * IS, sb = g * GS + (i + half) * IS;
      const fe29 r = pt29q_add(ptq_load(sh, 128, sa, cc), ptq_load(sh, 128, sb, cc), q);
      if (q < 3) ptq_store(sh, 128, sa, cc, r);
    }
  }
  __syncthreads();
}
__global__ void __launch_bounds__(256, 4)      // 128 registers: 4 waves per SIMD, 1024 of the 1152 workgroups resident at once
k_msm_fold(const uint32_t* __restrict__ sums, size_t stride, uint32_t* __restrict__ lvl, size_t lstride, uint32_t s_lo, uint32_t ns) {
  __builtin_amdgcn_s_setprio(3);   // (beside the lower part's bucket pass these waves issue first: they are few and the call waits for them)
  // the slots [s_lo, s_lo + ns) of this launch: 64 row blocks and 64 column blocks per slot (the whole key range: s_lo = 0, ns = 9)
  __shared__ uint32_t sh[PT_WORDS * 128];
  const uint32_t t = threadIdx.x;
  if (blockIdx.x < 64 * ns) {
    // two rows per workgroup, 128 threads each: two keys per thread and one level of the tree a lane per addition (128
    // additions per level), then 128 partials in LDS (row r: slots 64 r ..) and six levels by quads
    const uint32_t rb = 64 * s_lo + blockIdx.x;          // row block: rows 2 rb, 2 rb + 1
    const uint32_t half_id = t >> 7, u = t & 127u, row = 2 * rb + half_id;
    const size_t k0 = (size_t)row * 256 + u;
    pt29 acc = pt29_add(pt_load(sums, stride, k0), pt_load(sums, stride, k0 + 128));
    if (u >= 64) pt_store(sh, 128, half_id * 64 + (u - 64), acc);
    __syncthreads();
    if (u < 64) acc = pt29_add(acc, pt_load(sh, 128, half_id * 64 + u));
    __syncthreads();
    if (u < 64) pt_store(sh, 128, half_id * 64 + u, acc);
    fold_quad_tree(sh, 2, 64, 1, 32, t);
    if (t < 2) pt_store(lvl, lstride, 2 * rb + t, pt_load(sh, 128, t * 64));
  } else {
    // block of 128 rows x 4 columns: thread (hg, lc) sums rows 2 hg, 2 hg + 1 of column 4 lb + lc; the tree over the 64 hg:
    // one level a lane per addition (128 additions), then 128 partials in LDS (slot 4 hg + lc) and five levels by quads.
    // (Eight columns and four rows per thread had two more additions on every lane's chain: the column blocks ended 15 us
    // after the row blocks.)
    // Which column block a workgroup takes is XCD-aware: a wave reads 16 rows x 16 bytes per word, eight neighbouring column
    // blocks share every 128-byte line, and consecutive workgroups go to different XCDs (blockIdx % 8), each with an L2 of its
    // own - taken in blockIdx order every line was fetched by up to eight L2s (283 MB per launch, 2.2 M L2 misses; now 62 MB,
    // 0.5 M; the kernel 69.5 -> 63.4 us, profiles/r06_msm_attempts.txt section 10).  So
    // XCD x takes the 8 ns column blocks [8 ns x, 8 ns (x + 1)), neighbours at the same time.
    const uint32_t ci = blockIdx.x - 64 * ns, cb = (ci & 7u) * (8u * ns) + (ci >> 3);
    const uint32_t blk = s_lo + (cb >> 6), lb = cb & 63u, hg = t >> 2, lc = t & 3u;
    const size_t k0 = ((size_t)blk * 128 + 2 * hg) * 256 + 4 * lb + lc;
    pt29 acc = pt29_add(pt_load(sums, stride, k0), pt_load(sums, stride, k0 + 256));
    if (hg >= 32) pt_store(sh, 128, (hg - 32) * 4 + lc, acc);
    __syncthreads();
    if (hg < 32) acc = pt29_add(acc, pt_load(sh, 128, hg * 4 + lc));
    __syncthreads();
    if (hg < 32) pt_store(sh, 128, hg * 4 + lc, acc);
    fold_quad_tree(sh, 4, 1, 4, 16, t);
    if (t < 4) pt_store(lvl, lstride, FOLD_COL0 + blk * 256 + 4 * lb + t, pt_load(sh, 128, t));
  }
}

// weighted sum ws: 0..7 = the row sums of window ws (weights h; window 7 has 256 rows), 8..16 = the column sums of the block
// of rows ws - 8 (weights l + 1).  Member m of bit b: the m-th weight with bit b set.
S2K_DEV uint32_t fold_members(uint32_t ws, uint32_t b) {
  if (ws < 8) {
    const uint32_t H = ws == 7 ? 256u : 128u;
    return (1u << b) < H ? H / 2 : 0u;
  }
  return b < 8 ? 128u : 1u;      // weights 1 .. 256: bit 8 is the weight 256 alone
}
S2K_DEV uint32_t fold_ws_of(uint32_t idx, uint32_t w_lo, uint32_t nwp, uint32_t s_lo) { return idx < nwp ? w_lo + idx : 8u + s_lo + (idx - nwp); }
S2K_DEV uint32_t fold_member_slot(uint32_t ws, uint32_t b, uint32_t m) {
  const uint32_t w = b < 8 ? (((m >> b) << (b + 1)) | (1u << b) | (m & ((1u << b) - 1u))) : 256u;
  return ws < 8 ? ws * 128 + w : FOLD_COL0 + (ws - 8) * 256 + (w - 1);
}
__global__ void __launch_bounds__(1024)
k_msm_planes(uint32_t* __restrict__ lvl, size_t lstride, uint32_t w_lo, uint32_t nwp, uint32_t s_lo) {
  __builtin_amdgcn_s_setprio(3);   // (beside the lower part's bucket pass these waves issue first: they are few and the call waits for them)
  // the weighted sums of this launch: the row sums of the windows [w_lo, w_lo + nwp), then the column sums of the slots from s_lo
  // on (grid: (windows + slots) x FOLD_BITS; everything: 0, 8, 0 and 17 x 9 workgroups)
  __shared__ uint32_t sh[PT_WORDS * 8];
  const uint32_t idx = blockIdx.x / FOLD_BITS, ws = fold_ws_of(idx, w_lo, nwp, s_lo);
  const uint32_t b = blockIdx.x % FOLD_BITS, wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const fer_consts k = fer_setup(lane);
  const uint32_t M = fold_members(ws, b);
  pt29r acc = pt29r_identity(k);
  if (wave < M) {
    acc = pt29r_load(lvl, lstride, fold_member_slot(ws, b, wave), k);
#pragma unroll 1
    for (uint32_t m = wave + 16; m < M; m += 16) acc = pt29r_add(acc, pt29r_load(lvl, lstride, fold_member_slot(ws, b, m), k), k);
  }
  for (uint32_t half = 8; half >= 1; half >>= 1) {      // (wave-uniform branches: whole waves store, load and add)
    __syncthreads();
    if (wave >= half && wave < 2 * half) pt29r_store(sh, 8, wave - half, acc, k);
    __syncthreads();
    if (wave < half) acc = pt29r_add(acc, pt29r_load(sh, 8, wave, k), k);
  }
  if (wave == 0) pt29r_store(lvl, lstride, FOLD_PL0 + ws * FOLD_BITS + b, acc, k);
}
// one wave per weighted sum: sum_b 2^b plane_b
__global__ void __launch_bounds__(64)
k_msm_plane_horner(uint32_t* __restrict__ lvl, size_t lstride, uint32_t w_lo, uint32_t nwp, uint32_t s_lo) {
  __builtin_amdgcn_s_setprio(3);   // (beside the lower part's bucket pass these waves issue first: they are few and the call waits for them)
  const uint32_t ws = fold_ws_of(blockIdx.x, w_lo, nwp, s_lo);
  const fer_consts k = fer_setup(threadIdx.x);
  int b = (int)FOLD_BITS - 1;
  while (b > 0 && fold_members(ws, (uint32_t)b) == 0) --b;
  pt29r acc = pt29r_load(lvl, lstride, FOLD_PL0 + ws * FOLD_BITS + b, k);
#pragma unroll 1
  for (--b; b >= 0; --b) {
    acc = pt29r_double(acc, k);
    acc = pt29r_add(acc, pt29r_load(lvl, lstride, FOLD_PL0 + ws * FOLD_BITS + b, k), k);
  }
  pt29r_store(lvl, lstride, FOLD_WS0 + ws, acc, k);
}
// result = sum_w 2^(16 w) S_w, S_w = 256 RW_w + CW_w (CW_7 = the column sums of both blocks of the top window).  The waves
// [w_lo, w_hi) form their S_w side by side (eight doublings and one or two additions each), then wave w_hi - 1 runs the
// recurrence over those windows: 16 doublings and an addition per window, all waiting for each other.
// One launch: w_lo = 0, w_hi = 8.  Two parts (msm_core): the launch of the upper windows ends with 16 more doublings and leaves
// its point in the carry slot (carry_out); the launch of the lower windows starts from carry + S_(w_hi - 1) (carry_in).
__global__ void __launch_bounds__(512)
k_msm_final16(uint32_t* __restrict__ lvl, size_t lstride, uint8_t* __restrict__ out65, int affine, uint32_t w_lo, uint32_t w_hi,
              int carry_in, int carry_out) {
  __builtin_amdgcn_s_setprio(3);   // (beside the lower part's bucket pass these waves issue first: they are few and the call waits for them)
  __shared__ uint32_t sh[PT_WORDS * 8];
  const uint32_t wave = threadIdx.x >> 6, top = w_hi - 1u;
  const fer_consts k = fer_setup(threadIdx.x & 63u);
  const bool active = wave >= w_lo && wave < w_hi;         // (wave-uniform)
  pt29r accr = pt29r_identity(k);
  if (active) {
    accr = pt29r_load(lvl, lstride, FOLD_WS0 + wave, k);
#pragma unroll 1
    for (int t = 0; t < 8; ++t) accr = pt29r_double(accr, k);
    accr = pt29r_add(accr, pt29r_load(lvl, lstride, FOLD_WS0 + 8 + wave, k), k);
    if (wave == 7) accr = pt29r_add(accr, pt29r_load(lvl, lstride, FOLD_WS0 + 16, k), k);
    if (wave != top) pt29r_store(sh, 8, wave, accr, k);
  }
  __syncthreads();
  if (wave != top) return;                     // (whole waves)
  if (carry_in) accr = pt29r_add(accr, pt29r_load(lvl, lstride, FOLD_END, k), k);
#pragma unroll 1
  for (int w = (int)top - 1; w >= (int)w_lo; --w) {
#pragma unroll 1
    for (int t = 0; t < 16; ++t) accr = pt29r_double(accr, k);
    accr = pt29r_add(accr, pt29r_load(sh, 8, (uint32_t)w, k), k);
  }
  if (carry_out) {
#pragma unroll 1
    for (int t = 0; t < 16; ++t) accr = pt29r_double(accr, k);
    pt29r_store(lvl, lstride, FOLD_END, accr, k);
    return;
  }
  const pt29 acc = pt29r_gather(accr, k);
  if ((threadIdx.x & 63u) != 0) return;
  if (fe29_is_zero(acc.z)) {
    for (int i = 0; i < 65; ++i) out65[i] = 0;
    return;
  }
  if (!affine) {
    out65[0] = 0x04;
    for (int i = 1; i < 65; ++i) out65[i] = 0;
    return;
  }
  fe29 zi = fe29_inv_gcd(fe29_normalize_weak(acc.z));
  fe29 x = fe29_normalize(fe29_mul(acc.x, zi)), y = fe29_normalize(fe29_mul(acc.y, zi));
  uint32_t xw[8], yw[8];
  fe29_to_words(xw, x);
  fe29_to_words(yw, y);
  out65[0] = 0x04;
  store_be32_unaligned(out65 + 1, xw);
  store_be32_unaligned(out65 + 33, yw);
}

size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

struct msm_ws {
  msm_geom g;
  size_t nkeys, nslots;
  uint32_t *status, *matrix, *offset, *bsum, *scw, *ptw, *list, *sums, *xsum, *partial, *big, *lanekey;
  size_t cap;          // term capacity the workspace was carved for = plane stride of scw (a call may run on fewer terms)
  uint32_t lanes_cap;  // bucket pass: most ranges any term count up to cap can make
  size_t sum_stride;   // slots of `sums`: nkeys buckets, then the left and the right edge piece of every range
  void* pairs;
  bool wide;           // two-word pairs (term indices beyond 23 bits)
  uint32_t ncoarse, nsortblk, nblk_pad, sort_terms;
  uint8_t* flag;
  size_t zero_bytes;   // status + size bins + coarse matrix, contiguous from the start
  uint8_t* aux;        // extra caller-requested scratch
};

static uint32_t msm_lanes_target() {   // tuning hook: S2K_MSM_LANES overrides the lanes that fill the chip once
  static const uint32_t v = [] {
    const char* e = getenv("S2K_MSM_LANES");
    return e && atoi(e) > 0 ? (uint32_t)atoi(e) : MSM_LANES;
  }();
  return v;
}
// carve the MSM workspace for up to n terms (+ aux_bytes of scratch for the caller)
int msm_setup(s2k_ctx* ctx, size_t n, size_t aux_bytes, msm_ws& m) {
  msm_geom& g = m.g;
  // window width: measured (tools/msm_sweep.py, MI355X, ms for 2^11 .. 2^20 inputs = twice as many terms):
  // 16-bit windows from 2^13 terms on: 0.83 0.75 0.75 0.76 0.78 0.80 0.89 1.01 1.31 1.95; 12-bit windows up to
  // 2^17 terms: 0.84 1.04 1.26 1.78 2.70 then as above (big buckets serialise on their lanes); the floor of
  // about 0.75 ms is the reduction of 8 x 65535 buckets
  static const int c16_from_log2 = [] {   // tuning hook: S2K_MSM_C16_FROM_LOG2 overrides the measured default
    const char* e = getenv("S2K_MSM_C16_FROM_LOG2");
    return e ? atoi(e) : 13;
  }();
  g.c = n >= ((size_t)1 << c16_from_log2) ? 16 : (n >= 256 ? 12 : 8);
  g.nw = (SCALAR_BITS + g.c - 1) / g.c;
  g.nb = 1u << (g.c - 1);
  g.nslot = g.nw + 1;
  static const int chunk_log2 = [] {       // tuning hook
    const char* e = getenv("S2K_MSM_CHUNK_LOG2");
    int v = e ? atoi(e) : S2K_MSM_CHUNK_LOG2;
    return v < 1 ? 1 : (v > 6 ? 6 : v);
  }();
  g.chunk_log2 = (uint32_t)chunk_log2;
  g.nchunk = g.nb >> g.chunk_log2;
  m.nkeys = align_up((size_t)g.nslot * g.nb, FINE);   // (c = 8: 17 slots of 128 keys, padded; the padding keys stay empty)
  m.nslots = (size_t)g.nslot * g.nchunk;
  size_t off = 0;
  auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
  m.cap = n;
  {
    const size_t pairs_max = n * (size_t)g.nw, by_len = (pairs_max + MSM_L_MIN - 1) / MSM_L_MIN;
    m.lanes_cap = 2 * ((uint32_t)(by_len < msm_lanes_target() ? by_len : msm_lanes_target()) + 1);   // (two parts, each with up to the target)
    m.sum_stride = align_up(m.nkeys + 2 * (size_t)m.lanes_cap, 64);
  }
  // status word, then the queue of oversized buckets (counter + keys), zeroed with the counters
  m.ncoarse = (uint32_t)(m.nkeys >> FINE_BITS);
  m.sort_terms = (n <= ((size_t)1 << 22) && g.nw * SORT_TERMS_STAGED <= STAGE_PAIRS) ? SORT_TERMS_STAGED : SORT_TERMS_DIRECT;
  m.nsortblk = (uint32_t)((n + m.sort_terms - 1) / m.sort_terms);
  m.nblk_pad = (m.nsortblk + 1 + 1023) / 1024 * 1024;   // one spare column: the scan total lands in it
  const size_t mat_words = (size_t)m.ncoarse * m.nblk_pad;
  size_t o_status = carve(256), o_big = carve(2 * (STITCH_BIG_CAP + 1) * 4), o_matrix = carve((mat_words + 1) * 4),
         o_offset = carve((m.nkeys + 1) * 4), o_bsum = carve((mat_words / 1024 + 1) * 4),
         o_pairs = carve(n * (size_t)g.nw * 8), o_scw = carve(n * SCW_WORDS * 4), o_ptw = carve(n * 16 * 4), o_flag = carve(n),
         o_list = carve(n * (size_t)g.nw * 4), o_sums = carve(m.sum_stride * PT_WORDS * 4),
         o_partial = carve((m.nslots + 1) * PT_WORDS * 4), o_lanekey = carve((size_t)m.lanes_cap * 4),
         o_xsum = carve(m.sum_stride * XZ_WORDS * 4), o_aux = carve(aux_bytes);
  int rc = ctx_reserve(ctx, &ctx->msm_ws, &ctx->msm_ws_bytes, off);
  if (rc) return rc;
  uint8_t* ws = (uint8_t*)ctx->msm_ws;
  m.status = (uint32_t*)(ws + o_status);
  m.matrix = (uint32_t*)(ws + o_matrix);
  m.pairs = ws + o_pairs;
  static const bool force_wide = getenv("S2K_MSM_WIDE_PAIRS") != nullptr;   // test hook: the two-word pairs at any size
  m.wide = force_wide || n > ((size_t)1 << NARROW_TERM_BITS);
  m.offset = (uint32_t*)(ws + o_offset);
  m.bsum = (uint32_t*)(ws + o_bsum);
  m.scw = (uint32_t*)(ws + o_scw);
  m.ptw = (uint32_t*)(ws + o_ptw);
  m.flag = ws + o_flag;
  m.list = (uint32_t*)(ws + o_list);
  m.sums = (uint32_t*)(ws + o_sums);
  m.partial = (uint32_t*)(ws + o_partial);
  m.big = (uint32_t*)(ws + o_big);
  m.lanekey = (uint32_t*)(ws + o_lanekey);
  m.xsum = (uint32_t*)(ws + o_xsum);
  m.aux = ws + o_aux;
  m.zero_bytes = o_offset;   // status, the queue of oversized buckets and the coarse matrix
  return S2K_OK;
}

// buckets -> result, given scw / ptw / flag already filled and status/count/cursor zeroed
// the second stream of a multi-scalar call (the point half of the front end; the upper part's tail in the two-part flow)
static int msm_second_stream(s2k_ctx* ctx, hipStream_t* out) {
  if (!ctx->s_msm_tail) {
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) hi = 0;
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->s_msm_tail, hipStreamNonBlocking, hi));
  }
  *out = ctx->s_msm_tail;
  return S2K_OK;
}
int msm_core(s2k_ctx* ctx, hipStream_t st, size_t n, msm_ws& m, uint8_t* d_out65, bool affine = true) {
  const msm_geom& g = m.g;
  if (n > m.cap) return fail(ctx, S2K_ERR_ARG, "internal: more terms than the multiscalar workspace was carved for");
  const uint32_t nsortblk = (uint32_t)((n + m.sort_terms - 1) / m.sort_terms);  // <= m.nsortblk: the matrix columns beyond stay zero
  static const bool fold_off = getenv("S2K_MSM_OLD_REDUCE") != nullptr;      // A/B hook: the round-3 reduction (chunks of 8 buckets, quads)
  const bool fold16 = g.c == 16 && !fold_off && m.nslots + 1 > FOLD_END;
  // Two parts, top-down (16-bit windows, large inputs): the bucket pass of the windows from `wsplit` up
  // first; their stitching, row / column sums, bit planes and their share of the recurrence over the windows (all of it
  // serial or latency bound: 0.3 ms for the whole key range) then run on a second stream BESIDE the bucket pass of the lower
  // windows, which is launched three workgroups per CU (a block of LDS it never touches keeps the fourth out) so that every
  // SIMD has a wave slot and 128 VGPRs free for them; what is left behind the bucket pass is the tail of `wsplit` windows,
  // and the recurrence continues from the point the upper part left in the carry slot.  Each part is cut into its own
  // ranges (msm_parts), so either fills the chip by itself - round 3's attempt gave a part a SUBSET of the ranges of the
  // whole list and no room for the tail kernels, and was slower than one part.  S2K_MSM_SPLIT_WINDOW overrides (0: one part).
  static const int split_env = [] {
    const char* e = getenv("S2K_MSM_SPLIT_WINDOW");
    return e ? atoi(e) : S2K_MSM_SPLIT_DEFAULT;
  }();
  static const uint32_t b_wgs_per_cu = [] {   // S2K_MSM_B_WGS: workgroups per CU of the lower part's bucket pass (2 .. 4)
    const char* e = getenv("S2K_MSM_B_WGS");
    const int v = e ? atoi(e) : 2;
    return (uint32_t)(v < 2 ? 2 : (v > 4 ? 4 : v));
  }();
  const uint32_t wsplit = (fold16 && n >= ((size_t)1 << 17) && split_env > 0 && split_env < (int)g.nw) ? (uint32_t)split_env : 0u;
  // ranges of the bucket pass: as long as it takes for the lanes of a part to fill the chip once
  msm_parts P;
  {
    const size_t pairs_a = n * (size_t)(g.nw - wsplit), pairs_b = n * (size_t)wsplit;
    const size_t target_a = msm_lanes_target(), target_b = (size_t)msm_lanes_target() * b_wgs_per_cu / 4;
    size_t la = (pairs_a + target_a - 1) / target_a, lb = wsplit ? (pairs_b + target_b - 1) / target_b : MSM_L_MIN;
    if (la < MSM_L_MIN) la = MSM_L_MIN;
    if (lb < MSM_L_MIN) lb = MSM_L_MIN;
    P.split_key = wsplit * g.nb;
    P.L_A = (uint32_t)la;
    P.L_B = (uint32_t)lb;
    P.nlanesB = wsplit ? (uint32_t)((pairs_b + lb - 1) / lb) : 0u;
    P.nlanes = P.nlanesB + (uint32_t)((pairs_a + la - 1) / la);                  // <= m.lanes_cap
  }
  if (P.nlanes > m.lanes_cap) return fail(ctx, S2K_ERR_ARG, "internal: more ranges than the multiscalar workspace was carved for");
  const uint32_t nlanes_a = P.nlanes - P.nlanesB;
  // sort: coarse partition (counts -> scan -> scatter), then one workgroup per coarse bucket
  msm_prof_mark(ctx, st, 1);
  k_msm_coarse_count<<<nsortblk, SORT_THREADS, 0, st>>>((uint32_t)n, m.cap, g, m.scw, m.flag, m.ncoarse, m.nblk_pad, m.matrix, m.sort_terms);
  HIP_TRY(ctx, hipGetLastError());
  const size_t mat_words = (size_t)m.ncoarse * m.nblk_pad;
  const unsigned scan_blocks = (unsigned)(mat_words / 1024);
  if (scan_blocks > 8192) return fail(ctx, S2K_ERR_ARG, "batch too large for the multiscalar sort");
  // (one launch with a chained look-back instead of these three - every workgroup publishing its sum and adding up the sums
  // before it - was measured at + 0.4 ms: 1152 workgroups polling each other's words with agent-scope acquires; profiles/r06_msm_attempts.txt)
  k_msm_scan_blocks<<<scan_blocks, 256, 0, st>>>(m.matrix, m.bsum);
  k_msm_scan_top<<<1, 1024, 0, st>>>(m.bsum, scan_blocks, m.matrix + mat_words);
  k_msm_scan_apply<<<scan_blocks, 256, 0, st>>>(m.matrix, m.bsum, m.matrix);
  HIP_TRY(ctx, hipGetLastError());
  if (m.wide) {
    k_msm_coarse_scatter<true><<<nsortblk, SORT_THREADS, 0, st>>>((uint32_t)n, m.cap, g, m.scw, m.flag, m.ncoarse, m.nblk_pad, m.matrix, (uint2*)m.pairs, m.sort_terms);
    k_msm_fine_sort<true><<<m.ncoarse, FS_THREADS, 0, st>>>(m.ncoarse, m.nblk_pad, m.matrix, (uint32_t)mat_words, (const uint2*)m.pairs, m.offset, m.list, P, m.lanekey);
  } else {
    if (m.sort_terms == SORT_TERMS_STAGED)
      k_msm_coarse_scatter_staged<<<nsortblk, SORT_THREADS, 0, st>>>((uint32_t)n, m.cap, g, m.scw, m.flag, m.ncoarse, m.nblk_pad, m.matrix, (uint32_t*)m.pairs, m.sort_terms);
    else
      k_msm_coarse_scatter<false><<<nsortblk, SORT_THREADS, 0, st>>>((uint32_t)n, m.cap, g, m.scw, m.flag, m.ncoarse, m.nblk_pad, m.matrix, (uint32_t*)m.pairs, m.sort_terms);
    k_msm_fine_sort<false><<<m.ncoarse, FS_THREADS, 0, st>>>(m.ncoarse, m.nblk_pad, m.matrix, (uint32_t)mat_words, (const uint32_t*)m.pairs, m.offset, m.list, P, m.lanekey);
  }
  HIP_TRY(ctx, hipGetLastError());
  msm_prof_mark(ctx, st, 2);
  uint32_t* big2 = m.big + (STITCH_BIG_CAP + 1);
  // stitch and reduce the slots [slot_lo, slot_hi) (the windows [w_lo, w_hi)) on stream s_
  auto tail = [&](hipStream_t s_, uint32_t slot_lo, uint32_t slot_hi, uint32_t w_lo, uint32_t w_hi, uint32_t* big) -> int {
    const uint32_t key_lo = slot_lo * g.nb, key_hi = slot_hi == g.nslot ? (uint32_t)m.nkeys : slot_hi * g.nb;
    if (wsplit && s_ != st)      // (beside the lower part's bucket pass: the form that fits into the registers it leaves)
      k_msm_stitch<4><<<blocks_for(key_hi - key_lo), 256, 0, s_>>>(P, (uint32_t)m.nkeys, m.sum_stride, m.offset, m.xsum, m.list, m.ptw, m.sums, big, key_lo, key_hi);
    else
      k_msm_stitch<3><<<blocks_for(key_hi - key_lo), 256, 0, s_>>>(P, (uint32_t)m.nkeys, m.sum_stride, m.offset, m.xsum, m.list, m.ptw, m.sums, big, key_lo, key_hi);
    k_msm_stitch_big<<<64, 256, 0, s_>>>(P, (uint32_t)m.nkeys, m.sum_stride, m.offset, m.xsum, m.list, m.ptw, m.sums, big);
    HIP_TRY(ctx, hipGetLastError());
    if (fold16) {   // 16-bit windows: row / column sums, bit planes (above)
      const uint32_t ns = slot_hi - slot_lo, nwp = w_hi - w_lo;
      k_msm_fold<<<128 * ns, 256, 0, s_>>>(m.sums, m.sum_stride, m.partial, m.nslots + 1, slot_lo, ns);
      k_msm_planes<<<(nwp + ns) * FOLD_BITS, 1024, 0, s_>>>(m.partial, m.nslots + 1, w_lo, nwp, slot_lo);
      k_msm_plane_horner<<<nwp + ns, 64, 0, s_>>>(m.partial, m.nslots + 1, w_lo, nwp, slot_lo);
      HIP_TRY(ctx, hipGetLastError());
      return S2K_OK;
    }
    k_msm_reduce<<<blocks_for(4 * (size_t)(slot_hi - slot_lo) * g.nchunk), 256, 0, s_>>>(g, m.sums, m.sum_stride, m.partial, slot_lo, slot_hi);
    HIP_TRY(ctx, hipGetLastError());
    // level 1: groups of up to 512 chunk results; level 2: the group sums of each slot
    const uint32_t span1 = g.nchunk < 512 ? g.nchunk : 512, groups = g.nchunk / span1;
    k_msm_tree<<<(slot_hi - slot_lo) * groups, 256, 0, s_>>>((uint32_t)m.nslots + 1, span1, 1u, m.partial, slot_lo * groups);
    if (groups > 1) k_msm_tree<<<slot_hi - slot_lo, 256, 0, s_>>>((uint32_t)m.nslots + 1, groups, span1, m.partial, slot_lo);
    HIP_TRY(ctx, hipGetLastError());
    return S2K_OK;
  };
  if (wsplit == 0) {
    k_msm_accumulate<0><<<blocks_for(nlanes_a), 256, 0, st>>>(P, 0u, (uint32_t)m.nkeys, m.offset, m.lanekey, m.list, m.ptw, m.xsum);
    HIP_TRY(ctx, hipGetLastError());
    msm_prof_mark(ctx, st, 3);
    int rc = tail(st, 0, g.nslot, 0, g.nw, m.big);
    if (rc) return rc;
    msm_prof_mark(ctx, st, 4);
    if (fold16)
      k_msm_final16<<<1, 512, 0, st>>>(m.partial, m.nslots + 1, d_out65, affine ? 1 : 0, 0u, 8u, 0, 0);
    else
      k_msm_final<<<1, 64, 0, st>>>(g, m.partial, d_out65, affine ? 1 : 0, g.nw, 0u);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    int rc = ctx_aux_streams(ctx);                    // (the fork / join events)
    if (rc) return rc;
    hipStream_t s2 = nullptr;                         // the upper part's tail: a stream of its own, served first when a slot frees up
    rc = msm_second_stream(ctx, &s2);
    if (rc) return rc;
    k_msm_accumulate<0><<<blocks_for(nlanes_a), 256, 0, st>>>(P, 0u, (uint32_t)m.nkeys, m.offset, m.lanekey, m.list, m.ptw, m.xsum);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(s2, ctx->ev_fork, 0));
    // the tail of the upper part first (so that it is in its queue when the first part ends), then the lower part's bucket pass
    rc = tail(s2, wsplit, g.nslot, wsplit, g.nw, big2);
    if (rc == S2K_OK) {
      k_msm_final16<<<1, 512, 0, s2>>>(m.partial, m.nslots + 1, d_out65, 0, wsplit, 8u, 0, 1);     // upper windows -> carry slot
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, S2K_ERR_HIP, "k_msm_final16 launch failed");
    }
    if (rc == S2K_OK) {
      // (a block of LDS per workgroup keeps the others out of the CU: 48 KiB - three fit 160 KiB -, 72 KiB - two)
      const unsigned nb_ = blocks_for(P.nlanesB);
      if (b_wgs_per_cu >= 4)
        k_msm_accumulate<0><<<nb_, 256, 0, st>>>(P, 1u, (uint32_t)m.nkeys, m.offset, m.lanekey, m.list, m.ptw, m.xsum);
      else if (b_wgs_per_cu == 3)
        k_msm_accumulate<48><<<nb_, 256, 0, st>>>(P, 1u, (uint32_t)m.nkeys, m.offset, m.lanekey, m.list, m.ptw, m.xsum);
      else
        k_msm_accumulate<72><<<nb_, 256, 0, st>>>(P, 1u, (uint32_t)m.nkeys, m.offset, m.lanekey, m.list, m.ptw, m.xsum);
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, S2K_ERR_HIP, "k_msm_accumulate launch failed");
    }
    msm_prof_mark(ctx, st, 3);
    if (rc == S2K_OK) rc = tail(st, 0, wsplit, 0, wsplit, m.big);
    // error or not: nothing stays in flight on the second stream alone
    (void)hipEventRecord(ctx->ev_join, s2);
    (void)hipStreamWaitEvent(st, ctx->ev_join, 0);
    if (rc) return rc;
    msm_prof_mark(ctx, st, 4);
    k_msm_final16<<<1, 512, 0, st>>>(m.partial, m.nslots + 1, d_out65, affine ? 1 : 0, 0u, wsplit, 1, 0);
    HIP_TRY(ctx, hipGetLastError());
  }
  msm_prof_mark(ctx, st, 5);
  return S2K_OK;
}

// ---------------------------------------------------------------------------------------
// BIP-340 batch verification as ONE multi-scalar multiplication (BIP-340 "Batch Verification"):
//   (sum a_i s_i) * G  -  sum a_i * R_i  -  sum (a_i e_i) * P_i  ==  infinity,
// R_i = lift_x(r_i), P_i = lift_x(pk_i), a_0 = 1 and a_i 128-bit values from a keyed PRF
// (SHA-256(seed || i)); the seed is caller-supplied secret randomness.  The reference has
// only the single-signature Verify (schnorr.go:221-253); the contract is
// "batch accepts <=> every single Verify accepts", up to 2^-128.
// Terms, plain form: [0, n) = a_i * (-R_i); [n, 2n) and [2n, 3n) = the two halves of (a_i e_i) * (-P_i);
// 3n, 3n + 1 = the two halves of (sum a_i s_i) * G.
// Aggregated form (the whole-batch verdict; keys repeat in real batches): the signatures are grouped by key
// (keyed.hip) and each distinct key P contributes ONE term pair (sum over its signatures of a_i e_i) * (-P):
// [0, n) as before, n + 2t, n + 2t + 1 for group t, then the generator's pair.  One square root per distinct
// key instead of one per signature, n + 2 K + 2 terms instead of 3 n + 2.  The sum is the same group
// element, so the error point of a rejected batch can seed the bisection, which works on the plain form.
// ---------------------------------------------------------------------------------------
S2K_DEV bool lift_x_words(uint32_t yw[8], const uint32_t xw[8]) {
  if (!fe_is_canonical_raw(xw)) return false;
  fe29 x = fe29_from_words(xw);
  fe29 rhs = fe29_mul(fe29_sqr(x), x);
  rhs.n[0] += 7;
  fe29 y;
  if (!fe29_sqrt(y, rhs)) return false;
  y = fe29_normalize(y);
  y = fe29_normalize(fe29_select((y.n[0] & 1u) != 0, y, fe29_negate(y, 1)));   // even root
  fe29_to_words(yw, y);
  return true;
}

struct rlc_key {   // PRF key of the coefficients, passed by value as a kernel argument
  uint32_t w[8];
};
// AGG: the key terms are left to k_rlc_key_terms (one term pair per DISTINCT key, its coefficient the sum of
// a_i e_i over the key's signatures): this kernel stores a_i e_i in ae_out instead, and neither lifts nor
// judges the key.  N: plane stride of the term arrays (3n + 2 plain, n + 2 (groups) + 2 aggregated).
template <bool AGG>
__global__ void __launch_bounds__(256)
k_schnorr_rlc_prep(uint32_t n, size_t N, const uint8_t* __restrict__ pk, const uint8_t* __restrict__ sig,
                   const uint8_t* __restrict__ msgs, const uint64_t* __restrict__ offs, uint32_t msg_len,
                   rlc_key seed_be, uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw,
                   uint8_t* __restrict__ flag, uint32_t* __restrict__ as_out, uint32_t* __restrict__ ae_out,
                   uint32_t* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  uint32_t r_le[8], pk_le[8];
  sc s;
  load_be32(r_le, sig + i * 64);
  load_be32(s.v, sig + i * 64 + 32);
  load_be32(pk_le, pk + i * 32);
  bool ok = sc_is_canonical_raw(s.v);
  uint32_t ry[8], py[8];
  ok = lift_x_words(ry, r_le) && ok;      // r < p and on the curve (BIP-340 batch: fail if lift fails)
  if constexpr (!AGG) ok = lift_x_words(py, pk_le) && ok;
  if (!ok) atomicOr(status, 2u);
  // e_i
  uint32_t r_be[8], pk_be[8], dg[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    r_be[w] = r_le[7 - w];
    pk_be[w] = pk_le[7 - w];
  }
  const uint8_t* m = offs ? msgs + offs[i] : msgs + i * (size_t)msg_len;
  uint32_t len = offs ? (uint32_t)(offs[i + 1] - offs[i]) : msg_len;
  bip340_challenge(dg, r_be, pk_be, m, len);
  uint32_t e_raw[8];
#pragma unroll
  for (int w = 0; w < 8; ++w) e_raw[w] = dg[7 - w];
  sc e = sc_reduce_once(e_raw);
  // a_i = low 128 bits of SHA-256(seed || i), a_0 = 1
  sc a = sc_zero();
  if (i == 0) {
    a.v[0] = 1;
  } else {
    uint32_t st[8], w[16];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      st[j] = SHA256_IV[j];
      w[j] = seed_be.w[j];
    }
    w[8] = (uint32_t)(i >> 32);
    w[9] = (uint32_t)i;
    w[10] = 0x80000000u;
    w[11] = w[12] = w[13] = w[14] = 0;
    w[15] = 40 * 8;
    sha256_compress(st, w);
    a.v[0] = st[7]; a.v[1] = st[6]; a.v[2] = st[5]; a.v[3] = st[4];
  }
  sc a_m = sc_to_mont(a);
  sc ae = sc_montmul(e, a_m), as = sc_montmul(s, a_m);
  // -a_i * R_i and -(a_i e_i) * P_i are entered as a_i * (-R_i) and (a_i e_i) * (-P_i): negating the
  // point is free, whereas n - a_i (a_i < 2^128) would put every R term into the same 0xFFFF
  // buckets of the upper windows and serialise them on single lanes.
  if (ok) msm_store_term(scw, ptw, N, i, a, r_le, ry, true);                          // a_i < 2^128 already
  if constexpr (AGG) {
    // (a 32-byte record per signature, not planes: k_rlc_key_terms gathers them by key, from all over the array - eight
    // 4-byte reads per member, each a memory transaction of its own, were most of that kernel's 0.070 ms; now 0.035)
    uint4* rec = reinterpret_cast<uint4*>(ae_out + (size_t)i * 8);
    rec[0] = make_uint4(ae.v[0], ae.v[1], ae.v[2], ae.v[3]);
    rec[1] = make_uint4(ae.v[4], ae.v[5], ae.v[6], ae.v[7]);
  } else {
    uint32_t npy[8];
    u256_sub(npy, FE_P, py);
    if (ok) msm_store_split(scw, ptw, N, (size_t)n + i, 2 * (size_t)n + i, ae, pk_le, npy);
    flag[(size_t)n + i] = ok ? 1 : 0;
    flag[2 * (size_t)n + i] = ok ? 1 : 0;
  }
  // a signature whose r or key does not lift is invalid on its own; it takes no part in the combination
#pragma unroll
  for (int w = 0; w < 8; ++w) as_out[(size_t)w * n + i] = ok ? as.v[w] : 0u;
  flag[i] = ok ? 1 : 0;
}

// Aggregated key terms, in two kernels.  Lane t < ngroups: virtual group t (the signatures perm[first .. first + count)
// of one key); lane ngroups + q: the signature left[q] on its own.
// k_rlc_key_lift lifts the group's key ONCE (a square root: the expensive part; it needs the grouping only, so it runs on
// the second stream beside k_schnorr_rlc_prep): ky[w][t] = the even y, kok[t] = 1 if the key lifts.
__global__ void __launch_bounds__(64)
k_rlc_key_lift(uint32_t ngroups, uint32_t nleft, size_t kstride, const uint32_t* __restrict__ vslot, const uint32_t* __restrict__ rep,
               const uint32_t* __restrict__ left, const uint8_t* __restrict__ pk, uint32_t* __restrict__ ky, uint8_t* __restrict__ kok) {
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= ngroups + nleft) return;
  const uint32_t key_sig = t < ngroups ? rep[vslot[t]] : left[t - ngroups];
  uint32_t pk_le[8], py[8];
  load_be32(pk_le, pk + (size_t)key_sig * 32);
  const bool key_ok = lift_x_words(py, pk_le);
#pragma unroll
  for (int w = 0; w < 8; ++w) ky[(size_t)w * kstride + t] = key_ok ? py[w] : 0u;
  kok[t] = key_ok ? 1 : 0;
}
// k_rlc_key_terms: the group's coefficient is the sum of a_i e_i over its members that are still in the combination
// (flag).  A key that does not lift takes its signatures out of the combination (flag, a_i s_i) and fails the batch
// (status).  Terms n + 2t, n + 2t + 1 (N: plane stride of the term arrays).
__global__ void __launch_bounds__(64)
k_rlc_key_terms(uint32_t n, size_t N, uint32_t ngroups, uint32_t nleft, const uint32_t* __restrict__ vslot,
                const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cnt, const uint32_t* __restrict__ base,
                const uint32_t* __restrict__ tix, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ left,
                const uint8_t* __restrict__ pk, size_t kstride, const uint32_t* __restrict__ ky, const uint8_t* __restrict__ kok,
                const uint32_t* __restrict__ ae, uint32_t* __restrict__ as_io,
                uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, uint8_t* __restrict__ flag,
                uint32_t* __restrict__ status) {
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= ngroups + nleft) return;
  uint32_t first = 0, count = 1, key_sig;
  const uint32_t* members;
  if (t < ngroups) {
    const uint32_t s = vslot[t], v = t - tix[s];
    first = base[s] + v * KG_VGROUP;
    const uint32_t rest = cnt[s] - v * KG_VGROUP;
    count = rest < KG_VGROUP ? rest : KG_VGROUP;
    key_sig = rep[s];
    members = perm;
  } else {
    first = t - ngroups;
    key_sig = left[first];
    members = left;
  }
  const bool key_ok = kok[t] != 0;
  sc sum = sc_zero();
  bool any = false;
#pragma unroll 1
  for (uint32_t j = 0; j < count; ++j) {
    const size_t i = members[first + j];
    if (!flag[i]) continue;
    if (!key_ok) {   // out of the combination
      flag[i] = 0;
#pragma unroll
      for (int w = 0; w < 8; ++w) as_io[(size_t)w * n + i] = 0u;
      continue;
    }
    const uint4* rec = reinterpret_cast<const uint4*>(ae + i * 8);
    const uint4 lo4 = rec[0], hi4 = rec[1];
    sc v;
    v.v[0] = lo4.x; v.v[1] = lo4.y; v.v[2] = lo4.z; v.v[3] = lo4.w;
    v.v[4] = hi4.x; v.v[5] = hi4.y; v.v[6] = hi4.z; v.v[7] = hi4.w;
    sum = sc_add(sum, v);
    any = true;
  }
  if (!key_ok) atomicOr(status, 2u);
  const size_t t1 = (size_t)n + 2 * (size_t)t, t2 = t1 + 1;
  if (key_ok && any) {
    uint32_t pk_le[8], py[8], npy[8];
    load_be32(pk_le, pk + (size_t)key_sig * 32);
#pragma unroll
    for (int w = 0; w < 8; ++w) py[w] = ky[(size_t)w * kstride + t];
    u256_sub(npy, FE_P, py);
    msm_store_split(scw, ptw, N, t1, t2, sum, pk_le, npy);
    flag[t1] = 1;
    flag[t2] = 1;
  }
}

// sum of n scalars mod n in two launches: RLC_SUM_BLOCKS workgroups leave one partial sum each in
// `part` (8 words each), then one workgroup folds those and writes the result as terms 3n, 3n + 1
// with the point G (as == nullptr selects the second stage); N terms in arrays of plane stride tstride
constexpr uint32_t RLC_SUM_BLOCKS = 256;
// (`as` has plane stride `stride`; a sub-range of a saved batch passes as + lo with the batch's stride)
__global__ void __launch_bounds__(256)
k_schnorr_rlc_sum(uint32_t n, size_t stride, const uint32_t* __restrict__ as, uint32_t* __restrict__ part,
                  uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw, uint8_t* __restrict__ flag, size_t N, size_t tstride) {
  __shared__ uint32_t sh[256][8];
  sc acc = sc_zero();
  if (as) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)RLC_SUM_BLOCKS * 256) {
      sc v;
#pragma unroll
      for (int w = 0; w < 8; ++w) v.v[w] = as[(size_t)w * stride + i];
      acc = sc_add(acc, v);
    }
  } else if (threadIdx.x < RLC_SUM_BLOCKS) {
#pragma unroll
    for (int w = 0; w < 8; ++w) acc.v[w] = part[threadIdx.x * 8 + w];
  }
#pragma unroll
  for (int w = 0; w < 8; ++w) sh[threadIdx.x][w] = acc.v[w];
  __syncthreads();
  for (uint32_t half = 128; half >= 1; half >>= 1) {
    if (threadIdx.x < half) {
      sc a, b;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        a.v[w] = sh[threadIdx.x][w];
        b.v[w] = sh[threadIdx.x + half][w];
      }
      a = sc_add(a, b);
#pragma unroll
      for (int w = 0; w < 8; ++w) sh[threadIdx.x][w] = a.v[w];
    }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  if (as) {
#pragma unroll
    for (int w = 0; w < 8; ++w) part[blockIdx.x * 8 + w] = sh[0][w];
  } else {
    sc tot;                                   // the generator's terms are the last two of the N
#pragma unroll
    for (int w = 0; w < 8; ++w) tot.v[w] = sh[0][w];
    msm_store_split(scw, ptw, tstride, N - 2, N - 1, tot, FE_GX, FE_GY);
    flag[N - 2] = 1;
    flag[N - 1] = 1;
  }
}

// ---------------------------------------------------------------------------------------
// Locating the failing signatures of a rejected batch.  The terms of the whole
// batch (coefficients, challenges times coefficients, lifted points: the expensive part of the
// preparation) are kept; a sub-range [lo, lo + m) of the signatures is re-checked by gathering its
// 3m terms into a fresh term array, summing its a_i s_i for the generator term and running the
// multiscalar core on 3m + 2 terms.  The result is the range's error point E = sum of the
// coefficients times the individual errors; E_right = E_parent - E_left, so every level costs one
// multiscalar multiplication of half its parent's size.
// Saved layout: s_scw [4][3n] (magnitudes), s_ptw [3n][16], s_flag [n], s_as [8][n].
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_rlc_gather(uint32_t lo, uint32_t m, uint32_t n, const uint32_t* __restrict__ s_scw, const uint32_t* __restrict__ s_ptw,
             const uint8_t* __restrict__ s_flag, uint32_t* __restrict__ scw, uint32_t* __restrict__ ptw,
             uint8_t* __restrict__ flag) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;     // destination term among the 3m
  if (j >= 3 * (size_t)m) return;
  const size_t part = j / m, i = j - part * m;            // 0: R terms, 1 and 2: the halves of the P terms
  const size_t src = part * n + lo + i, N = 3 * (size_t)m + 2, NS = 3 * (size_t)n;
#pragma unroll
  for (int w = 0; w < 4; ++w) scw[(size_t)w * N + j] = s_scw[(size_t)w * NS + src];
  scw[(size_t)4 * N + j] = 0;
  const uint4* a = reinterpret_cast<const uint4*>(s_ptw + src * 16);
  uint4* b = reinterpret_cast<uint4*>(ptw + j * 16);
  b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; b[3] = a[3];
  flag[j] = s_flag[lo + i];
}
// terms of the whole batch -> saved arrays (3n + 2 -> 3n: the generator terms are rebuilt per range)
__global__ void __launch_bounds__(256)
k_rlc_save(uint32_t n, const uint32_t* __restrict__ scw, const uint32_t* __restrict__ ptw, const uint8_t* __restrict__ flag,
           uint32_t* __restrict__ s_scw, uint32_t* __restrict__ s_ptw, uint8_t* __restrict__ s_flag) {
  size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= 3 * (size_t)n) return;
  const size_t N = 3 * (size_t)n + 2, NS = 3 * (size_t)n;
#pragma unroll
  for (int w = 0; w < 4; ++w) s_scw[(size_t)w * NS + j] = scw[(size_t)w * N + j];
  const uint4* a = reinterpret_cast<const uint4*>(ptw + j * 16);
  uint4* b = reinterpret_cast<uint4*>(s_ptw + j * 16);
  b[0] = a[0]; b[1] = a[1]; b[2] = a[2]; b[3] = a[3];
  if (j < n) s_flag[j] = flag[j];
}
// valid[lo + i] = ok-to-lift flag of signature lo + i (a range the combination has cleared)
__global__ void __launch_bounds__(256)
k_rlc_mark_valid(uint32_t lo, uint32_t m, const uint8_t* __restrict__ s_flag, uint8_t* __restrict__ valid) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < m) valid[lo + i] = s_flag[lo + i];
}

}  // namespace

extern "C" {

int s2k_ctx_profile_msm(s2k_ctx* ctx, int enable) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (enable && !ctx->msm_prof_ev) {
    const size_t cap = MSM_PROF_EV * 256;   // 256 calls between two reads
    ctx->msm_prof_ev = new (std::nothrow) hipEvent_t[cap];
    if (!ctx->msm_prof_ev) return fail(ctx, S2K_ERR_NOMEM, "out of host memory");
    for (size_t i = 0; i < cap; ++i) {
      hipError_t e = hipEventCreate(&ctx->msm_prof_ev[i]);
      if (e != hipSuccess) {
        for (size_t j = 0; j < i; ++j) (void)hipEventDestroy(ctx->msm_prof_ev[j]);
        delete[] ctx->msm_prof_ev;
        ctx->msm_prof_ev = nullptr;
        return fail(ctx, S2K_ERR_HIP, "hipEventCreate: %s", hipGetErrorString(e));
      }
    }
    ctx->msm_prof_cap = cap;
  }
  ctx->msm_prof_on = enable != 0;
  ctx->msm_prof_used = 0;
  return S2K_OK;
}
int s2k_ctx_profile_read_msm(s2k_ctx* ctx, double ms_sum5[5], size_t* calls) {
  if (!ctx || !ms_sum5 || !calls) return fail(ctx, S2K_ERR_ARG, "null argument");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  HIP_TRY(ctx, hipDeviceSynchronize());
  for (int j = 0; j < MSM_PROF_EV - 1; ++j) ms_sum5[j] = 0.0;
  const size_t k = ctx->msm_prof_used / MSM_PROF_EV;
  for (size_t i = 0; i < k; ++i)
    for (int j = 0; j < MSM_PROF_EV - 1; ++j) {
      float ms = 0.f;
      HIP_TRY(ctx, hipEventElapsedTime(&ms, ctx->msm_prof_ev[MSM_PROF_EV * i + j], ctx->msm_prof_ev[MSM_PROF_EV * i + j + 1]));
      ms_sum5[j] += ms;
    }
  *calls = k;
  ctx->msm_prof_used = 0;
  return S2K_OK;
}

int s2k_multi_scalar_mult_device(s2k_ctx* ctx, size_t n, const void* d_scalars, const void* d_points, void* d_out65,
                                 void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!d_out65) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n && (!d_scalars || !d_points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  if (n > 0x1fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  if (n == 0) {   // l == 0: identity (point_mul_multi.go:37 v.Identity())
    HIP_TRY(ctx, hipMemsetAsync(d_out65, 0, 65, st));
    return S2K_OK;
  }
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  msm_ws m;
  rc = msm_setup(ctx, 2 * n, 0, m);   // every input is two terms (endomorphism split)
  if (rc) return rc;
  msm_prof_mark(ctx, st, 0);
  HIP_TRY(ctx, hipMemsetAsync(ctx->msm_ws, 0, m.zero_bytes, st));
  // (the front end moves 275 MB in 0.075 ms - it is bound by that, not by its arithmetic: splitting it into a scalar kernel in
  // front of the sort and a point kernel beside it on a second stream was measured 0.03 ms SLOWER, profiles/r06_msm_attempts.txt)
  k_msm_parse<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, (const uint8_t*)d_scalars, (const uint8_t*)d_points, m.scw,
                                             m.ptw, m.flag, m.status);
  HIP_TRY(ctx, hipGetLastError());
  rc = msm_core(ctx, st, 2 * n, m, (uint8_t*)d_out65);
  if (rc) return rc;
  // malformed point records are a caller error (the reference cannot even construct such Points)
  uint32_t h_status = 0;
  HIP_TRY(ctx, hipMemcpyAsync(&h_status, m.status, 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;   // the stream has been synchronised: nothing of this context is in flight
  if (h_status) return fail(ctx, S2K_ERR_ARG, "malformed point record in multi-scalar multiplication input");
  return S2K_OK;
}

int s2k_multi_scalar_mult(s2k_ctx* ctx, size_t n, const uint8_t* scalars, const uint8_t* points, uint8_t* out65) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!out65) return fail(ctx, S2K_ERR_ARG, "null output buffer");
  if (n && (!scalars || !points)) return fail(ctx, S2K_ERR_ARG, "null input buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  const size_t sizes[3] = {n * 32 + 16, n * 65 + 16, 128};
  uint8_t* d[3];
  rc = ctx_stage(ctx, sizes, 3, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  s2k_phase_guard phase(ctx->device, n * 97);            // (two verifiers on two threads: engine_internal.h)
  if (n) {
    HIP_TRY(ctx, hipMemcpyAsync(d[0], scalars, n * 32, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d[1], points, n * 65, hipMemcpyHostToDevice, st));
  }
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_multi_scalar_mult_device(ctx, n, d[0], d[1], d[2], st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(out65, d[2], 65, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  return S2K_OK;
}

// SHA-256 on the host (FIPS 180-4), for mixing the caller's seed with operating-system randomness
static void host_sha256(uint8_t out[32], const uint8_t* msg, size_t len) {
  static const uint32_t K[64] = {
      0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
      0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
      0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
      0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
      0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
      0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
      0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
  uint32_t st[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
  uint8_t buf[128];
  size_t full = len / 64, rem = len % 64;
  auto block = [&](const uint8_t* b) {
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    auto rotr = [](uint32_t x, int n) { return (x >> n) | (x << (32 - n)); };
    for (int i = 16; i < 64; ++i) {
      uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
      w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], bb = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], hh = st[7];
    for (int i = 0; i < 64; ++i) {
      uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25), ch = (e & f) ^ (~e & g), t1 = hh + S1 + ch + K[i] + w[i];
      uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22), mj = (a & bb) ^ (a & c) ^ (bb & c), t2 = S0 + mj;
      hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    st[0] += a; st[1] += bb; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += hh;
  };
  for (size_t i = 0; i < full; ++i) block(msg + 64 * i);
  memset(buf, 0, sizeof buf);
  memcpy(buf, msg + 64 * full, rem);
  buf[rem] = 0x80;
  size_t padded = rem + 9 <= 64 ? 64 : 128;
  uint64_t bits = (uint64_t)len * 8;
  for (int i = 0; i < 8; ++i) buf[padded - 1 - i] = (uint8_t)(bits >> (8 * i));
  block(buf);
  if (padded == 128) block(buf + 64);
  for (int i = 0; i < 8; ++i) {
    out[4 * i] = (uint8_t)(st[i] >> 24); out[4 * i + 1] = (uint8_t)(st[i] >> 16); out[4 * i + 2] = (uint8_t)(st[i] >> 8); out[4 * i + 3] = (uint8_t)st[i];
  }
}

// ct_cpu.cpp: out = a - b on 65-byte records (host; public data)
extern "C" __attribute__((visibility("hidden"))) int s2k_internal_point_sub65(const uint8_t* a, const uint8_t* b, uint8_t* out);

static int stage_schnorr(s2k_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msgs, const uint64_t* msg_offsets, size_t msg_len,
                         const uint8_t* sig, uint8_t* d[5], hipStream_t* st_out);

// whole-batch combination: preparation, generator term, multiscalar core; leaves the 129 bytes
// [status word .. 64: error point record] in h.  The caller's seed is mixed with 32 bytes from the
// operating system's CSPRNG, so a reused or predictable seed does not weaken the check (the
// coefficients have to be unpredictable to whoever chose the signatures).
enum rlc_mode {
  RLC_AGGREGATED,        // one term pair per distinct key; combination evaluated
  RLC_PLAIN_TERMS_ONLY   // one term pair per signature (the layout the bisection gathers from); terms only
};
// `seed_be`: the coefficient key; derived here from seed32 and OS randomness when `fresh`, otherwise taken
// as it is (a second pass over the same batch must use the coefficients of the first).
static int rlc_run_full(s2k_ctx* ctx, hipStream_t st, size_t n, const void* d_pk, const void* d_msgs,
                        const void* d_msg_offsets, size_t msg_len, const void* d_sig, const uint8_t* seed32, msm_ws& m,
                        uint32_t** as_out, uint8_t h[192], rlc_mode mode, rlc_key& seed_be, bool fresh) {
  if (fresh) {
    uint8_t mix[64], key[32];
    memcpy(mix, seed32, 32);
    if (getrandom(mix + 32, 32, 0) != 32) return fail(ctx, S2K_ERR_HIP, "getrandom failed: no randomness for the batch coefficients");
    host_sha256(key, mix, 64);
    for (int j = 0; j < 8; ++j)
      seed_be.w[j] = ((uint32_t)key[4 * j] << 24) | ((uint32_t)key[4 * j + 1] << 16) | ((uint32_t)key[4 * j + 2] << 8) | key[4 * j + 3];
  }
  // The workspace is carved for the plain form's 3n + 2 terms in both modes (the aggregated form has fewer, but how many
  // is known only once the keys are grouped, and the preparation must not wait for that): `cap` is the plane stride.
  const size_t cap = 3 * n + 2;
  size_t N = cap;
  // aux: 256 bytes | partial sums of the generator's coefficient | a_i s_i planes | a_i e_i records, lifted keys (aggregated)
  const size_t as_bytes = n * 8 * 4, kstride = (n + 63) & ~(size_t)63;
  int rc = msm_setup(ctx, cap, 256 + RLC_SUM_BLOCKS * 32 + as_bytes + (mode == RLC_AGGREGATED ? as_bytes + kstride * 36 : 0), m);
  if (rc) return rc;
  uint32_t* sum_part = (uint32_t*)(m.aux + 256);
  uint32_t* as = (uint32_t*)(m.aux + 256 + RLC_SUM_BLOCKS * 32);
  uint32_t* ae = as + n * 8;
  uint32_t* ky = ae + n * 8;
  uint8_t* kok = (uint8_t*)(ky + kstride * 8);
  msm_prof_mark(ctx, st, 0);
  HIP_TRY(ctx, hipMemsetAsync(ctx->msm_ws, 0, m.zero_bytes, st));
  if (mode == RLC_AGGREGATED) {
    // Two streams.  Caller's: the per-signature preparation (challenge hash, square root of r, coefficients: 1.2 ms of
    // multiplier time for 2^20 signatures).  Second: the grouping of the keys (hash table, latency bound; it ends with
    // the one host read of the call, the group count) and one square root per DISTINCT key.  They meet at the key terms.
    rc = ctx_aux_streams(ctx);
    if (rc) return rc;
    rc = s2k_internal_key_reserve32(ctx, n);   // the grouping arrays are grown before the fork (growing synchronises the device)
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(m.flag, 0, cap, st));     // the key terms set their own flags
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, st));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_aux, ctx->ev_fork, 0));
    k_schnorr_rlc_prep<true><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, cap, (const uint8_t*)d_pk, (const uint8_t*)d_sig,
                                                            (const uint8_t*)d_msgs, (const uint64_t*)d_msg_offsets,
                                                            (uint32_t)msg_len, seed_be, m.scw, m.ptw, m.flag, as, ae, m.status);
    key_groups32 kg;
    rc = hipGetLastError() == hipSuccess ? S2K_OK : fail(ctx, S2K_ERR_HIP, "k_schnorr_rlc_prep launch failed");
    if (rc == S2K_OK) rc = s2k_internal_key_group32(ctx, n, (const uint8_t*)d_pk, ctx->s_aux, &kg);   // (synchronises the second stream)
    uint32_t lanes = 0;
    if (rc == S2K_OK) {
      lanes = kg.ngroups + kg.nleft;
      N = n + 2 * (size_t)lanes + 2;
      k_rlc_key_lift<<<(lanes + 63) / 64, 64, 0, ctx->s_aux>>>(kg.ngroups, kg.nleft, kstride, kg.vslot, kg.rep, kg.left, (const uint8_t*)d_pk, ky, kok);
      if (hipGetLastError() != hipSuccess) rc = fail(ctx, S2K_ERR_HIP, "k_rlc_key_lift launch failed");
    }
    ctx_aux_join(ctx, st);            // error or not: nothing stays in flight on the second stream alone
    if (rc) {
      (void)ctx_leave(ctx, st);
      return rc;
    }
    k_rlc_key_terms<<<(lanes + 63) / 64, 64, 0, st>>>((uint32_t)n, cap, kg.ngroups, kg.nleft, kg.vslot, kg.rep, kg.cnt, kg.base, kg.tix,
                                                      kg.perm, kg.left, (const uint8_t*)d_pk, kstride, ky, kok, ae, as, m.scw, m.ptw, m.flag, m.status);
    HIP_TRY(ctx, hipGetLastError());
  } else {
    k_schnorr_rlc_prep<false><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, cap, (const uint8_t*)d_pk, (const uint8_t*)d_sig,
                                                             (const uint8_t*)d_msgs, (const uint64_t*)d_msg_offsets,
                                                             (uint32_t)msg_len, seed_be, m.scw, m.ptw, m.flag, as, nullptr, m.status);
    HIP_TRY(ctx, hipGetLastError());
  }
  if (as_out) *as_out = as;
  if (mode == RLC_PLAIN_TERMS_ONLY) return S2K_OK;
  k_schnorr_rlc_sum<<<RLC_SUM_BLOCKS, 256, 0, st>>>((uint32_t)n, n, as, sum_part, m.scw, m.ptw, m.flag, N, cap);
  k_schnorr_rlc_sum<<<1, 256, 0, st>>>((uint32_t)n, n, nullptr, sum_part, m.scw, m.ptw, m.flag, N, cap);
  HIP_TRY(ctx, hipGetLastError());
  uint8_t* d_out = (uint8_t*)m.status + 64;   // 65-byte record inside the 256-byte status slot
  rc = msm_core(ctx, st, N, m, d_out, /*affine=*/as_out != nullptr);   // the verdict alone needs no coordinates; the bisection does
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(h, m.status, 64 + 65, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

int s2k_schnorr_batch_verify_rlc_device(s2k_ctx* ctx, size_t n, const void* d_pk, const void* d_msgs,
                                        const void* d_msg_offsets, size_t msg_len, const void* d_sig,
                                        const uint8_t* seed32, int* all_valid, void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!all_valid || !seed32) return fail(ctx, S2K_ERR_ARG, "null argument");
  *all_valid = 0;
  if (n == 0) {
    *all_valid = 1;   // an empty batch has no failing signature
    return S2K_OK;
  }
  if (!d_pk || !d_sig || (!d_msgs && (d_msg_offsets || msg_len))) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n > 0x0fffffffu || msg_len > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  msm_ws m;
  uint8_t h[192];
  rlc_key coeff;
  rc = rlc_run_full(ctx, st, n, d_pk, d_msgs, d_msg_offsets, msg_len, d_sig, seed32, m, nullptr, h, RLC_AGGREGATED, coeff, true);
  if (rc) return rc;
  uint32_t h_status;
  memcpy(&h_status, h, 4);
  *all_valid = (h_status == 0 && h[64] == 0x00) ? 1 : 0;
  return S2K_OK;
}

// Per-signature verdicts at the price of the whole-batch check when (as usual) everything verifies:
// one combination over the batch; if it is rejected, the failing signatures are located by
// bisection on the kept terms (header).  stats (host, optional): [0] sub-range multiscalar
// multiplications, [1] signatures verified one by one, [2] levels descended, [3] 1 = gave up
// bisecting (many failing ranges) and verified the rest one by one.
int s2k_schnorr_verify_batch_bisect_device(s2k_ctx* ctx, size_t n, const void* d_pk, const void* d_msgs,
                                           const void* d_msg_offsets, size_t msg_len, const void* d_sig,
                                           const uint8_t* seed32, void* d_valid, uint32_t* stats, void* hip_stream) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!seed32) return fail(ctx, S2K_ERR_ARG, "null argument");
  if (stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
  if (n == 0) return S2K_OK;
  if (!d_pk || !d_sig || !d_valid || (!d_msgs && (d_msg_offsets || msg_len))) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (n > 0x0fffffffu || msg_len > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipStream_t st = (hipStream_t)hip_stream;
  int rc = ctx_enter(ctx, st);
  if (rc) return rc;
  const uint8_t *pk = (const uint8_t*)d_pk, *sig = (const uint8_t*)d_sig, *msgs = (const uint8_t*)d_msgs;
  const uint64_t* offs = (const uint64_t*)d_msg_offsets;
  uint8_t* valid = (uint8_t*)d_valid;
  // Below LEAF signatures a range is verified signature by signature: a combination costs at least the
  // multiscalar floor (0.75 ms + preparation), per-signature verification 0.56 ms per 2^16 signatures.
  constexpr uint32_t LEAF = 1u << 17;
  constexpr size_t MAX_FAILING = 8;        // more failing ranges than this on one level: stop bisecting
  auto verify_each = [&](uint32_t lo, uint32_t cnt) -> int {
    if (stats) stats[1] += cnt;
    return s2k_schnorr_verify_batch_device(ctx, cnt, pk + (size_t)lo * 32, offs ? msgs : (msgs ? msgs + (size_t)lo * msg_len : nullptr),
                                           offs ? offs + lo : nullptr, msg_len, sig + (size_t)lo * 64, 0, valid + lo, st);
  };
  if (n <= LEAF) return verify_each(0, (uint32_t)n);

  msm_ws m;
  uint32_t* as = nullptr;
  uint8_t h[192];
  rlc_key coeff;
  rc = rlc_run_full(ctx, st, n, d_pk, d_msgs, d_msg_offsets, msg_len, d_sig, seed32, m, &as, h, RLC_AGGREGATED, coeff, true);
  if (rc) return rc;
  if (h[64] == 0x00) {   // the combination of every liftable signature vanishes: those are all valid
    k_rlc_mark_valid<<<blocks_for(n), 256, 0, st>>>(0u, (uint32_t)n, m.flag, valid);
    HIP_TRY(ctx, hipGetLastError());
    return ctx_leave(ctx, st);
  }
  // Rejected.  The bisection gathers per-signature terms, so the plain form is prepared now, with the
  // coefficients of the pass that has just failed: its error point h + 64 is the plain form's as well.
  {
    uint8_t h2[192];
    rc = rlc_run_full(ctx, st, n, d_pk, d_msgs, d_msg_offsets, msg_len, d_sig, seed32, m, &as, h2, RLC_PLAIN_TERMS_ONLY, coeff, false);
    if (rc) return rc;
  }
  // keep the terms: the sub-range runs re-carve the multiscalar workspace
  const size_t NS = 3 * n;
  auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t o_scw = 0, o_ptw = pad(NS * 4 * 4), o_flag = o_ptw + pad(NS * 16 * 4), o_as = o_flag + pad(n),
               total = o_as + pad(n * 8 * 4);
  rc = ctx_reserve(ctx, &ctx->rlc_save, &ctx->rlc_save_bytes, total);
  if (rc) return rc;
  uint8_t* sv = (uint8_t*)ctx->rlc_save;
  uint32_t *s_scw = (uint32_t*)(sv + o_scw), *s_ptw = (uint32_t*)(sv + o_ptw), *s_as = (uint32_t*)(sv + o_as);
  uint8_t* s_flag = sv + o_flag;
  k_rlc_save<<<blocks_for(NS), 256, 0, st>>>((uint32_t)n, m.scw, m.ptw, m.flag, s_scw, s_ptw, s_flag);
  HIP_TRY(ctx, hipGetLastError());
  HIP_TRY(ctx, hipMemcpyAsync(s_as, as, n * 8 * 4, hipMemcpyDeviceToDevice, st));

  // error point of a sub-range
  auto sub = [&](uint32_t lo, uint32_t cnt, uint8_t E[65]) -> int {
    msm_ws w;
    const size_t N = 3 * (size_t)cnt + 2;
    int r = msm_setup(ctx, N, 256 + RLC_SUM_BLOCKS * 32, w);
    if (r) return r;
    uint32_t* sum_part = (uint32_t*)w.aux;
    HIP_TRY(ctx, hipMemsetAsync(ctx->msm_ws, 0, w.zero_bytes, st));
    k_rlc_gather<<<blocks_for(3 * (size_t)cnt), 256, 0, st>>>(lo, cnt, (uint32_t)n, s_scw, s_ptw, s_flag, w.scw, w.ptw, w.flag);
    k_schnorr_rlc_sum<<<RLC_SUM_BLOCKS, 256, 0, st>>>(cnt, n, s_as + lo, sum_part, w.scw, w.ptw, w.flag, N, N);
    k_schnorr_rlc_sum<<<1, 256, 0, st>>>(cnt, n, nullptr, sum_part, w.scw, w.ptw, w.flag, N, N);
    HIP_TRY(ctx, hipGetLastError());
    uint8_t* d_out = (uint8_t*)w.status + 64;
    r = msm_core(ctx, st, N, w, d_out);
    if (r) return r;
    HIP_TRY(ctx, hipMemcpyAsync(E, d_out, 65, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (stats) stats[0] += 1;
    return S2K_OK;
  };
  struct range {
    uint32_t lo, cnt;
    uint8_t E[65];
  };
  std::vector<range> cur(1), next;
  cur[0].lo = 0;
  cur[0].cnt = (uint32_t)n;
  memcpy(cur[0].E, h + 64, 65);
  while (!cur.empty()) {
    if (cur.size() > MAX_FAILING) {   // errors everywhere: bisection would cost more than it saves
      if (stats) stats[3] = 1;
      for (const range& r : cur) {
        rc = verify_each(r.lo, r.cnt);
        if (rc) return rc;
      }
      break;
    }
    next.clear();
    for (const range& r : cur) {
      if (r.cnt <= LEAF) {
        rc = verify_each(r.lo, r.cnt);
        if (rc) return rc;
        continue;
      }
      range left, right;
      left.lo = r.lo;
      left.cnt = r.cnt / 2;
      right.lo = r.lo + left.cnt;
      right.cnt = r.cnt - left.cnt;
      rc = sub(left.lo, left.cnt, left.E);
      if (rc) return rc;
      if (s2k_internal_point_sub65(r.E, left.E, right.E) != 0) return fail(ctx, S2K_ERR_HIP, "internal: bad error point");
      for (const range* h2 : {&left, &right}) {
        if (h2->E[0] == 0x00) {
          k_rlc_mark_valid<<<blocks_for(h2->cnt), 256, 0, st>>>(h2->lo, h2->cnt, s_flag, valid);
          HIP_TRY(ctx, hipGetLastError());
        } else {
          next.push_back(*h2);
        }
      }
    }
    if (stats && !next.empty()) stats[2] += 1;
    cur.swap(next);
  }
  return ctx_leave(ctx, st);
}

int s2k_schnorr_verify_batch_bisect(s2k_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                    size_t msg_len, const uint8_t* sig, const uint8_t* seed32, uint8_t* valid, uint32_t* stats) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (n == 0) return S2K_OK;
  if (!pk || !sig || !valid || !seed32) return fail(ctx, S2K_ERR_ARG, "null buffer");
  uint8_t* d[5];
  hipStream_t st;
  s2k_phase_guard phase(ctx->device, n * 128);           // (two verifiers on two threads: engine_internal.h)
  int rc = stage_schnorr(ctx, n, pk, msgs, msg_offsets, msg_len, sig, d, &st);
  if (rc) return rc;
  HIP_TRY(ctx, phase.landed(st));
  rc = s2k_schnorr_verify_batch_bisect_device(ctx, n, d[0], d[1], msg_offsets ? d[2] : nullptr, msg_len, d[3], seed32, d[4], stats, st);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(valid, d[4], n, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  ctx->have_last = false;
  return S2K_OK;
}

// stage (pk, msgs, offsets, sig [, valid]) of a BIP-340 batch in the context's buffers on its compute stream
static int stage_schnorr(s2k_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msgs, const uint64_t* msg_offsets, size_t msg_len,
                         const uint8_t* sig, uint8_t* d[5], hipStream_t* st_out) {
  size_t total = msg_offsets ? (size_t)msg_offsets[n] : n * msg_len;
  if (total && !msgs) return fail(ctx, S2K_ERR_ARG, "null message buffer");
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int rc = ctx_streams(ctx);
  if (rc) return rc;
  const size_t sizes[5] = {n * 32, total ? total : 16, (n + 1) * sizeof(uint64_t), n * 64, n};
  rc = ctx_stage(ctx, sizes, 5, d);
  if (rc) return rc;
  hipStream_t st = ctx->s_comp;
  HIP_TRY(ctx, hipMemcpyAsync(d[0], pk, n * 32, hipMemcpyHostToDevice, st));
  if (total) HIP_TRY(ctx, hipMemcpyAsync(d[1], msgs, total, hipMemcpyHostToDevice, st));
  if (msg_offsets) HIP_TRY(ctx, hipMemcpyAsync(d[2], msg_offsets, (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st));
  HIP_TRY(ctx, hipMemcpyAsync(d[3], sig, n * 64, hipMemcpyHostToDevice, st));
  *st_out = st;
  return S2K_OK;
}

int s2k_schnorr_batch_verify_rlc(s2k_ctx* ctx, size_t n, const uint8_t* pk, const uint8_t* msgs,
                                 const uint64_t* msg_offsets, size_t msg_len, const uint8_t* sig,
                                 const uint8_t* seed32, int* all_valid) {
  if (!ctx) return fail(nullptr, S2K_ERR_ARG, "ctx is NULL");
  if (!all_valid || !seed32) return fail(ctx, S2K_ERR_ARG, "null argument");
  if (n == 0) {
    *all_valid = 1;
    return S2K_OK;
  }
  if (!pk || !sig) return fail(ctx, S2K_ERR_ARG, "null buffer");
  uint8_t* d[5];
  hipStream_t st;
  s2k_phase_guard phase(ctx->device, n * 128);           // (two verifiers on two threads: engine_internal.h)
  int rc = stage_schnorr(ctx, n, pk, msgs, msg_offsets, msg_len, sig, d, &st);
  if (rc) return rc;
  HIP_TRY(ctx, phase.landed(st));
  return s2k_schnorr_batch_verify_rlc_device(ctx, n, d[0], d[1], msg_offsets ? d[2] : nullptr, msg_len, d[3], seed32, all_valid, st);
}

}  // extern "C"

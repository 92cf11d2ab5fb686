// lane_tables.h — storage of the fast path's per-lane scratch elements and point tables (shared by
// engine.hip and keyed.hip).
#pragma once
#include "fe29.h"

namespace s2k {

// Small per-lane scratch elements ("fin" region, 4 elements): 16-byte planes [quad][lane], three quads
// per field element (limbs 0-3 | 4-7 | 8,-,-,-).  Element 0-2: (X, Y, Z) handed to k_affine_finish,
// element 3: Z_7 * C during the ladder, then the prefix products of the shared inversion.
constexpr int FQT_FE_WORDS = 12;
constexpr int FIN_ELEMS = 4, FIN_WORDS = FIN_ELEMS * FQT_FE_WORDS;
constexpr int TBL_WORDS = 8 * 8 * 4;          // per-lane table: 8 entries x 8 quads (7 used)
S2K_DEV void fq_store(uint32_t* __restrict__ base, size_t stride, size_t lane, int elem, const fe29& v) {
  uint4* q = reinterpret_cast<uint4*>(base) + (size_t)(elem * 3) * stride + lane;
  q[0] = make_uint4(v.n[0], v.n[1], v.n[2], v.n[3]);
  q[stride] = make_uint4(v.n[4], v.n[5], v.n[6], v.n[7]);
  q[2 * stride] = make_uint4(v.n[8], 0u, 0u, 0u);
}
S2K_DEV fe29 fq_load(const uint32_t* __restrict__ base, size_t stride, size_t lane, uint32_t elem) {
  fe29 r;
  const uint4* q = reinterpret_cast<const uint4*>(base) + (size_t)(elem * 3) * stride + lane;
  uint4 a = q[0], b = q[stride];
  r.n[0] = a.x; r.n[1] = a.y; r.n[2] = a.z; r.n[3] = a.w; r.n[4] = b.x; r.n[5] = b.y; r.n[6] = b.z; r.n[7] = b.w;
  r.n[8] = reinterpret_cast<const uint32_t*>(q + 2 * stride)[0];
  return r;
}
// Per-signature table.  Entry j (of 8 odd multiples) is seven quads of 16 bytes,
//   [x limbs 0-3][x 4-7][y 0-3][y 4-7][beta*x 0-3][beta*x 4-7][x8, y8, (beta*x)8, -]
// so that a ladder lookup (x or beta*x, and y) is five 16-byte loads.  During the table build the
// beta*x slot of entry j holds H_j.  Where the quads live (S2K_QT_PACK):
//   2 (default)  the quads of one (entry, lane) pair are CONTIGUOUS: 128 bytes = one cache line per
//                lookup, [entry][lane][8 quads]
//   1            planes [entry * 7 + quad][lane] (a wave's access to one quad is contiguous, but every
//                lane of a lookup lands in a different line of five different planes)
// Measured on MI355X, 2^20 signatures, same box (tools/ab_libs.sh; profiles/r02_table_layouts.md): the
// round-1 layout (three 16-byte planes per element, separate beta*x column: six loads per lookup,
// 1200 bytes per signature) 7.93-8.04 ms and 16.2 GB of L2-miss reads per launch (FETCH_SIZE, raw);
// planes of packed entries 7.76-7.86 ms, 12.4 GB; contiguous entries 7.61-7.66 ms, 4.7 GB.
#ifndef S2K_QT_PACK
#define S2K_QT_PACK 2
#endif
enum { TB_X = 0, TB_Y = 1, TB_BX = 2 };
constexpr int TB_ZC_ELEM = 3;   // in the fin region
#if S2K_QT_PACK == 2
#define TB_ENTRY(base4, stride, lane, entry) ((base4) + ((size_t)(entry) * (stride) + (lane)) * 8)
#define TB_Q(stride, q) ((size_t)(q))
#else
#define TB_ENTRY(base4, stride, lane, entry) ((base4) + (size_t)((entry) * 7) * (stride) + (lane))
#define TB_Q(stride, q) ((size_t)(q) * (stride))
#endif
S2K_DEV void tb_store(uint32_t* __restrict__ base, size_t stride, size_t lane, int entry, int which, const fe29& v) {
  uint4* e = TB_ENTRY(reinterpret_cast<uint4*>(base), stride, lane, entry);
  uint4* q = e + TB_Q(stride, which * 2);
  q[0] = make_uint4(v.n[0], v.n[1], v.n[2], v.n[3]);
  q[TB_Q(stride, 1)] = make_uint4(v.n[4], v.n[5], v.n[6], v.n[7]);
  reinterpret_cast<uint32_t*>(e + TB_Q(stride, 6))[which] = v.n[8];
}
S2K_DEV fe29 tb_load(const uint32_t* __restrict__ base, size_t stride, size_t lane, uint32_t entry, int which) {
  const uint4* e = TB_ENTRY(reinterpret_cast<const uint4*>(base), stride, lane, entry);
  const uint4* q = e + TB_Q(stride, which * 2);
  uint4 a = q[0], b = q[TB_Q(stride, 1)];
  fe29 r;
  r.n[0] = a.x; r.n[1] = a.y; r.n[2] = a.z; r.n[3] = a.w; r.n[4] = b.x; r.n[5] = b.y; r.n[6] = b.z; r.n[7] = b.w;
  r.n[8] = reinterpret_cast<const uint32_t*>(e + TB_Q(stride, 6))[which];
  return r;
}
// the ladder's lookup: x (lam: beta*x) and y of one entry
S2K_DEV void tb_load_xy(const uint32_t* __restrict__ base, size_t stride, size_t lane, uint32_t entry, bool lam, fe29& x, fe29& y) {
  const uint4* e = TB_ENTRY(reinterpret_cast<const uint4*>(base), stride, lane, entry);
  const uint4* qx = e + TB_Q(stride, lam ? 4 : 0);
  uint4 a = qx[0], b = qx[TB_Q(stride, 1)], c = e[TB_Q(stride, 2)], d = e[TB_Q(stride, 3)], t = e[TB_Q(stride, 6)];
  x.n[0] = a.x; x.n[1] = a.y; x.n[2] = a.z; x.n[3] = a.w; x.n[4] = b.x; x.n[5] = b.y; x.n[6] = b.z; x.n[7] = b.w;
  y.n[0] = c.x; y.n[1] = c.y; y.n[2] = c.z; y.n[3] = c.w; y.n[4] = d.x; y.n[5] = d.y; y.n[6] = d.z; y.n[7] = d.w;
  x.n[8] = lam ? t.z : t.x;
  y.n[8] = t.y;
}

// One table entry given by its address (eight contiguous quads, the S2K_QT_PACK == 2 format): the
// per-key tables of the repeated-key path (keyed.hip) are arrays of these.
S2K_DEV void ke_store(uint4* __restrict__ e, int which, const fe29& v) {
  e[which * 2] = make_uint4(v.n[0], v.n[1], v.n[2], v.n[3]);
  e[which * 2 + 1] = make_uint4(v.n[4], v.n[5], v.n[6], v.n[7]);
  reinterpret_cast<uint32_t*>(e + 6)[which] = v.n[8];
}
// a whole entry in seven 16-byte stores
S2K_DEV void ke_store3(uint4* __restrict__ e, const fe29& x, const fe29& y, const fe29& bx) {
  e[0] = make_uint4(x.n[0], x.n[1], x.n[2], x.n[3]);
  e[1] = make_uint4(x.n[4], x.n[5], x.n[6], x.n[7]);
  e[2] = make_uint4(y.n[0], y.n[1], y.n[2], y.n[3]);
  e[3] = make_uint4(y.n[4], y.n[5], y.n[6], y.n[7]);
  e[4] = make_uint4(bx.n[0], bx.n[1], bx.n[2], bx.n[3]);
  e[5] = make_uint4(bx.n[4], bx.n[5], bx.n[6], bx.n[7]);
  e[6] = make_uint4(x.n[8], y.n[8], bx.n[8], 0u);
}
S2K_DEV fe29 ke_load(const uint4* __restrict__ e, int which) {
  uint4 a = e[which * 2], b = e[which * 2 + 1];
  fe29 r;
  r.n[0] = a.x; r.n[1] = a.y; r.n[2] = a.z; r.n[3] = a.w; r.n[4] = b.x; r.n[5] = b.y; r.n[6] = b.z; r.n[7] = b.w;
  r.n[8] = reinterpret_cast<const uint32_t*>(e + 6)[which];
  return r;
}
// a whole entry in seven 16-byte loads
S2K_DEV void ke_load3(const uint4* __restrict__ e, fe29& x, fe29& y, fe29& bx) {
  const uint4 a = e[0], b = e[1], c = e[2], d = e[3], f = e[4], g = e[5], t = e[6];
  x.n[0] = a.x; x.n[1] = a.y; x.n[2] = a.z; x.n[3] = a.w; x.n[4] = b.x; x.n[5] = b.y; x.n[6] = b.z; x.n[7] = b.w;
  y.n[0] = c.x; y.n[1] = c.y; y.n[2] = c.z; y.n[3] = c.w; y.n[4] = d.x; y.n[5] = d.y; y.n[6] = d.z; y.n[7] = d.w;
  bx.n[0] = f.x; bx.n[1] = f.y; bx.n[2] = f.z; bx.n[3] = f.w; bx.n[4] = g.x; bx.n[5] = g.y; bx.n[6] = g.z; bx.n[7] = g.w;
  x.n[8] = t.x;
  y.n[8] = t.y;
  bx.n[8] = t.z;
}
S2K_DEV void ke_load_xy(const uint4* __restrict__ e, bool lam, fe29& x, fe29& y) {
  const uint4* qx = e + (lam ? 4 : 0);
  uint4 a = qx[0], b = qx[1], c = e[2], d = e[3], t = e[6];
  x.n[0] = a.x; x.n[1] = a.y; x.n[2] = a.z; x.n[3] = a.w; x.n[4] = b.x; x.n[5] = b.y; x.n[6] = b.z; x.n[7] = b.w;
  y.n[0] = c.x; y.n[1] = c.y; y.n[2] = c.z; y.n[3] = c.w; y.n[4] = d.x; y.n[5] = d.y; y.n[6] = d.z; y.n[7] = d.w;
  x.n[8] = lam ? t.z : t.x;
  y.n[8] = t.y;
}

// joint entries of a key set (engine_internal.h: KJ_*): x and y in five quads
S2K_DEV void je_store(uint4* __restrict__ e, const fe29& x, const fe29& y) {
  e[0] = make_uint4(x.n[0], x.n[1], x.n[2], x.n[3]);
  e[1] = make_uint4(x.n[4], x.n[5], x.n[6], x.n[7]);
  e[2] = make_uint4(y.n[0], y.n[1], y.n[2], y.n[3]);
  e[3] = make_uint4(y.n[4], y.n[5], y.n[6], y.n[7]);
  e[4] = make_uint4(x.n[8], y.n[8], 0u, 0u);
}
S2K_DEV void je_load(const uint4* __restrict__ e, fe29& x, fe29& y) {
  const uint4 a = e[0], b = e[1], c = e[2], d = e[3], t = e[4];
  x.n[0] = a.x; x.n[1] = a.y; x.n[2] = a.z; x.n[3] = a.w; x.n[4] = b.x; x.n[5] = b.y; x.n[6] = b.z; x.n[7] = b.w;
  y.n[0] = c.x; y.n[1] = c.y; y.n[2] = c.z; y.n[3] = c.w; y.n[4] = d.x; y.n[5] = d.y; y.n[6] = d.z; y.n[7] = d.w;
  x.n[8] = t.x;
  y.n[8] = t.y;
}
// wide joint entries (kjw_geom): x and y as canonical 8 x 32-bit words, four quads
S2K_DEV void jw_store(uint4* __restrict__ e, const fe29& x, const fe29& y) {
  uint32_t xw[8], yw[8];
  fe29_to_words(xw, fe29_normalize(x));
  fe29_to_words(yw, fe29_normalize(y));
  e[0] = make_uint4(xw[0], xw[1], xw[2], xw[3]);
  e[1] = make_uint4(xw[4], xw[5], xw[6], xw[7]);
  e[2] = make_uint4(yw[0], yw[1], yw[2], yw[3]);
  e[3] = make_uint4(yw[4], yw[5], yw[6], yw[7]);
}
S2K_DEV void jw_load(const uint4* __restrict__ e, fe29& x, fe29& y) {
  const uint4 a = e[0], b = e[1], c = e[2], d = e[3];
  const uint32_t xw[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w}, yw[8] = {c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
  x = fe29_from_words(xw);
  y = fe29_from_words(yw);
}
// the same in two steps: the 64 bytes asked for (in flight while the previous addition runs), the limbs cut when they are wanted
struct jw_raw {
  uint4 a, b, c, d;
};
S2K_DEV jw_raw jw_fetch(const uint4* __restrict__ e) {
  jw_raw r;
  r.a = e[0]; r.b = e[1]; r.c = e[2]; r.d = e[3];
  return r;
}
S2K_DEV void jw_point(const jw_raw& r, fe29& x, fe29& y) {
  const uint32_t xw[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w}, yw[8] = {r.c.x, r.c.y, r.c.z, r.c.w, r.d.x, r.d.y, r.d.z, r.d.w};
  x = fe29_from_words(xw);
  y = fe29_from_words(yw);
}
// one field element parked in the first three quads of a joint entry's slot (the build's prefix products)
S2K_DEV void je_store1(uint4* __restrict__ e, const fe29& v) {
  e[0] = make_uint4(v.n[0], v.n[1], v.n[2], v.n[3]);
  e[1] = make_uint4(v.n[4], v.n[5], v.n[6], v.n[7]);
  e[2] = make_uint4(v.n[8], 0u, 0u, 0u);
}
S2K_DEV fe29 je_load1(const uint4* __restrict__ e) {
  const uint4 a = e[0], b = e[1], c = e[2];
  fe29 r;
  r.n[0] = a.x; r.n[1] = a.y; r.n[2] = a.z; r.n[3] = a.w; r.n[4] = b.x; r.n[5] = b.y; r.n[6] = b.z; r.n[7] = b.w;
  r.n[8] = c.x;
  return r;
}

}  // namespace s2k

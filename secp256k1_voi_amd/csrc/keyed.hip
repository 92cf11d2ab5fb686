// keyed.hip — signatures that share public keys: grouping by key and the per-key tables.
//
// The reference's verify (secec/ecdsa.go:392-470) computes u1*G + u2*Q from Q alone every time
// (DoubleScalarMultBasepointVartime, point_mul_glv.go:307): a table of Q's small multiples, then 128
// doublings interleaved with the table additions.  When a batch holds several signatures of one
// key, the doublings can be done once per KEY instead: with 2^(16c) Q (c = 0..7) at hand, the 32
// signed 4-bit digits of a half scalar are consumed four rounds of eight, 12 doublings in all.
// This file does what has to happen before that ladder (k_verify_fast<MODE_ECDSA_KEYED>, engine.hip):
//
//   k_key_insert   one lane per signature: its key goes into an open-addressing hash table whose
//                  slots hold the index of the first signature seen with that key; a hit compares
//                  all 64 key bytes, so a group is exactly the set of signatures with identical key
//                  bytes (whatever the hash does), and the lane takes a rank inside its group
//   k_key_alloc    over the slots: groups of at least `min_group` signatures get a table index and a
//                  contiguous range of ladder lanes, both from ONE 64-bit atomic per workgroup, so that
//                  lane order is table order; k_key_counts then fixes the split of the two-part flow
//   k_key_place    one lane per signature: writes itself into its table's lane range (perm / ptab) or
//                  appends itself to the list of the general kernel (left)
//   k_key_chain / k_key_odd / k_key_cofactors / k_key_scale   the tables: key validation as NewPublicKey
//                  (secec.go:188-216, point_s11n.go:298-307), the chain of 116 doublings per key, the eight
//                  8-entry tables of odd multiples, NO inversion: the key's 65 points are brought to one common
//                  Z (the product W of the nine Z the build produces) and stored as affine points of the curve
//                  isomorphic by W; beta*x column
//
// Nothing here touches the host between the launches; the counts stay on the device and the
// ladder kernels read them.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "engine_internal.h"
#include "fe29_inv.h"
#include "jacobian29.h"
#include "lane_tables.h"

namespace {

// Every thread of the (256-thread) workgroup calls this, threads with nothing to take with want = 0;
// returns the start of this thread's range in what the workgroup took from *counter with ONE atomic
// (same-address atomics cost ~12 ns each on MI355X: one per wave made k_key_alloc 0.65 ms).
S2K_DEV uint32_t block_alloc(uint32_t* counter, uint32_t want, uint32_t* sh /* 8 words of LDS */) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t incl = want;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t v = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t)d) incl += v;
  }
  if (lane == 63u) sh[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t total = sh[0] + sh[1] + sh[2] + sh[3];
    sh[4] = total ? atomicAdd(counter, total) : 0u;
  }
  __syncthreads();
  uint32_t off = sh[4];
  for (uint32_t w = 0; w < wave; ++w) off += sh[w];
  __syncthreads();
  return off + incl - want;
}

S2K_DEV uint32_t mix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x7feb352du;
  h ^= h >> 15;
  h *= 0x846ca68bu;
  h ^= h >> 16;
  return h;
}

constexpr uint32_t KG_MAX_PROBES = 64;
constexpr int KG_SHARE_ROUNDS = 4;

// Rank of this lane's signature in its group `found` (KG_NONE: none): one atomic per lane, except that lanes of this wave
// which share a slot go together.  Up to KG_SHARE_ROUNDS times the first unserved lane collects everyone with its slot (a
// batch under one key, or four: 1 or 4 atomics per wave instead of 64 on the same address, 13 ms -> 0.2 ms); a round
// that serves a single lane ends the sharing (a wave of distinct keys) and the rest go alone.  (Lanes that have left the kernel simply take no part.)
S2K_DEV uint32_t kg_take_rank(uint32_t found, uint32_t* __restrict__ cnt) {
  const uint32_t lane = threadIdx.x & 63u;
  bool pending = found != KG_NONE;
  uint32_t pos = 0;
#pragma unroll 1
  for (int round = 0; round < KG_SHARE_ROUNDS; ++round) {
    const unsigned long long act = __ballot(pending);
    if (!act) break;
    const int leader = __ffsll((long long)act) - 1;
    const uint32_t s0 = __shfl(found, leader, 64);
    const bool mine = pending && found == s0;
    const unsigned long long m = __ballot(mine);
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(&cnt[s0], (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (mine) {
      pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
      pending = false;
    }
    if (__popcll(m) == 1) break;   // nobody shared the leader's slot: a wave of distinct keys, the rest go alone
  }
  if (pending) pos = atomicAdd(&cnt[found], 1u);
  return pos;
}

// slot_of[i]: the hash slot = group of signature i (KG_NONE: probe chain too long, general kernel);
// pos_of[i]: its rank in the group.  rep[] starts as KG_NONE, cnt[] as 0.
// KEYBYTES: 64 (X || Y, ECDSA) or 32 (x-only BIP-340 keys: the whole-batch check sums the coefficients of
// each distinct key, msm.hip)
template <int KEYBYTES>
__global__ void __launch_bounds__(256)
k_key_insert(uint32_t n, const uint8_t* __restrict__ pub, uint32_t hmask, uint64_t seed, uint32_t* __restrict__ rep,
             uint32_t* __restrict__ cnt, uint32_t* __restrict__ slot_of, uint32_t* __restrict__ pos_of) {
  uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint4* k = reinterpret_cast<const uint4*>(pub + (size_t)i * KEYBYTES);
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  const uint4 k0 = k[0], k1 = k[1], k2 = KEYBYTES == 64 ? k[2] : zero4, k3 = KEYBYTES == 64 ? k[3] : zero4;
  // every word of the key goes through the mixer, chained: h = mix(h ^ word).  The difference two keys leave in the
  // state after a word depends on the state before it, i.e. on the seed, so no pair of byte strings collides for every
  // seed (the round-2 hash took X and the last 8 bytes of Y, some words by plain addition: keys differing elsewhere in Y,
  // or by a cancelling add / xor pair, met in one slot whatever the seed, and an engineered batch could walk every lane
  // through all 64 probes).  The 64-bit seed is drawn per context from the operating system.
  uint32_t h = (uint32_t)seed;
  h = mix32(h ^ k0.x); h = mix32(h ^ k0.y); h = mix32(h ^ k0.z); h = mix32(h ^ k0.w);
  h = mix32(h ^ k1.x); h = mix32(h ^ k1.y); h = mix32(h ^ k1.z); h = mix32(h ^ k1.w);
  h ^= (uint32_t)(seed >> 32);
  if (KEYBYTES == 64) {
    h = mix32(h ^ k2.x); h = mix32(h ^ k2.y); h = mix32(h ^ k2.z); h = mix32(h ^ k2.w);
    h = mix32(h ^ k3.x); h = mix32(h ^ k3.y); h = mix32(h ^ k3.z); h = mix32(h ^ k3.w);
  }
  uint32_t s = mix32(h) & hmask;
  uint32_t found = KG_NONE;
#pragma unroll 1
  for (uint32_t probe = 0; probe < KG_MAX_PROBES; ++probe) {
    // a slot never changes once it is taken, so a plain load that sees it taken is final; only a slot
    // seen empty needs the atomic (2^20 compare-and-swaps on ONE address, a batch under a single key,
    // cost 6 ms)
    uint32_t old = __atomic_load_n(&rep[s], __ATOMIC_RELAXED);
    if (old == KG_NONE) old = atomicCAS(&rep[s], KG_NONE, i);
    if (old == KG_NONE) {
      found = s;
      break;
    }
    const uint4* o = reinterpret_cast<const uint4*>(pub + (size_t)old * KEYBYTES);
    const uint4 o0 = o[0], o1 = o[1], o2 = KEYBYTES == 64 ? o[2] : zero4, o3 = KEYBYTES == 64 ? o[3] : zero4;
    uint32_t diff = (o0.x ^ k0.x) | (o0.y ^ k0.y) | (o0.z ^ k0.z) | (o0.w ^ k0.w) | (o1.x ^ k1.x) | (o1.y ^ k1.y) |
                    (o1.z ^ k1.z) | (o1.w ^ k1.w) | (o2.x ^ k2.x) | (o2.y ^ k2.y) | (o2.z ^ k2.z) | (o2.w ^ k2.w) |
                    (o3.x ^ k3.x) | (o3.y ^ k3.y) | (o3.z ^ k3.z) | (o3.w ^ k3.w);
    if (diff == 0) {
      found = s;
      break;
    }
    s = (s + 1) & hmask;
  }
  slot_of[i] = found;
  pos_of[i] = kg_take_rank(found, cnt);
}

constexpr int ALLOC_ITEMS = 16, PLACE_ITEMS = 4;

// Two counts per thread, ONE 64-bit atomic per workgroup (tables in the high word, signatures in the low):
// table indices and ladder-lane ranges are handed out in the same order, so lane order is table order and
// the tables of the first k lanes are the first tables (the two-part flow splits both at one point).
S2K_DEV void block_alloc2(unsigned long long* counter, uint32_t want_hi, uint32_t want_lo, uint32_t* sh /* 16 words */,
                          uint32_t& off_hi, uint32_t& off_lo) {
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  uint32_t ih = want_hi, il = want_lo;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint32_t vh = __shfl_up(ih, d, 64), vl = __shfl_up(il, d, 64);
    if (lane >= (uint32_t)d) {
      ih += vh;
      il += vl;
    }
  }
  if (lane == 63u) {
    sh[wave] = ih;
    sh[4 + wave] = il;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t th = sh[0] + sh[1] + sh[2] + sh[3], tl = sh[4] + sh[5] + sh[6] + sh[7];
    unsigned long long old = (th | tl) ? atomicAdd(counter, ((unsigned long long)th << 32) | tl) : 0ull;
    sh[8] = (uint32_t)(old >> 32);
    sh[9] = (uint32_t)old;
  }
  __syncthreads();
  off_hi = sh[8];
  off_lo = sh[9];
  for (uint32_t w = 0; w < wave; ++w) {
    off_hi += sh[w];
    off_lo += sh[4 + w];
  }
  off_hi += ih - want_hi;
  off_lo += il - want_lo;
  __syncthreads();
}

// (every table holds at least min_group signatures and the host keeps n / min_group <= max_tables, so the
// table count cannot pass max_tables)
__global__ void __launch_bounds__(256)
k_key_alloc(uint32_t slots, uint32_t min_group, const uint32_t* __restrict__ rep, const uint32_t* __restrict__ cnt,
            uint32_t* __restrict__ tix, uint32_t* __restrict__ trep, uint32_t* __restrict__ tbase,
            uint32_t* __restrict__ counters) {
  __shared__ uint32_t sh[16];
  const uint32_t s0 = blockIdx.x * (256 * ALLOC_ITEMS) + threadIdx.x;
  uint32_t c[ALLOC_ITEMS];
  uint32_t ntab = 0, nsig = 0;
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    c[k] = s < slots ? cnt[s] : 0u;
    if (c[k] < min_group) c[k] = 0;          // (min_group >= 1: empty slots drop out too)
    ntab += c[k] ? 1u : 0u;
    nsig += c[k];
  }
  uint32_t t, b;
  block_alloc2(reinterpret_cast<unsigned long long*>(&counters[KG_ALLOC64]), ntab, nsig, sh, t, b);
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    if (s >= slots) continue;
    if (c[k]) {
      tix[s] = t;
      trep[t] = rep[s];
      tbase[t] = b;
      ++t;
      b += c[k];
    } else {
      tix[s] = KG_NONE;
    }
  }
}

// the counts where their readers expect them, and the split of the two-part flow: tables [0, split) are
// built first and their signatures verified while the rest of the tables is being built.  Few tables: no split.
constexpr uint32_t KG_SPLIT_MIN_TABLES = 4096;
// note (S2K_KEYS_ADAPTIVE, engine_internal.h): page-locked host memory; the call's number and how many signatures got a table
__global__ void k_key_counts(uint32_t* __restrict__ tbase, uint32_t* __restrict__ counters, unsigned long long* __restrict__ note,
                             uint32_t note_seq) {
  const unsigned long long both = *reinterpret_cast<const unsigned long long*>(&counters[KG_ALLOC64]);
  const uint32_t T = (uint32_t)(both >> 32), nk = (uint32_t)both;
  if (note) __hip_atomic_store(note, ((unsigned long long)note_seq << 32) | nk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  counters[KG_NTAB] = T;
  counters[KG_NKEYED] = nk;
  tbase[T] = nk;
  const uint32_t split = T < KG_SPLIT_MIN_TABLES ? T : T / 2;
  counters[KG_SPLIT_T] = split;
  counters[KG_SPLIT_LANE] = split == T ? nk : tbase[split];
}

__global__ void __launch_bounds__(256)
k_key_place(uint32_t n, const uint32_t* __restrict__ slot_of, const uint32_t* __restrict__ pos_of,
            const uint32_t* __restrict__ base, bool base_by_table, const uint32_t* __restrict__ tix,
            uint32_t* __restrict__ perm, uint32_t* __restrict__ ptab, uint32_t* __restrict__ left,
            uint32_t* __restrict__ counters) {
  __shared__ uint32_t sh[8];
  const uint32_t i0 = blockIdx.x * (256 * PLACE_ITEMS) + threadIdx.x;
  uint32_t t[PLACE_ITEMS];
  uint32_t nleft = 0;
#pragma unroll
  for (int k = 0; k < PLACE_ITEMS; ++k) {
    const uint32_t i = i0 + k * 256;
    t[k] = KG_NONE;
    if (i < n) {
      const uint32_t s = slot_of[i];
      if (s != KG_NONE) t[k] = tix[s];
      if (t[k] == KG_NONE) ++nleft;
      else {
        const uint32_t p = base[base_by_table ? t[k] : s] + pos_of[i];   // (tbase[table] or base[slot])
        perm[p] = i;
        ptab[p] = t[k];
      }
    }
  }
  uint32_t q = block_alloc(&counters[KG_NLEFT], nleft, sh);
#pragma unroll
  for (int k = 0; k < PLACE_ITEMS; ++k) {
    const uint32_t i = i0 + k * 256;
    if (i < n && t[k] == KG_NONE) left[q++] = i;
  }
}

// ---- per-key tables ----
// Entry layout: lane_tables.h (x, y, beta*x in eight quads).  Entries 8c + j (c < 8, j < 8) hold
// (2j + 1) * 2^(16c) Q, entries 64 and 65 hold L + phi(L) and L - phi(L) for L = 2^116 Q.  Four launches (only the first is a serial chain per key;
// one lane doing everything took 1.5 ms for 2^16 keys, latency bound at one wave per SIMD):
//   k_key_chain  lane per key: key validation, doubling chain from Q; the Jacobian 2^(16c) Q (c = 1..7)
//                are parked in the entries 8c (X, Y, and Z in the beta*x slot), the two lead points (one co-Z
//                addition of L and phi(L), common Z) in the entries 64 and 65
//   k_key_odd    lane per (key, chunk): the common-Z table of odd multiples exactly as k_verify_fast
//                builds its per-signature table, started from the base's JACOBIAN X, Y as if they were
//                affine: neither the doubling nor the addition formula contains the curve constant b,
//                so this computes on the curve isomorphic by the base's Z, and the chunk's points end
//                up with the common Z_total = Z_7 * C * Z_base (scratch slot c)
//   k_key_cofactors lane per key: for each of the nine Z_total (eight chunks and the lead point) the product of the
//                other eight, W / Z_total, by prefix and suffix products (23 products; the inversion this replaces was
//                ~10 k instructions of one serial lane per key on the critical path); W itself stays in the table
//                (scratch element KT_W_SLOT); the two lead points are finished here
//   k_key_scale  lane per (key, chunk): the scaling pass of k_key_odd's table starts from W / Z_total
//                instead of 1: the points come out as (W^2 x, W^3 y), i.e. affine on the curve y^2 = x^3 + 7 W^6
//                that (x, y) -> (W^2 x, W^3 y) maps secp256k1 onto; beta*x column (beta commutes with the map).
//                The ladder's formulas (a = 0, no curve constant) run on that curve unchanged and its result
//                (X, Y, Z) is the secp256k1 point (X, Y, Z W): one product per signature.
// Cost per key: 116 doublings, 8 * (1 doubling + 7 additions), 23 products, ~7 products per point.
template <int CHUNKS>
S2K_DEV uint4* kt_scratch(uint4* kt, int slot, int& which) {   // field elements in the scratch entries, three per entry
  which = slot % 3;
  return kt + (size_t)(kt_geom<CHUNKS>::SCR + slot / 3) * 8;
}
template <int CHUNKS>
S2K_DEV void scr_store(uint4* kt, int slot, const fe29& v) {
  int which;
  uint4* e = kt_scratch<CHUNKS>(kt, slot, which);
  ke_store(e, which, v);
}
template <int CHUNKS>
S2K_DEV fe29 scr_load(uint4* kt, int slot) {
  int which;
  uint4* e = kt_scratch<CHUNKS>(kt, slot, which);
  return ke_load(e, which);
}
S2K_DEV uint32_t table_count(const uint32_t* __restrict__ counters, uint32_t max_tables) {
  uint32_t ntab = counters[KG_NTAB];
  return ntab > max_tables ? max_tables : ntab;
}
// tables [lo, hi) of part `part` of `nparts` (1: all of them; 2: split at counters[KG_SPLIT_T], k_key_scan)
S2K_DEV void table_range(const uint32_t* __restrict__ counters, uint32_t max_tables, uint32_t part, uint32_t nparts,
                         uint32_t& lo, uint32_t& hi) {
  const uint32_t T = table_count(counters, max_tables);
  lo = 0;
  hi = T;
  if (nparts > 1) {
    const uint32_t sp = counters[KG_SPLIT_T];
    if (part) lo = sp;
    else hi = sp;
  }
}

// XONLY: the keys are 32-byte BIP-340 x-only keys, validated and lifted to the even-y point as
// NewSchnorrPublicKey does (schnorr.go:257-275); otherwise 64-byte X || Y.
template <bool XONLY, int CHUNKS>
__global__ void __launch_bounds__(64)
k_key_chain(const uint32_t* __restrict__ counters, uint32_t max_tables, const uint32_t* __restrict__ trep,
            const uint8_t* __restrict__ pub, uint4* __restrict__ ktab, uint8_t* __restrict__ tinfo) {
  using G = kt_geom<CHUNKS>;
  uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= table_count(counters, max_tables)) return;
  const size_t sig = trep[t];
  uint4* kt = ktab + (size_t)t * (G::SLOTS * 8);
  bool ok;
  fe29 qx, qy;
  if constexpr (XONLY) {
    uint32_t xw[8];
    load_be32(xw, pub + sig * 32);
    ok = fe_is_canonical_raw(xw);
    if (!ok) {
#pragma unroll
      for (int i = 0; i < 8; ++i) xw[i] = FE_GX[i];
    }
    qx = fe29_from_words(xw);
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    if (!fe29_sqrt(qy, rhs)) {   // not an x-coordinate of the curve
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
    qy = fe29_normalize(qy);
    qy = fe29_select((qy.n[0] & 1u) != 0, qy, fe29_normalize_weak(fe29_negate(qy, 1)));   // even y
  } else {
    // NewPublicKey: canonical coordinates, on the curve (the identity has no 64-byte encoding)
    apt q;
    load_be32(q.x.v, pub + sig * 64);
    load_be32(q.y.v, pub + sig * 64 + 32);
    ok = fe_is_canonical_raw(q.x.v) && fe_is_canonical_raw(q.y.v);
    if (!ok) {
      q.x = fe_from_limbs(FE_GX);
      q.y = fe_from_limbs(FE_GY);
    }
    qx = fe29_from_words(q.x.v);
    qy = fe29_from_words(q.y.v);
    fe29 rhs = fe29_mul(fe29_sqr(qx), qx);
    rhs.n[0] += 7;
    if (!fe29_eq(fe29_sqr(qy), rhs)) {
      ok = false;
      qx = fe29_from_words(FE_GX);
      qy = fe29_from_words(FE_GY);
    }
  }
  tinfo[t] = ok ? 1 : 0;

  jpt29 cur;
  cur.x = qx;
  cur.y = qy;
  cur.z = fe29_one();
  ke_store(kt, TB_X, cur.x);
  ke_store(kt, TB_Y, cur.y);
  ke_store(kt, TB_BX, cur.z);
#pragma unroll 1
  for (int c = 1; c <= CHUNKS; ++c) {
    const int nd = c < CHUNKS ? G::DBL : 4;   // (8 chunks: L = 2^116 Q, the ladder's 12 doublings make it 16^32 Q; 32 chunks: L = 2^128 Q)
#pragma unroll 1
    for (int j = 0; j < nd; ++j) cur = jpt29_double(cur);
    if (c == CHUNKS) break;                  // the lead point: below
    uint4* e = kt + (size_t)(c * 8) * 8;
    ke_store(e, TB_X, cur.x);
    ke_store(e, TB_Y, cur.y);
    ke_store(e, TB_BX, cur.z);
  }
  // The ladder starts from +-L +- phi(L), L = 2^116 Q (the recoding's leading digit, engine.hip): instead of L and one
  // addition per SIGNATURE, the table holds L + phi(L) and L - phi(L) (the other two combinations are their negatives).
  // L = (X, Y, Z) and phi(L) = (beta X, Y, Z) share their Z, so both sums come from one co-Z addition (Meloni):
  // with A = (X2 - X1)^2, B = X1 A, C = X2 A:  X3 = (Y2 - Y1)^2 - B - C,  Y3 = (Y2 - Y1)(B - X3) - Y1 (C - B),
  // Z3 = Z (X2 - X1) - the same Z3 for the sum (Y2 = Y) and the difference (Y2 = -Y).  8 products per key.
  {
    const fe29 bx = fe29_mul(cur.x, fe29_from_words(FE_BETA));
    const fe29 h = fe29_normalize_weak(fe29_add(bx, fe29_negate(cur.x, 1)));          // X2 - X1 [1] (never 0: no point has x = 0)
    const fe29 z3 = fe29_mul(cur.z, h);
    const fe29 a = fe29_sqr(h);
    const fe29 b = fe29_mul(cur.x, a), c = fe29_mul(bx, a);
    const fe29 bc = fe29_add(b, c);                                                   // [2]
    const fe29 yn = fe29_normalize_weak(cur.y);                                       // [1]
    // sum: Y2 - Y1 = 0
    const fe29 xs = fe29_normalize_weak(fe29_negate(bc, 2));                          // -(B + C)
    const fe29 ys = fe29_mul(fe29_negate(yn, 1), fe29_normalize_weak(fe29_add(c, fe29_negate(b, 1))));   // -Y (C - B)
    // difference: Y2 - Y1 = -2 Y:  X3 = 4 Y^2 - B - C,  Y3 = -Y (B + C - 2 X3)
    const fe29 xd = fe29_normalize_weak(fe29_add(fe29_sqr(fe29_mul_int(yn, 2)), fe29_negate(bc, 2)));
    const fe29 yd = fe29_mul(fe29_negate(yn, 1), fe29_normalize_weak(fe29_add(bc, fe29_negate(fe29_mul_int(xd, 2), 2))));
    uint4* e = kt + (size_t)G::LEAD * 8;
    ke_store3(e, xs, ys, z3);
    ke_store3(e + 8, xd, yd, z3);
  }
}

template <int CHUNKS>
__global__ void __launch_bounds__(256)
k_key_odd(const uint32_t* __restrict__ counters, uint32_t max_tables, uint32_t part, uint32_t nparts, uint4* __restrict__ ktab) {
  uint32_t id = blockIdx.x * 256 + threadIdx.x, lo, hi;
  table_range(counters, max_tables, part, nparts, lo, hi);
  const uint32_t t = lo + id / CHUNKS, c = id % CHUNKS;
  if (t >= hi) return;
  uint4* kt = ktab + (size_t)t * (kt_geom<CHUNKS>::SLOTS * 8);
  uint4* e0 = kt + (size_t)(c * 8) * 8;
  jpt29 a0;
  a0.x = ke_load(e0, TB_X);
  a0.y = ke_load(e0, TB_Y);
  a0.z = fe29_one();
  const fe29 zb = ke_load(e0, TB_BX);
  jpt29 d = jpt29_double(a0);
  fe29 c2 = fe29_sqr(d.z);
  fe29 c3 = fe29_mul(c2, d.z);
  const fe29 dx = d.x, dy = d.y;
  jpt29 cur;
  cur.x = fe29_mul(a0.x, c2);
  cur.y = fe29_mul(a0.y, c3);
  cur.z = fe29_one();
  ke_store(e0, TB_X, cur.x);
  ke_store(e0, TB_Y, cur.y);
#pragma unroll 1
  for (int j = 1; j < 8; ++j) {
    fe29 h;
    cur = jpt29_add_affine(cur, dx, dy, &h);
    ke_store3(e0 + (size_t)j * 8, cur.x, cur.y, h);
  }
  scr_store<CHUNKS>(kt, (int)c, fe29_mul(fe29_mul(cur.z, d.z), zb));
}

template <int CHUNKS>
__global__ void __launch_bounds__(64)
k_key_cofactors(const uint32_t* __restrict__ counters, uint32_t max_tables, uint32_t part, uint32_t nparts, uint4* __restrict__ ktab) {
  using G = kt_geom<CHUNKS>;
  uint32_t lo, hi;
  table_range(counters, max_tables, part, nparts, lo, hi);
  const uint32_t t = lo + blockIdx.x * 64 + threadIdx.x;
  if (t >= hi) return;
  uint4* kt = ktab + (size_t)t * (G::SLOTS * 8);
  uint4* el = kt + (size_t)G::LEAD * 8;
  if constexpr (CHUNKS == 8) {
    // everything this lane needs is loaded up front (nine independent loads: one memory latency)
    fe29 z[CHUNKS + 1], pre[CHUNKS + 1];
#pragma unroll
    for (int c = 0; c < CHUNKS; ++c) z[c] = scr_load<CHUNKS>(kt, c);
    fe29 lx, ly, dx, dy, dz;
    ke_load3(el, lx, ly, z[CHUNKS]);
    ke_load3(el + 8, dx, dy, dz);                                               // (the two lead points share their Z)
    pre[0] = z[0];
#pragma unroll
    for (int c = 1; c <= CHUNKS; ++c) pre[c] = fe29_mul(pre[c - 1], z[c]);   // z_0 ... z_c
    scr_store<CHUNKS>(kt, G::W_SLOT, pre[CHUNKS]);                            // W
    fe29 suf = fe29_one();                                                       // z_(c+1) ... z_8
#pragma unroll
    for (int c = CHUNKS; c >= 0; --c) {
      const fe29 co = c == CHUNKS ? pre[c - 1] : (c > 0 ? fe29_mul(pre[c - 1], suf) : suf);   // W / z_c
      if (c == CHUNKS) {   // the two lead points, finished here
        const fe29 s2 = fe29_sqr(co), s3 = fe29_mul(s2, co);
        ke_store3(el, fe29_mul(lx, s2), fe29_mul(ly, s3), fe29_zero());
        ke_store3(el + 8, fe29_mul(dx, s2), fe29_mul(dy, s3), fe29_zero());
      } else {
        scr_store<CHUNKS>(kt, c, co);
      }
      if (c > 0) suf = c == CHUNKS ? z[c] : fe29_mul(suf, z[c]);
    }
  } else {
    // 33 Z do not fit in registers: the prefix products go through the key's scratch elements (built once per key set)
    fe29 lx, ly, lz, dx, dy, dz;
    ke_load3(el, lx, ly, lz);
    ke_load3(el + 8, dx, dy, dz);
    fe29 pre = scr_load<CHUNKS>(kt, 0);
    scr_store<CHUNKS>(kt, G::PRE_SLOT, pre);
#pragma unroll 1
    for (int c = 1; c < CHUNKS; ++c) {
      pre = fe29_mul(pre, scr_load<CHUNKS>(kt, c));
      scr_store<CHUNKS>(kt, G::PRE_SLOT + c, pre);                              // z_0 ... z_c
    }
    scr_store<CHUNKS>(kt, G::W_SLOT, fe29_mul(pre, lz));                        // W
    {   // the two lead points: cofactor z_0 ... z_(CHUNKS - 1)
      const fe29 s2 = fe29_sqr(pre), s3 = fe29_mul(s2, pre);
      ke_store3(el, fe29_mul(lx, s2), fe29_mul(ly, s3), fe29_zero());
      ke_store3(el + 8, fe29_mul(dx, s2), fe29_mul(dy, s3), fe29_zero());
    }
    fe29 suf = lz;                                                               // z_(c+1) ... z_CHUNKS
#pragma unroll 1
    for (int c = CHUNKS - 1; c >= 0; --c) {
      const fe29 zc = scr_load<CHUNKS>(kt, c);
      const fe29 co = c > 0 ? fe29_mul(scr_load<CHUNKS>(kt, G::PRE_SLOT + c - 1), suf) : suf;   // W / z_c
      scr_store<CHUNKS>(kt, c, co);
      suf = fe29_mul(suf, zc);
    }
  }
}

template <int CHUNKS>
__global__ void __launch_bounds__(256)
k_key_scale(const uint32_t* __restrict__ counters, uint32_t max_tables, uint32_t part, uint32_t nparts, uint4* __restrict__ ktab) {
  uint32_t id = blockIdx.x * 256 + threadIdx.x, lo, hi;
  table_range(counters, max_tables, part, nparts, lo, hi);
  const uint32_t t = lo + id / CHUNKS, c = id % CHUNKS;
  if (t >= hi) return;
  uint4* kt = ktab + (size_t)t * (kt_geom<CHUNKS>::SLOTS * 8);
  uint4* e0 = kt + (size_t)(c * 8) * 8;
  const fe29 beta = fe29_from_words(FE_BETA);
  fe29 rr = scr_load<CHUNKS>(kt, (int)c);
  // entry j - 1 is in flight while entry j is scaled (a lane walks its eight entries one after the other: with the
  // load at the top of each step the kernel waited for memory eight times per lane)
  fe29 ex, ey, eh;
  ke_load3(e0 + (size_t)7 * 8, ex, ey, eh);
#pragma unroll 1
  for (int j = 7; j >= 0; --j) {
    uint4* e = e0 + (size_t)j * 8;
    const fe29 cx = ex, cy = ey, ch = eh;
    if (j > 0) ke_load3(e - 8, ex, ey, eh);
    fe29 r2 = fe29_sqr(rr);
    fe29 r3 = fe29_mul(r2, rr);
    fe29 x = fe29_mul(cx, r2);
    fe29 y = fe29_mul(cy, r3);
    if (j > 0) rr = fe29_mul(rr, ch);                  // H_j: entry j - 1 sits one addition lower
    ke_store3(e, x, y, fe29_mul(x, beta));
  }
}

// The same pass with ONE LANE PER ENTRY (round 5, VERDICT r04 next #4a; NOT the default: see s2k_internal_key_tables).  The lane-per-chunk form above walks its eight
// entries one after the other - eight dependent round trips to memory per lane, 0.44 ms for 2^16 keys at 0.22 of the issue
// slots and 2.7 TB/s: bound by neither.  Here a lane loads its entry at once, the factor of entry j - (W / Z_total) times
// H_(j+1) ... H_7 - comes from a suffix product over the eight lanes of the chunk (three products on lane-shifted copies,
// row_shl:1 / 2 / 4), and a wave reads and writes 8 KiB of consecutive entries.  Nine products per entry instead of six,
// no dependence between entries.
template <int N>
S2K_DEV fe29 fe29_group_shl(const fe29& a, bool keep) {     // lane j of a group of eight reads lane j + N; `keep` false: one
  fe29 r;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const uint32_t v = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)a.n[i], 0x100 + N, 0xF, 0xF, true);   // row_shl:N
    r.n[i] = keep ? v : (i == 0 ? 1u : 0u);
  }
  return r;
}
template <int CHUNKS>
__global__ void __launch_bounds__(256)
k_key_scale_wide(const uint32_t* __restrict__ counters, uint32_t max_tables, uint32_t part, uint32_t nparts, uint4* __restrict__ ktab) {
  // A wave's 64 entries are 8 KiB of consecutive memory.  They come in and go out as eight 1 KiB wave accesses (lane i takes
  // quad i of the block: 64 lanes x 16 bytes in a row) through the wave's own 8 KiB of LDS, where lane l then finds ITS entry
  // (quad k of entry e sits at e * 8 + ((k + e) & 7): the lanes' 128-byte rows are skewed against the banks).  Lane-per-entry
  // loads straight from memory - 64 lines touched by every one of 7 load instructions - were no faster than the
  // lane-per-chunk walk (0.46 against 0.44 ms for 2^16 keys).
  __shared__ uint4 stage[4][64 * 8];
  constexpr uint32_t PER_KEY = CHUNKS * 8;
  const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  const size_t id0 = (size_t)blockIdx.x * 256 + wave * 64;
  uint32_t lo, hi;
  table_range(counters, max_tables, part, nparts, lo, hi);
  const size_t t = lo + id0 / PER_KEY;
  const uint32_t e0 = (uint32_t)(id0 % PER_KEY), c = (e0 + lane) >> 3, j = lane & 7u;
  if (t >= hi) return;                                    // (whole waves: 64 divides the entries of a key)
  uint4* kt = ktab + t * (kt_geom<CHUNKS>::SLOTS * 8);
  uint4* blk = kt + (size_t)e0 * 8;
  uint4* sh = stage[wave];
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    const uint32_t i = q * 64 + lane, e = i >> 3, kq = i & 7u;
    sh[e * 8 + ((kq + e) & 7u)] = blk[i];
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (the wave's LDS operations complete in order; nothing crosses waves)
  uint4 qd[7];
#pragma unroll
  for (uint32_t kq = 0; kq < 7; ++kq) qd[kq] = sh[lane * 8 + ((kq + lane) & 7u)];
  fe29 x, y, h;
  x.n[0] = qd[0].x; x.n[1] = qd[0].y; x.n[2] = qd[0].z; x.n[3] = qd[0].w; x.n[4] = qd[1].x; x.n[5] = qd[1].y; x.n[6] = qd[1].z; x.n[7] = qd[1].w;
  y.n[0] = qd[2].x; y.n[1] = qd[2].y; y.n[2] = qd[2].z; y.n[3] = qd[2].w; y.n[4] = qd[3].x; y.n[5] = qd[3].y; y.n[6] = qd[3].z; y.n[7] = qd[3].w;
  h.n[0] = qd[4].x; h.n[1] = qd[4].y; h.n[2] = qd[4].z; h.n[3] = qd[4].w; h.n[4] = qd[5].x; h.n[5] = qd[5].y; h.n[6] = qd[5].z; h.n[7] = qd[5].w;
  x.n[8] = qd[6].x;
  y.n[8] = qd[6].y;
  h.n[8] = qd[6].z;
  const fe29 co = scr_load<CHUNKS>(kt, (int)c);            // W / Z_total of the chunk (k_key_cofactors)
  fe29 s = fe29_group_shl<1>(h, j < 7);                    // a_j = H_(j+1), a_7 = 1
  s = fe29_mul(s, fe29_group_shl<1>(s, j < 7));
  s = fe29_mul(s, fe29_group_shl<2>(s, j < 6));
  s = fe29_mul(s, fe29_group_shl<4>(s, j < 4));            // H_(j+1) ... H_7
  const fe29 rr = fe29_mul(co, s);
  const fe29 r2 = fe29_sqr(rr), r3 = fe29_mul(r2, rr);
  const fe29 xs = fe29_mul(x, r2), ys = fe29_mul(y, r3);
  const fe29 bx = fe29_mul(xs, fe29_from_words(FE_BETA));
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every lane has read its entry before any lane overwrites the block
  const uint4 out[8] = {make_uint4(xs.n[0], xs.n[1], xs.n[2], xs.n[3]), make_uint4(xs.n[4], xs.n[5], xs.n[6], xs.n[7]),
                        make_uint4(ys.n[0], ys.n[1], ys.n[2], ys.n[3]), make_uint4(ys.n[4], ys.n[5], ys.n[6], ys.n[7]),
                        make_uint4(bx.n[0], bx.n[1], bx.n[2], bx.n[3]), make_uint4(bx.n[4], bx.n[5], bx.n[6], bx.n[7]),
                        make_uint4(xs.n[8], ys.n[8], bx.n[8], 0u), make_uint4(0u, 0u, 0u, 0u)};
#pragma unroll
  for (uint32_t kq = 0; kq < 8; ++kq) sh[lane * 8 + ((kq + lane) & 7u)] = out[kq];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (uint32_t q = 0; q < 8; ++q) {
    const uint32_t i = q * 64 + lane, e = i >> 3, kq = i & 7u;
    blk[i] = sh[e * 8 + ((kq + e) & 7u)];
  }
}

// Grouping for the BIP-340 whole-batch check: EVERY key forms a group; groups of more than
// KG_VGROUP signatures are cut into virtual groups of that size (each gets its own term: the lane that
// sums a group's coefficients walks its members one by one).  vslot[t]: slot of virtual group t,
// tix[s]: first virtual group of slot s.
__global__ void __launch_bounds__(256)
k_key_alloc_all(uint32_t slots, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ base, uint32_t* __restrict__ tix,
                uint32_t* __restrict__ vslot, uint32_t* __restrict__ counters) {
  __shared__ uint32_t sh[8];
  const uint32_t s0 = blockIdx.x * (256 * ALLOC_ITEMS) + threadIdx.x;
  uint32_t c[ALLOC_ITEMS];
  uint32_t ntab = 0, nsig = 0;
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    c[k] = s < slots ? cnt[s] : 0u;
    ntab += (c[k] + KG_VGROUP - 1) / KG_VGROUP;
    nsig += c[k];
  }
  uint32_t t = block_alloc(&counters[KG_NTAB], ntab, sh);
  uint32_t b = block_alloc(&counters[KG_NKEYED], nsig, sh);
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    if (s >= slots) continue;
    if (!c[k]) {
      tix[s] = KG_NONE;
      continue;
    }
    base[s] = b;
    tix[s] = t;
    const uint32_t nv = (c[k] + KG_VGROUP - 1) / KG_VGROUP;
#pragma unroll 1
    for (uint32_t v = 0; v < nv; ++v) vslot[t + v] = s;
    t += nv;
    b += c[k];
  }
}

static uint32_t pow2_at_least(size_t v) {
  uint32_t b = 0;
  while (((size_t)1 << b) < v) ++b;
  return b;
}

}  // namespace

// sizes of one call's grouping arrays and table buffer
struct key_group_sizes {
  uint32_t min_group;
  size_t slots, max_tables, np, tp, kg_bytes, ktab_bytes;
};
static key_group_sizes key_group_plan(const s2k_ctx* ctx, size_t n) {
  key_group_sizes z;
  const uint32_t min_group_asked = ctx->kg_mode == S2K_KEYS_ALWAYS ? 1u : (ctx->kg_min_group ? ctx->kg_min_group : KG_MIN_GROUP);
  uint32_t bits = ctx->kg_hash_bits ? ctx->kg_hash_bits : pow2_at_least(2 * n);
  if (bits < 4) bits = 4;
  if (bits > 30) bits = 30;
  z.slots = (size_t)1 << bits;
  // tables are never refused on the device: the threshold is raised until n / threshold of them fit the cap
  z.min_group = min_group_asked;
  // (kg_table_cap: lowered by s2k_internal_key_reserve when the device could not give the buffer the setting allows)
  const size_t cap = ctx->kg_table_cap && ctx->kg_table_cap < ctx->kg_max_tables ? ctx->kg_table_cap : ctx->kg_max_tables;
  if (n / z.min_group > cap) z.min_group = (uint32_t)((n + cap - 1) / cap);
  z.max_tables = n / z.min_group;
  if (z.max_tables == 0) z.max_tables = 1;
  // grouping arrays: counters | rep, cnt, tix [slots] | slot_of, pos_of, perm, ptab, left [n] | trep, tbase [tables] | tinfo
  z.np = (n + 63) & ~(size_t)63;
  z.tp = ((z.max_tables + 63) & ~(size_t)63) + 64;
  const size_t words = KG_COUNTERS + 3 * z.slots + 5 * z.np + 2 * z.tp;
  z.kg_bytes = words * sizeof(uint32_t) + z.tp;
  z.ktab_bytes = z.max_tables * (size_t)KT_SLOTS * 128;
  return z;
}
// device memory the grouped flow of a batch of n holds on top of the verification workspace (grouping arrays + per-key tables)
__attribute__((visibility("hidden"))) size_t s2k_internal_key_bytes(const s2k_ctx* ctx, size_t n) {
  const key_group_sizes z = key_group_plan(ctx, n);
  return z.kg_bytes + z.ktab_bytes;
}
// grows the context's grouping arrays and table buffer for a batch of n: to be called BEFORE work of the call is put on a
// second stream (growing frees and allocates, which synchronises the device)
// The table buffer is sized by the batch (n / min_group tables of 9 KiB: 2.4 GB at 2^20, 36 GiB at the default cap), whether
// or not keys repeat.  A device that cannot give it - a busy one, several contexts, a large batch - is no reason to fail a
// verification: the cap is halved (which raises the threshold: fewer, longer groups get tables) until the buffer fits, and
// when not even 1024 tables fit the call is told S2K_ERR_NOMEM, which the verification entry points take as "verify
// without tables" (the general ladder; same verdicts).
__attribute__((visibility("hidden"))) int s2k_internal_key_reserve(s2k_ctx* ctx, size_t n, int key_bytes) {
  (void)key_bytes;
  for (;;) {
    const key_group_sizes z = key_group_plan(ctx, n);
    int rc = ctx_reserve(ctx, &ctx->kg, &ctx->kg_bytes, z.kg_bytes);
    if (rc) return rc;
    if (z.ktab_bytes <= ctx->ktab_bytes) return S2K_OK;
    if (ctx->ktab) {
      HIP_TRY(ctx, hipFree(ctx->ktab));
      ctx->ktab = nullptr;
      ctx->ktab_bytes = 0;
    }
    size_t want = z.ktab_bytes;
    if (const size_t lim = s2k_internal_key_table_limit().load()) {   // s2k_set_table_memory_budgets: larger buffers count as unobtainable
      if (want > lim) want = ~(size_t)0 >> 8;
    }
    const hipError_t e = want == (~(size_t)0 >> 8) ? hipErrorOutOfMemory : hipMalloc(&ctx->ktab, want);
    if (e == hipSuccess) {
      ctx->ktab_bytes = z.ktab_bytes;
      return S2K_OK;
    }
    (void)hipGetLastError();
    ctx->ktab = nullptr;
    if (z.max_tables <= 1024) return fail(ctx, S2K_ERR_NOMEM, "no device memory for per-key tables (%zu bytes for %zu tables): verifying without them", z.ktab_bytes, z.max_tables);
    ctx->kg_table_cap = (uint32_t)(z.max_tables / 2);
  }
}

__attribute__((visibility("hidden"))) int s2k_internal_key_group(s2k_ctx* ctx, size_t n, const uint8_t* d_pub, int key_bytes,
                                                                 hipStream_t st, key_groups* out) {
  const key_group_sizes z = key_group_plan(ctx, n);
  const size_t slots = z.slots, max_tables = z.max_tables, np = z.np, tp = z.tp;
  const uint32_t min_group = z.min_group;
  int rc = s2k_internal_key_reserve(ctx, n, key_bytes);   // (a no-op when the caller has reserved already)
  if (rc) return rc;
  uint32_t* w = (uint32_t*)ctx->kg;
  uint32_t* counters = w;
  uint32_t* cnt = w + KG_COUNTERS;      // (counters and cnt[] are cleared together: one fill instead of two)
  uint32_t* rep = cnt + slots;
  uint32_t* tix = rep + slots;
  uint32_t* slot_of = tix + slots;
  uint32_t* pos_of = slot_of + np;
  uint32_t* perm = pos_of + np;
  uint32_t* ptab = perm + np;
  uint32_t* left = ptab + np;
  uint32_t* trep = left + np;
  uint32_t* tbase = trep + tp;
  uint8_t* tinfo = (uint8_t*)(tbase + tp);
  ctx->kg_counters = counters;
  ctx->kg_last_max_tables = (uint32_t)max_tables;
  HIP_TRY(ctx, hipMemsetAsync(counters, 0, (KG_COUNTERS + slots) * sizeof(uint32_t), st));
  HIP_TRY(ctx, hipMemsetAsync(rep, 0xff, slots * sizeof(uint32_t), st));
  if (key_bytes == 64)
    k_key_insert<64><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, d_pub, (uint32_t)(slots - 1), ctx->kg_seed, rep, cnt, slot_of, pos_of);
  else
    k_key_insert<32><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, d_pub, (uint32_t)(slots - 1), ctx->kg_seed, rep, cnt, slot_of, pos_of);
  HIP_TRY(ctx, hipGetLastError());
  k_key_alloc<<<(unsigned)((slots + 256 * ALLOC_ITEMS - 1) / (256 * ALLOC_ITEMS)), 256, 0, st>>>((uint32_t)slots, min_group, rep, cnt, tix, trep, tbase, counters);
  HIP_TRY(ctx, hipGetLastError());
  k_key_counts<<<1, 1, 0, st>>>(tbase, counters, ctx->kg_note_dst, ctx->kg_note_seq);
  ctx->kg_note_dst = nullptr;            // (the note of THIS call: other callers of the grouping leave none)
  HIP_TRY(ctx, hipGetLastError());
  k_key_place<<<(unsigned)((n + 256 * PLACE_ITEMS - 1) / (256 * PLACE_ITEMS)), 256, 0, st>>>((uint32_t)n, slot_of, pos_of, tbase, true, tix, perm, ptab, left, counters);
  HIP_TRY(ctx, hipGetLastError());
  out->counters = counters;
  out->perm = perm;
  out->ptab = ptab;
  out->left = left;
  out->ktab = (const uint4*)ctx->ktab;
  out->jtab = nullptr;
  out->tinfo = tinfo;
  out->trep = trep;
  out->max_tables = (uint32_t)max_tables;
  out->chunks = KT_CHUNKS;
  out->key_bytes = key_bytes;
  return S2K_OK;
}

// the doubling chains of ALL tables (one launch: its duration is one lane's latency whatever the count)
__attribute__((visibility("hidden"))) int s2k_internal_key_chains(s2k_ctx* ctx, const uint8_t* d_pub, hipStream_t st,
                                                                  const key_groups* g) {
  uint4* ktab = const_cast<uint4*>(g->ktab);     // (the context's table buffer, or a key set's own)
  const size_t max_tables = g->max_tables;
  if (g->chunks == KS_CHUNKS)
    k_key_chain<false, KS_CHUNKS><<<(unsigned)((max_tables + 63) / 64), 64, 0, st>>>(g->counters, g->max_tables, g->trep, d_pub, ktab, (uint8_t*)g->tinfo);
  else if (g->key_bytes == 64)
    k_key_chain<false, KT_CHUNKS><<<(unsigned)((max_tables + 63) / 64), 64, 0, st>>>(g->counters, g->max_tables, g->trep, d_pub, ktab, (uint8_t*)g->tinfo);
  else
    k_key_chain<true, KT_CHUNKS><<<(unsigned)((max_tables + 63) / 64), 64, 0, st>>>(g->counters, g->max_tables, g->trep, d_pub, ktab, (uint8_t*)g->tinfo);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}

// odd multiples, inversion, scaling for the tables of part `part` of `nparts`
__attribute__((visibility("hidden"))) int s2k_internal_key_tables(s2k_ctx* ctx, hipStream_t st, const key_groups* g, uint32_t part,
                                                                  uint32_t nparts, hipEvent_t ev_after_odd) {
  uint4* ktab = const_cast<uint4*>(g->ktab);
  const size_t max_tables = g->max_tables;
  // The scaling pass: lane per chunk (k_key_scale) unless S2K_KEY_SCALE_WIDE is set.  The lane-per-entry form through LDS
  // (k_key_scale_wide, round 5) shortens the pass from 0.44 to 0.37 ms and the tables stage by 0.06 ms - and the step got
  // LONGER, 4.93-5.02 against 4.85 ms on the same box (profiles/r05_key_scale_ab.txt): the ladder behind it ran the same
  // number of cycles at 2.21 instead of 2.31 GHz.  The chip holds a power budget, not a clock: nine products per entry
  // instead of six is energy the ladder then does not get.  Kept as a knob because it is the measured answer to "make the
  // scaling pass faster" (VERDICT r04 next #4a).
  static const bool scale_old = getenv("S2K_KEY_SCALE_WIDE") == nullptr;
  if (g->chunks == KS_CHUNKS) {
    k_key_odd<KS_CHUNKS><<<blocks_for(max_tables * KS_CHUNKS), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
    HIP_TRY(ctx, hipGetLastError());
    if (ev_after_odd) HIP_TRY(ctx, hipEventRecord(ev_after_odd, st));
    k_key_cofactors<KS_CHUNKS><<<(unsigned)((max_tables + 63) / 64), 64, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
    HIP_TRY(ctx, hipGetLastError());
    if (scale_old) k_key_scale<KS_CHUNKS><<<blocks_for(max_tables * KS_CHUNKS), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
    else k_key_scale_wide<KS_CHUNKS><<<blocks_for(max_tables * KS_CHUNKS * 8), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
    HIP_TRY(ctx, hipGetLastError());
    return S2K_OK;
  }
  k_key_odd<KT_CHUNKS><<<blocks_for(max_tables * KT_CHUNKS), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
  HIP_TRY(ctx, hipGetLastError());
  if (ev_after_odd) HIP_TRY(ctx, hipEventRecord(ev_after_odd, st));
  k_key_cofactors<KT_CHUNKS><<<(unsigned)((max_tables + 63) / 64), 64, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
  HIP_TRY(ctx, hipGetLastError());
  if (scale_old) k_key_scale<KT_CHUNKS><<<blocks_for(max_tables * KT_CHUNKS), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
  else k_key_scale_wide<KT_CHUNKS><<<blocks_for(max_tables * KT_CHUNKS * 8), 256, 0, st>>>(g->counters, g->max_tables, part, nparts, ktab);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}

// ---------------------------------------------------------------------------------------
// Key sets (s2k_keyset_*, engine.hip): tables of a fixed list of keys built ONCE; a verification call names each
// signature's key by its index in the list.  What is left of the grouping is a counting sort of the signatures by key
// index, so that the lanes of the ladder that share a table sit together: ranks as in k_key_insert (without the hash
// table: the slot IS the key index), ranges from one atomic per workgroup, placement by k_key_place with the identity as
// "table of slot".
// ---------------------------------------------------------------------------------------
namespace {
__global__ void __launch_bounds__(256)
k_ks_rank(uint32_t n, uint32_t nkeys, const uint32_t* __restrict__ kidx, uint32_t* __restrict__ cnt, uint32_t* __restrict__ slot_of,
          uint32_t* __restrict__ pos_of) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = kidx[i];
  const uint32_t s = k < nkeys ? k : KG_NONE;           // an index outside the set: no key, the signature is invalid
  slot_of[i] = s;
  pos_of[i] = kg_take_rank(s, cnt);
}
__global__ void __launch_bounds__(256)
k_ks_alloc(uint32_t nkeys, const uint32_t* __restrict__ cnt, uint32_t* __restrict__ base, uint32_t* __restrict__ counters) {
  __shared__ uint32_t sh[8];
  const uint32_t s0 = blockIdx.x * (256 * ALLOC_ITEMS) + threadIdx.x;
  uint32_t c[ALLOC_ITEMS];
  uint32_t nsig = 0;
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    c[k] = s < nkeys ? cnt[s] : 0u;
    nsig += c[k];
  }
  uint32_t b = block_alloc(&counters[KG_NKEYED], nsig, sh);
#pragma unroll
  for (int k = 0; k < ALLOC_ITEMS; ++k) {
    const uint32_t s = s0 + k * 256;
    if (s >= nkeys) continue;
    base[s] = b;
    b += c[k];
  }
}
__global__ void __launch_bounds__(256) k_ks_iota(uint32_t n, uint32_t* __restrict__ a, uint32_t* __restrict__ counters) {
  const uint32_t i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) a[i] = i;
  if (i == 0) counters[KG_NTAB] = n;
}

// Joint tables of a key set (engine_internal.h: KJ_*), one lane per (key, digit position).  The position's eight odd
// multiples E_0 .. E_7 are affine points of the key's isomorphic curve (k_key_scale), and so are their images under the
// endomorphism, phi(E_b) = (beta x_b, y_b).  For every pair (a, b): J[a][b][0] = E_a + phi(E_b), J[a][b][1] = E_a - phi(E_b),
// by the AFFINE addition law (lambda = (y2 - y1) / (x2 - x1): it contains no curve constant, so it holds on the isomorphic
// curve), the 64 denominators beta x_b - x_a - the same for both signs - inverted together (Montgomery's trick, one safegcd
// inversion per lane; the prefix products are parked in the entries' own slots).  x2 = x1 would need (2a+1) = +-(2b+1) lambda
// mod n: never.  Cost per lane: 64 * 9 products and an inversion; once per key set.
__global__ void __launch_bounds__(64)
k_ks_joint(uint32_t nkeys, const uint4* __restrict__ ktab, uint4* __restrict__ jtab) {
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  const uint32_t t = id / KS_CHUNKS, c = id % KS_CHUNKS;
  if (t >= nkeys) return;
  const uint4* e0 = ktab + (size_t)t * (KS_SLOTS * 8) + (size_t)(c * 8) * 8;
  uint4* j0 = jtab + (size_t)t * KJ_KEY_QUADS + (size_t)c * KJ_PER_CHUNK * KJ_ENTRY_QUADS;
  fe29 pre = fe29_one();
#pragma unroll 1
  for (int k = 0; k < 64; ++k) {
    const int a = k >> 3, b = k & 7;
    const fe29 xa = ke_load(e0 + (size_t)a * 8, TB_X), bxb = ke_load(e0 + (size_t)b * 8, TB_BX);
    const fe29 d = fe29_add(bxb, fe29_negate(xa, 1));                            // x2 - x1 [3]
    pre = fe29_mul(pre, d);
    je_store1(j0 + (size_t)(2 * k) * KJ_ENTRY_QUADS, pre);
  }
  fe29 inv = fe29_inv_gcd(pre);
#pragma unroll 1
  for (int k = 63; k >= 0; --k) {
    const int a = k >> 3, b = k & 7;
    fe29 xa, ya, bxa, xb, yb, bxb;
    ke_load3(e0 + (size_t)a * 8, xa, ya, bxa);
    ke_load3(e0 + (size_t)b * 8, xb, yb, bxb);
    const fe29 d = fe29_add(bxb, fe29_negate(xa, 1));
    const fe29 prev = k ? je_load1(j0 + (size_t)(2 * (k - 1)) * KJ_ENTRY_QUADS) : fe29_one();
    const fe29 di = fe29_mul(inv, prev);                                         // 1 / (x2 - x1)
    inv = fe29_mul(inv, d);
    const fe29 nxs = fe29_negate(fe29_add(xa, bxb), 2);                          // -(x1 + x2) [3]
    const fe29 nya = fe29_negate(ya, 1);                                         // -y1 [2]
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      // y2 - y1 with y2 = +-y_b
      const fe29 dy = s_ ? fe29_negate(fe29_add(yb, ya), 2) : fe29_add(yb, nya);   // [3]
      const fe29 lam = fe29_mul(dy, di);                                         // [1]
      const fe29 x3 = fe29_sqr_plus(lam, nxs);                                   // lambda^2 - x1 - x2 [1]
      const fe29 y3 = fe29_mul_plus(lam, fe29_add(xa, fe29_negate(x3, 1)), nya); // lambda (x1 - x3) - y1 [1]
      je_store(j0 + (size_t)(2 * k + s_) * KJ_ENTRY_QUADS, x3, y3);
    }
  }
}

// ---- wide joint tables (kjw_geom<W>, engine_internal.h), built once per key set: inversions are affordable here ----
// affine doubling and addition on y^2 = x^3 + b for any b (the formulas contain no curve constant: they hold on the key's
// isomorphic curve), one safegcd inversion each; operands and results with 1 unit
S2K_DEV void aff_double(fe29& x, fe29& y) {
  const fe29 inv = fe29_inv_gcd(fe29_add(y, y));                                        // 1 / 2y
  const fe29 lam = fe29_mul(fe29_mul_int(fe29_sqr(x), 3), inv);                          // 3 x^2 / 2y  ([3] x [1])
  const fe29 x3 = fe29_sqr_plus(lam, fe29_negate(fe29_add(x, x), 2));                    // lambda^2 - 2x
  y = fe29_mul_plus(lam, fe29_add(x, fe29_negate(x3, 1)), fe29_negate(y, 1));            // lambda (x - x3) - y
  x = x3;
}
S2K_DEV void aff_add(const fe29& x1, const fe29& y1, const fe29& x2, const fe29& y2, fe29& x3, fe29& y3) {
  const fe29 di = fe29_inv_gcd(fe29_add(x2, fe29_negate(x1, 1)));                        // 1 / (x2 - x1)
  const fe29 nya = fe29_negate(y1, 1);
  const fe29 lam = fe29_mul(fe29_add(y2, nya), di);                                      // ([3] x [1])
  x3 = fe29_sqr_plus(lam, fe29_negate(fe29_add(x1, x2), 2));                             // lambda^2 - x1 - x2
  y3 = fe29_mul_plus(lam, fe29_add(x1, fe29_negate(x3, 1)), nya);                        // lambda (x1 - x3) - y1
}
// One lane per (key, position i): B_i = 2^(W i) Q from the 32-chunk table's entry 16^c Q, c = W i / 4, by W i mod 4 doublings;
// then the odd multiples (2a + 1) B_i, a < NE, by repeated addition of 2 B_i.  (x2 = x1 would need 2a + 1 = +-2 mod n.)
template <int W>
__global__ void __launch_bounds__(64)
k_ksw_odd(uint32_t nkeys, const uint4* __restrict__ ktab, uint4* __restrict__ odd) {
  using G = kjw_geom<W>;
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  const uint32_t t = id / G::POS, i = id % G::POS;
  if (t >= nkeys) return;
  const uint32_t bit = (uint32_t)W * i, c = bit >> 2, r = bit & 3u;
  fe29 x, y;
  ke_load_xy(ktab + (size_t)t * (KS_SLOTS * 8) + (size_t)(c * 8) * 8, false, x, y);
#pragma unroll 1
  for (uint32_t k = 0; k < r; ++k) aff_double(x, y);
  uint4* o = odd + ((size_t)t * G::POS + i) * G::NE * KJ_ENTRY_QUADS;
  je_store(o, x, y);
  fe29 dx = x, dy = y;
  aff_double(dx, dy);                                                                   // 2 B_i
#pragma unroll 1
  for (int a = 1; a < G::NE; ++a) {
    fe29 nx, ny;
    aff_add(x, y, dx, dy, nx, ny);
    x = nx;
    y = ny;
    je_store(o + (size_t)a * KJ_ENTRY_QUADS, x, y);
  }
}
// One lane per (key, position): the NE * NE * 2 joint entries, exactly as k_ks_joint does for 8 * 8 * 2 (one inversion for all
// the denominators beta x_b - x_a; the prefix products parked in the entries' own slots); the beta x column is computed here
template <int W>
__global__ void __launch_bounds__(64)
k_ksw_joint(uint32_t nkeys, const uint4* __restrict__ odd, uint4* __restrict__ jtab) {
  using G = kjw_geom<W>;
  const uint32_t id = blockIdx.x * 64 + threadIdx.x;
  const uint32_t t = id / G::POS, c = id % G::POS;
  if (t >= nkeys) return;
  const uint4* e0 = odd + ((size_t)t * G::POS + c) * G::NE * KJ_ENTRY_QUADS;
  uint4* j0 = jtab + (size_t)t * G::KEY_QUADS + (size_t)c * G::PER_POS * G::EQ;
  const fe29 beta = fe29_from_words(FE_BETA);
  constexpr int PAIRS = G::NE * G::NE;
  fe29 pre = fe29_one();
#pragma unroll 1
  for (int k = 0; k < PAIRS; ++k) {
    const int a = k / G::NE, b = k % G::NE;
    fe29 xa, ya, xb, yb;
    je_load(e0 + (size_t)a * KJ_ENTRY_QUADS, xa, ya);
    je_load(e0 + (size_t)b * KJ_ENTRY_QUADS, xb, yb);
    const fe29 d = fe29_add(fe29_mul(xb, beta), fe29_negate(xa, 1));              // x2 - x1 [3]
    pre = fe29_mul(pre, d);
    je_store1(j0 + (size_t)(2 * k) * G::EQ, pre);
  }
  fe29 inv = fe29_inv_gcd(pre);
#pragma unroll 1
  for (int k = PAIRS - 1; k >= 0; --k) {
    const int a = k / G::NE, b = k % G::NE;
    fe29 xa, ya, xb, yb;
    je_load(e0 + (size_t)a * KJ_ENTRY_QUADS, xa, ya);
    je_load(e0 + (size_t)b * KJ_ENTRY_QUADS, xb, yb);
    const fe29 bxb = fe29_mul(xb, beta);
    const fe29 d = fe29_add(bxb, fe29_negate(xa, 1));
    const fe29 prev = k ? je_load1(j0 + (size_t)(2 * (k - 1)) * G::EQ) : fe29_one();
    const fe29 di = fe29_mul(inv, prev);                                         // 1 / (x2 - x1)
    inv = fe29_mul(inv, d);
    const fe29 nxs = fe29_negate(fe29_add(xa, bxb), 2);                          // -(x1 + x2) [3]
    const fe29 nya = fe29_negate(ya, 1);                                         // -y1 [2]
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
      const fe29 dy = s_ ? fe29_negate(fe29_add(yb, ya), 2) : fe29_add(yb, nya);   // y2 - y1 with y2 = +-y_b [3]
      const fe29 lam = fe29_mul(dy, di);
      const fe29 x3 = fe29_sqr_plus(lam, nxs);
      const fe29 y3 = fe29_mul_plus(lam, fe29_add(xa, fe29_negate(x3, 1)), nya);
      jw_store(j0 + (size_t)(2 * k + s_) * G::EQ, x3, y3);
    }
  }
}
// One lane per key: the ladder's two starting points, 2^(W POS) Q +- phi(2^(W POS) Q) = 2^LEAD_SHIFT (L +- phi(L)) from the
// chunk table's lead pair (doubling commutes with phi)
template <int W>
__global__ void __launch_bounds__(64)
k_ksw_lead(uint32_t nkeys, const uint4* __restrict__ ktab, uint4* __restrict__ jtab) {
  using G = kjw_geom<W>;
  const uint32_t t = blockIdx.x * 64 + threadIdx.x;
  if (t >= nkeys) return;
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    fe29 x, y;
    ke_load_xy(ktab + (size_t)t * (KS_SLOTS * 8) + (size_t)(kt_geom<KS_CHUNKS>::LEAD + k) * 8, false, x, y);
#pragma unroll 1
    for (int d = 0; d < G::LEAD_SHIFT; ++d) aff_double(x, y);
    jw_store(jtab + (size_t)t * G::KEY_QUADS + (G::LEAD + k) * G::EQ, x, y);
  }
}
}  // namespace

// device memory of a key set of n keys: keys | tables | validity | identity | counters
__attribute__((visibility("hidden"))) size_t s2k_internal_keyset_bytes(size_t n, size_t off[5]) {
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  off[0] = 0;
  off[1] = off[0] + al(n * 64);
  off[2] = off[1] + al(n * (size_t)KS_SLOTS * 128);           // 36 KiB per key: the 32-chunk geometry (engine_internal.h)
  off[3] = off[2] + al(n + 64);
  off[4] = off[3] + al(n * 4);
  return off[4] + al(KG_COUNTERS * 4);
}
// builds the tables of the n keys at base + off[0] (already on the device) into the set's buffers
__attribute__((visibility("hidden"))) int s2k_internal_keyset_build(s2k_ctx* ctx, uint8_t* base, size_t n, hipStream_t st) {
  size_t off[5];
  (void)s2k_internal_keyset_bytes(n, off);
  key_groups g{};
  g.counters = (uint32_t*)(base + off[4]);
  g.ktab = (const uint4*)(base + off[1]);
  g.tinfo = base + off[2];
  g.trep = (const uint32_t*)(base + off[3]);          // table t is the key of "signature" t: the identity
  g.max_tables = (uint32_t)n;
  g.chunks = KS_CHUNKS;
  g.key_bytes = 64;
  g.part = 0;
  g.nparts = 1;
  HIP_TRY(ctx, hipMemsetAsync(g.counters, 0, KG_COUNTERS * sizeof(uint32_t), st));
  k_ks_iota<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, (uint32_t*)(base + off[3]), g.counters);
  HIP_TRY(ctx, hipGetLastError());
  int rc = s2k_internal_key_chains(ctx, base + off[0], st, &g);
  if (rc) return rc;
  return s2k_internal_key_tables(ctx, st, &g, 0, 1, nullptr);
}
// the joint tables of a key set whose 32-chunk tables are built (same stream, behind them)
__attribute__((visibility("hidden"))) int s2k_internal_keyset_build_joint(s2k_ctx* ctx, const uint8_t* base, size_t n, uint4* joint,
                                                                         hipStream_t st) {
  size_t off[5];
  (void)s2k_internal_keyset_bytes(n, off);
  k_ks_joint<<<(unsigned)((n * KS_CHUNKS + 63) / 64), 64, 0, st>>>((uint32_t)n, (const uint4*)(base + off[1]), joint);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}
// the wide layouts: bytes of the joint tables and of the build's scratch (the odd multiples per position), and the build itself
__attribute__((visibility("hidden"))) size_t s2k_internal_keyset_joint_bytes(size_t n, int w) {
  const size_t quads = w == 6 ? kjw_geom<6>::KEY_QUADS : w == 5 ? kjw_geom<5>::KEY_QUADS : KJ_KEY_QUADS;
  return n * quads * sizeof(uint4);
}
__attribute__((visibility("hidden"))) size_t s2k_internal_keyset_joint_scratch_bytes(size_t n, int w) {
  if (w != 5 && w != 6) return 0;
  const size_t per_key = w == 6 ? (size_t)kjw_geom<6>::POS * kjw_geom<6>::NE : (size_t)kjw_geom<5>::POS * kjw_geom<5>::NE;
  return n * per_key * KJ_ENTRY_QUADS * sizeof(uint4);
}
template <int W>
static int keyset_build_joint_wide(s2k_ctx* ctx, const uint4* ktab, size_t n, uint4* joint, uint4* scratch, hipStream_t st) {
  using G = kjw_geom<W>;
  const unsigned lanes_blocks = (unsigned)((n * G::POS + 63) / 64);
  k_ksw_odd<W><<<lanes_blocks, 64, 0, st>>>((uint32_t)n, ktab, scratch);
  HIP_TRY(ctx, hipGetLastError());
  k_ksw_joint<W><<<lanes_blocks, 64, 0, st>>>((uint32_t)n, scratch, joint);
  HIP_TRY(ctx, hipGetLastError());
  k_ksw_lead<W><<<(unsigned)((n + 63) / 64), 64, 0, st>>>((uint32_t)n, ktab, joint);
  HIP_TRY(ctx, hipGetLastError());
  return S2K_OK;
}
__attribute__((visibility("hidden"))) int s2k_internal_keyset_build_joint_wide(s2k_ctx* ctx, const uint8_t* base, size_t n, int w, uint4* joint,
                                                                              uint4* scratch, hipStream_t st) {
  size_t off[5];
  (void)s2k_internal_keyset_bytes(n, off);
  const uint4* ktab = (const uint4*)(base + off[1]);
  if (w == 5) return keyset_build_joint_wide<5>(ctx, ktab, n, joint, scratch, st);
  if (w == 6) return keyset_build_joint_wide<6>(ctx, ktab, n, joint, scratch, st);
  return fail(ctx, S2K_ERR_ARG, "joint tables of digit width %d", w);
}
// scratch of the sort below, in the context's grouping arrays: counters | cnt, base [nkeys] | slot_of, pos_of, perm, ptab, left [n]
__attribute__((visibility("hidden"))) int s2k_internal_keyset_reserve(s2k_ctx* ctx, size_t nkeys, size_t n) {
  const size_t np = (n + 63) & ~(size_t)63, kp = (nkeys + 63) & ~(size_t)63;
  const size_t words = KG_COUNTERS + 2 * kp + 5 * np;
  if (words * sizeof(uint32_t) > ctx->kg_bytes) ctx->kg_counters = nullptr;
  return ctx_reserve(ctx, &ctx->kg, &ctx->kg_bytes, words * sizeof(uint32_t));
}
// signatures sorted by key index -> the lane lists of the keyed ladder over the set's tables (enqueue only)
__attribute__((visibility("hidden"))) int s2k_internal_keyset_sort(s2k_ctx* ctx, const uint8_t* set_base, size_t nkeys, size_t n,
                                                                   const uint32_t* d_kidx, hipStream_t st, key_groups* out) {
  size_t off[5];
  (void)s2k_internal_keyset_bytes(nkeys, off);
  int rc = s2k_internal_keyset_reserve(ctx, nkeys, n);   // (a no-op when the caller has reserved already)
  if (rc) return rc;
  const size_t np = (n + 63) & ~(size_t)63, kp = (nkeys + 63) & ~(size_t)63;
  uint32_t* w = (uint32_t*)ctx->kg;
  uint32_t* counters = w;
  uint32_t* cnt = w + KG_COUNTERS;
  uint32_t* base = cnt + kp;
  uint32_t* slot_of = base + kp;
  uint32_t* pos_of = slot_of + np;
  uint32_t* perm = pos_of + np;
  uint32_t* ptab = perm + np;
  uint32_t* left = ptab + np;
  ctx->kg_counters = counters;
  ctx->kg_last_max_tables = (uint32_t)nkeys;
  HIP_TRY(ctx, hipMemsetAsync(counters, 0, (KG_COUNTERS + kp) * sizeof(uint32_t), st));     // counters and cnt[]
  k_ks_rank<<<blocks_for(n), 256, 0, st>>>((uint32_t)n, (uint32_t)nkeys, d_kidx, cnt, slot_of, pos_of);
  HIP_TRY(ctx, hipGetLastError());
  k_ks_alloc<<<(unsigned)((nkeys + 256 * ALLOC_ITEMS - 1) / (256 * ALLOC_ITEMS)), 256, 0, st>>>((uint32_t)nkeys, cnt, base, counters);
  HIP_TRY(ctx, hipGetLastError());
  k_key_place<<<(unsigned)((n + 256 * PLACE_ITEMS - 1) / (256 * PLACE_ITEMS)), 256, 0, st>>>(
      (uint32_t)n, slot_of, pos_of, base, false, (const uint32_t*)(set_base + off[3]), perm, ptab, left, counters);
  HIP_TRY(ctx, hipGetLastError());
  *out = key_groups{};
  out->counters = counters;
  out->perm = perm;
  out->ptab = ptab;
  out->left = left;
  out->ktab = (const uint4*)(set_base + off[1]);
  out->jtab = nullptr;
  out->tinfo = set_base + off[2];
  out->trep = (const uint32_t*)(set_base + off[3]);
  out->max_tables = (uint32_t)nkeys;
  out->chunks = KS_CHUNKS;
  out->key_bytes = 64;
  out->part = 0;
  out->nparts = 1;
  return S2K_OK;
}

// grows the grouping arrays for s2k_internal_key_group32 on n keys (before any work of the call is on a second stream)
__attribute__((visibility("hidden"))) int s2k_internal_key_reserve32(s2k_ctx* ctx, size_t n) {
  uint32_t bits = ctx->kg_hash_bits ? ctx->kg_hash_bits : pow2_at_least(2 * n);
  if (bits < 4) bits = 4;
  if (bits > 30) bits = 30;
  const size_t slots = (size_t)1 << bits, np = (n + 63) & ~(size_t)63;
  const size_t words = KG_COUNTERS + 4 * slots + 6 * np;
  if (words * sizeof(uint32_t) > ctx->kg_bytes) ctx->kg_counters = nullptr;   // (of an earlier verification call: the buffer moves)
  return ctx_reserve(ctx, &ctx->kg, &ctx->kg_bytes, words * sizeof(uint32_t));
}

// BIP-340 whole-batch check (msm.hip): all n x-only keys grouped, every group gets (virtual) group indices.
// Synchronises the stream to hand the counts to the host (they size the multiscalar multiplication).
__attribute__((visibility("hidden"))) int s2k_internal_key_group32(s2k_ctx* ctx, size_t n, const uint8_t* d_pk32,
                                                                   hipStream_t st, key_groups32* out) {
  uint32_t bits = ctx->kg_hash_bits ? ctx->kg_hash_bits : pow2_at_least(2 * n);
  if (bits < 4) bits = 4;
  if (bits > 30) bits = 30;
  const size_t slots = (size_t)1 << bits;
  // counters | rep, cnt, base, tix [slots] | slot_of, pos_of, perm, ptab, left, vslot [n]
  const size_t np = (n + 63) & ~(size_t)63;
  ctx->kg_counters = nullptr;   // (of an earlier verification call: these arrays overwrite its counters)
  int rc = s2k_internal_key_reserve32(ctx, n);   // (a no-op when the caller has reserved already)
  if (rc) return rc;
  uint32_t* w = (uint32_t*)ctx->kg;
  uint32_t* counters = w;
  uint32_t* cnt = w + KG_COUNTERS;      // (counters and cnt[] are cleared together)
  uint32_t* rep = cnt + slots;
  uint32_t* base = rep + slots;
  uint32_t* tix = base + slots;
  uint32_t* slot_of = tix + slots;
  uint32_t* pos_of = slot_of + np;
  uint32_t* perm = pos_of + np;
  uint32_t* ptab = perm + np;
  uint32_t* left = ptab + np;
  uint32_t* vslot = left + np;
  HIP_TRY(ctx, hipMemsetAsync(counters, 0, (KG_COUNTERS + slots) * sizeof(uint32_t), st));
  HIP_TRY(ctx, hipMemsetAsync(rep, 0xff, slots * sizeof(uint32_t), st));
  k_key_insert<32><<<blocks_for(n), 256, 0, st>>>((uint32_t)n, d_pk32, (uint32_t)(slots - 1), ctx->kg_seed, rep, cnt, slot_of, pos_of);
  HIP_TRY(ctx, hipGetLastError());
  k_key_alloc_all<<<(unsigned)((slots + 256 * ALLOC_ITEMS - 1) / (256 * ALLOC_ITEMS)), 256, 0, st>>>((uint32_t)slots, cnt, base, tix, vslot, counters);
  HIP_TRY(ctx, hipGetLastError());
  k_key_place<<<(unsigned)((n + 256 * PLACE_ITEMS - 1) / (256 * PLACE_ITEMS)), 256, 0, st>>>((uint32_t)n, slot_of, pos_of, base, false, tix, perm, ptab, left, counters);
  HIP_TRY(ctx, hipGetLastError());
  uint32_t hc[KG_COUNTERS];
  HIP_TRY(ctx, hipMemcpyAsync(hc, counters, sizeof hc, hipMemcpyDeviceToHost, st));
  HIP_TRY(ctx, hipStreamSynchronize(st));
  out->rep = rep;
  out->cnt = cnt;
  out->base = base;
  out->tix = tix;
  out->perm = perm;
  out->left = left;
  out->vslot = vslot;
  out->ngroups = hc[KG_NTAB];
  out->nleft = hc[KG_NLEFT];
  if ((size_t)hc[KG_NKEYED] + hc[KG_NLEFT] != n) return fail(ctx, S2K_ERR_HIP, "internal: key grouping lost signatures");
  return S2K_OK;
}

// ingest.hip — the byte-level work the reference does in Go before `verify` runs (signature and
// public-key (de)serialisation), for whole batches:
//   PublicKey.Verify option handling                     secec/ecdsa.go:171-228
//   NewPublicKey length / prefix dispatch                secec/secec.go:188-216, point_s11n.go:215-230
//   ParseASN1Signature / ParseCompactSignature / BIP-0066 shape check: der.h
// The single-item parsers are host functions (no GPU needed).  The batch entry point uploads the
// raw bytes and parses on the device, one lane per item (k_parse_encoded): at 10^8 verifications
// per second a host loop over DER signatures (about 60 ns each) would be eight times slower than
// the GPU.  Compressed keys are decompressed in the same kernel (one field square root).
#include <cstdint>
#include <cstring>

#include "../../include/secp256k1_voi_amd.h"
#include "der.h"
#include "engine_internal.h"
#include "fe29.h"

namespace {

// A big-endian byte string of n <= 32 bytes, right-aligned in 32 bytes, as the eight 32-bit words the output arrays hold
// (word k = bytes 4k .. 4k + 3 of the 32-byte string, first byte in the low bits: what a 16-byte store puts back in
// order).  Branch-free over n: every output byte is one conditional byte read.
S2K_DEV void be_string_words(uint32_t w[8], const uint8_t* p, uint32_t n) {
  const uint32_t pad = 32u - n;
#pragma unroll
  for (uint32_t k = 0; k < 8; ++k) {
    uint32_t v = 0;
#pragma unroll
    for (uint32_t j = 0; j < 4; ++j) {
      const uint32_t idx = 4 * k + j;
      const uint32_t b = idx >= pad ? (uint32_t)p[idx - pad] : 0u;
      v |= b << (8 * j);
    }
    w[k] = v;
  }
}
// bytesToCanonicalScalar (s11n.go:203-218) with the callers' IsZero test, on such words: 0 < value < n
S2K_DEV bool words_canonical_nonzero(const uint32_t w[8]) {
  const uint32_t order[8] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xfffffffeu, 0xbaaedce6u, 0xaf48a03bu, 0xbfd25e8cu, 0xd0364141u};
  int cmp = 0;
  uint32_t acc = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t v = __builtin_bswap32(w[k]);          // most significant word first
    if (cmp == 0 && v != order[k]) cmp = v < order[k] ? -1 : 1;
    acc |= v;
  }
  return cmp < 0 && acc != 0;
}
S2K_DEV void store32(uint8_t* dst, const uint32_t w[8], bool keep) {
  uint4* o = reinterpret_cast<uint4*>(dst);                // (the output arrays are 256-byte aligned, items 32 or 64 bytes)
  o[0] = keep ? make_uint4(w[0], w[1], w[2], w[3]) : make_uint4(0u, 0u, 0u, 0u);
  o[1] = keep ? make_uint4(w[4], w[5], w[6], w[7]) : make_uint4(0u, 0u, 0u, 0u);
}

// 256 items per workgroup: bytes -> pub X||Y (64), digest (32), r (32), s (32), all big-endian as the verification entry
// point takes them.  Items that fail any pre-check get r = 0, which the verifier's range check rejects.
// The items of a workgroup are contiguous in each of the three blobs, so the workgroup first copies its stretch of every
// blob into LDS with aligned 16-byte loads (coalesced; up to 15 bytes before and after the stretch ride along: the staging
// buffers are padded) and the lanes parse from there; outputs leave as 16-byte stores.  (The first version read and wrote
// single bytes from and to global memory - 170 byte loads and 160 byte stores per item, every one of them a scattered
// access of its own: 0.84 ms per 2^20 items, a sixth of the verification behind it.)  A stretch that does not fit - keys,
// digests or signatures far longer than any valid ones - is read from global memory as before.
constexpr uint32_t PE_PUB_CAP = 256 * 65 + 64, PE_DIG_CAP = 256 * 64 + 64, PE_SIG_CAP = 256 * 80 + 64;   // bytes of LDS per blob
S2K_DEV bool pe_stage(const uint8_t* __restrict__ blob, uint64_t lo, uint64_t hi, uint8_t* lds, uint32_t cap, uint64_t& a) {
  a = lo & ~(uint64_t)15;
  const uint64_t bytes = hi - a;
  if (bytes + 16 > cap) return false;
  for (uint32_t t = threadIdx.x * 16; t < (uint32_t)bytes; t += 256 * 16)
    *reinterpret_cast<uint4*>(lds + t) = *reinterpret_cast<const uint4*>(blob + a + t);
  return true;
}
__global__ void __launch_bounds__(256)
k_parse_encoded(uint32_t first, uint32_t n, const uint8_t* __restrict__ pubs, const uint64_t* __restrict__ pub_off,
                const uint8_t* __restrict__ digests, const uint64_t* __restrict__ dig_off, const uint8_t* __restrict__ sigs,
                const uint64_t* __restrict__ sig_off, int encoding, uint32_t digest_len, int bip66, int low_s, uint8_t* __restrict__ xy,
                uint8_t* __restrict__ dg, uint8_t* __restrict__ rr, uint8_t* __restrict__ ss, uint8_t* __restrict__ recid) {
  __shared__ uint4 lds_pub[PE_PUB_CAP / 16], lds_dig[PE_DIG_CAP / 16], lds_sig[PE_SIG_CAP / 16];
  const uint32_t b0 = blockIdx.x * 256;
  if (b0 >= n) return;
  const uint32_t cnt = n - b0 < 256u ? n - b0 : 256u;
  const size_t i0 = (size_t)first + b0;                  // items [first, first + n) of the batch
  uint64_t a_pub, a_dig, a_sig;
  const bool st_pub = pe_stage(pubs, pub_off[i0], pub_off[i0 + cnt], reinterpret_cast<uint8_t*>(lds_pub), PE_PUB_CAP, a_pub);
  const bool st_dig = pe_stage(digests, dig_off[i0], dig_off[i0 + cnt], reinterpret_cast<uint8_t*>(lds_dig), PE_DIG_CAP, a_dig);
  const bool st_sig = pe_stage(sigs, sig_off[i0], sig_off[i0 + cnt], reinterpret_cast<uint8_t*>(lds_sig), PE_SIG_CAP, a_sig);
  __syncthreads();
  if (threadIdx.x >= cnt) return;
  const size_t i = i0 + threadIdx.x;
  const uint64_t po = pub_off[i], dof = dig_off[i], so = sig_off[i];
  const uint8_t* pk = st_pub ? reinterpret_cast<const uint8_t*>(lds_pub) + (po - a_pub) : pubs + po;
  const size_t pk_len = (size_t)(pub_off[i + 1] - po);
  const uint8_t* d = st_dig ? reinterpret_cast<const uint8_t*>(lds_dig) + (dof - a_dig) : digests + dof;
  const size_t d_len = (size_t)(dig_off[i + 1] - dof);
  const uint8_t* sg = st_sig ? reinterpret_cast<const uint8_t*>(lds_sig) + (so - a_sig) : sigs + so;
  size_t sg_len = (size_t)(sig_off[i + 1] - so);
  uint32_t rw[8], sw[8], qw[16], dw[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rw[j] = sw[j] = dw[j] = 0;
#pragma unroll
  for (int j = 0; j < 16; ++j) qw[j] = 0;
  bool ok = !(digest_len && d_len != digest_len);        // ecdsa.go:186-188
  ok = ok && d_len >= 32;                                // hashToScalar, ecdsa.go:478-480
  if (ok && bip66) {
    ok = s2k_der::is_valid_signature_encoding_bip0066(sg, sg_len) != 0;
    --sg_len;                                            // drop the sighash byte
  }
  uint8_t v = 0;
  if (ok) {
    // 0 ok, 1 malformed encoding, 2 scalar out of range or zero (der.h: parse_asn1_signature / parse_compact_signature,
    // the same structure checks; the scalars come out as words)
    int rc = 1;
    if (encoding == S2K_ENCODING_ASN1) {
      s2k_der::cb_str in{sg, sg_len}, inner, rb, sb;
      if (s2k_der::cb_read_asn1(in, inner, 0x30) && in.n == 0 && s2k_der::cb_read_asn1_integer(inner, rb) &&
          s2k_der::cb_read_asn1_integer(inner, sb) && inner.n == 0) {      // else errInvalidAsn1Sig
        rc = 2;                                                           // errInvalidScalar unless both are canonical
        if (rb.n <= 32 && sb.n <= 32) {                                  // (never 0: ReadASN1Integer refuses empty integers)
          be_string_words(rw, rb.p, (uint32_t)rb.n);
          be_string_words(sw, sb.p, (uint32_t)sb.n);
          if (words_canonical_nonzero(rw) && words_canonical_nonzero(sw)) rc = 0;
        }
      }
    } else if ((encoding == S2K_ENCODING_COMPACT && sg_len == 64) || (encoding != S2K_ENCODING_COMPACT && sg_len == 65)) {
      // ParseCompactSignature (s11n.go:129-144); ParseCompactRecoverableSignature (s11n.go:156-168): [R | S | V], 65 bytes
      be_string_words(rw, sg, 32);
      be_string_words(sw, sg + 32, 32);
      rc = (words_canonical_nonzero(rw) && words_canonical_nonzero(sw)) ? 0 : 2;
      if (encoding != S2K_ENCODING_COMPACT && rc == 0) {
        v = sg[64];
        // the recovery path takes no options: the low-s rule of Verify (ecdsa.go:212) is applied here
        if (low_s) {
          // (n - 1) / 2 (scalar.go:190 IsGreaterThanHalfN)
          const uint32_t half[8] = {0x7fffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0x5d576e73u, 0x57a4501du, 0xdfe92f46u, 0x681b20a0u};
          int cmp = 0;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const uint32_t sv = __builtin_bswap32(sw[j]);
            if (cmp == 0 && sv != half[j]) cmp = sv < half[j] ? -1 : 1;
          }
          if (cmp > 0) rc = 2;
        }
      }
    }
    ok = rc == 0;
  }
  if (ok) {
    if (pk_len == 65 && pk[0] == 0x04) {
      be_string_words(qw, pk + 1, 32);                   // canonical coordinates and the curve equation: k_verify_fast
      be_string_words(qw + 8, pk + 33, 32);
    } else if (pk_len == 33 && (pk[0] == 0x02 || pk[0] == 0x03)) {
      // SetCompressedBytes (point_s11n.go:140-176)
      uint32_t xb[8], xw[8];
      be_string_words(xb, pk + 1, 32);
#pragma unroll
      for (int j = 0; j < 8; ++j) xw[j] = __builtin_bswap32(xb[7 - j]);   // little-endian words of the value
      ok = fe_is_canonical_raw(xw);
      if (ok) {
        fe29 x = fe29_from_words(xw);
        fe29 rhs = fe29_mul(fe29_sqr(x), x);
        rhs.n[0] += 7;
        fe29 y;
        ok = fe29_sqrt(y, rhs);
        y = fe29_normalize(y);
        const bool want_odd = pk[0] == 0x03;
        y = fe29_normalize(fe29_select(((y.n[0] & 1u) != 0) != want_odd, y, fe29_negate(y, 1)));
        uint32_t yw[8];
        fe29_to_words(yw, y);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          qw[j] = xb[j];
          qw[8 + j] = __builtin_bswap32(yw[7 - j]);
        }
      }
    } else {
      ok = false;                                        // bad length / prefix, or the identity (secec.go:206-209)
    }
  }
  if (ok) be_string_words(dw, d, 32);                    // the leftmost 32 bytes (d_len >= 32 was checked)
  store32(xy + i * 64, qw, true);                        // (a key that failed to decode: whatever was decoded, or zeros - r = 0 rejects the item)
  store32(xy + i * 64 + 32, qw + 8, true);
  store32(dg + i * 32, dw, ok);
  store32(rr + i * 32, rw, ok);
  store32(ss + i * 32, sw, ok);
  if (recid) recid[i] = ok ? v : 0xff;                   // (0xff: no recovery id, RecoverPublicKey refuses it)
}

// EncodingCompactRecoverable: valid iff the key recovered from (digest, r, s, v) is the key the caller holds
// (k.Equal(q), ecdsa.go:220-226; PublicKey.Equal compares the serialised points, secec.go:121-129).  A supplied key
// that NewPublicKey would have refused (off the curve, non-canonical: parsed to whatever bytes it carried, or zeros)
// equals no recovered key: those are always canonical points of the curve.
__global__ void __launch_bounds__(256)
k_recovered_equals(uint32_t n, const uint8_t* __restrict__ xy, const uint8_t* __restrict__ rec65, const uint8_t* __restrict__ ok,
                   uint8_t* __restrict__ valid) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint8_t* a = xy + i * 64;
  const uint8_t* b = rec65 + i * 65;
  uint32_t diff = b[0] ^ 0x04u;
  for (int j = 0; j < 64; ++j) diff |= (uint32_t)(a[j] ^ b[1 + j]);
  valid[i] = (ok[i] == 1 && diff == 0) ? 1 : 0;
}

}  // namespace

extern "C" {

int s2k_parse_asn1_signature(const uint8_t* der, size_t len, uint8_t r[32], uint8_t s[32]) {
  if (!der || !r || !s) return S2K_ERR_ARG;
  return s2k_der::parse_asn1_signature(der, len, r, s);
}

int s2k_parse_compact_signature(const uint8_t* sig, size_t len, uint8_t r[32], uint8_t s[32]) {
  if (!sig || !r || !s) return S2K_ERR_ARG;
  return s2k_der::parse_compact_signature(sig, len, r, s);
}

int s2k_is_valid_signature_encoding_bip0066(const uint8_t* d, size_t n) {
  if (!d) return 0;
  return s2k_der::is_valid_signature_encoding_bip0066(d, n);
}

// PublicKey.Verify(digest, sig, opts) for n encoded items (ecdsa.go:171-228).
//   pubs / digests / sigs: concatenated byte strings with n+1 offsets each
//   encoding: S2K_ENCODING_ASN1, S2K_ENCODING_COMPACT, or S2K_ENCODING_COMPACT_RECOVERABLE (the key is recovered from
//             [R | S | V] and compared with the supplied one, ecdsa.go:204-205,220-226)
//   digest_len: 0 = opts == nil (any length >= 32 is taken, leftmost 32 bytes used);
//               otherwise opts.Hash.Size(): other lengths verify false (ecdsa.go:184-188)
//   flags: S2K_ECDSA_REJECT_MALLEABLE, S2K_ECDSA_BIP0066 (bitcoin.VerifyASN1,
//          ecdsa_shitcoin.go:29-35: shape check, strip the sighash byte, low-s, 32-byte digest)
// Public keys are any SEC1 encoding NewPublicKey accepts (33 or 65 bytes).  A malformed key or
// signature makes that item false (the reference could not have constructed the PublicKey / returns
// false from Verify).
// Enqueue only (all work ends on ctx->s_comp; verdicts go to h_out, asynchronously when that is page-locked);
// one_shot: the whole batch in one piece (submit / wait).
// small_out: a synchronous call of up to the small-batch threshold takes its bytes without DMA transfers (s2k_internal_small_block:
// the parse kernel reads the caller's strings from a page-locked block, the verdicts are written there); *small_out is then where
// the caller finds them after the wait, else NULL.
static int encoded_enqueue_inner(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off, const uint8_t* digests,
                                 const uint64_t* dig_off, const uint8_t* sigs, const uint64_t* sig_off, int encoding, size_t digest_len,
                                 uint32_t flags, uint8_t* h_out, bool one_shot, const uint8_t** small_out = nullptr) {
  if (small_out) *small_out = nullptr;
  const bool bip66 = (flags & S2K_ECDSA_BIP0066) != 0;
  const bool recoverable = encoding == S2K_ENCODING_COMPACT_RECOVERABLE;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  const size_t pub_bytes = pub_off[n], dig_bytes = dig_off[n], sig_bytes = sig_off[n];
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t o_pub = 0, o_dig = o_pub + al(pub_bytes), o_sig = o_dig + al(dig_bytes), o_po = o_sig + al(sig_bytes),
               o_do = o_po + al((n + 1) * 8), o_so = o_do + al((n + 1) * 8), o_xy = o_so + al((n + 1) * 8), o_dg = o_xy + al(n * 64),
               o_r = o_dg + al(n * 32), o_s = o_r + al(n * 32), o_v = o_s + al(n * 32), o_id = o_v + al(n),
               o_rec = o_id + (recoverable ? al(n) : 0), o_ok = o_rec + (recoverable ? al(n * 65) : 0),
               total = o_ok + (recoverable ? al(n) : 0);
  int rc = ctx_reserve(ctx, &ctx->io, &ctx->io_bytes, total);
  if (rc) return rc;
  uint8_t* io = (uint8_t*)ctx->io;
  rc = ctx_streams(ctx);
  if (rc) return rc;
  if (small_out && !one_shot && s2k_internal_small_call(ctx, n, flags) && pub_bytes + dig_bytes + sig_bytes <= ((size_t)1 << 20) &&
      (ctx->kg_mode == S2K_KEYS_ADAPTIVE || ctx->kg_mode == S2K_KEYS_OFF)) {
    const size_t sizes[7] = {pub_bytes, dig_bytes, sig_bytes, (n + 1) * 8, (n + 1) * 8, (n + 1) * 8, n};
    uint8_t *h[7], *d[7];
    rc = s2k_internal_small_block(ctx, sizes, 7, h, d);
    if (rc) return rc;
    memcpy(h[0], pubs, pub_bytes);
    memcpy(h[1], digests, dig_bytes);
    memcpy(h[2], sigs, sig_bytes);
    memcpy(h[3], pub_off, (n + 1) * 8);
    memcpy(h[4], dig_off, (n + 1) * 8);
    memcpy(h[5], sig_off, (n + 1) * 8);
    k_parse_encoded<<<(unsigned)((n + 255) / 256), 256, 0, ctx->s_comp>>>(
        0u, (uint32_t)n, d[0], (const uint64_t*)d[3], d[1], (const uint64_t*)d[4], d[2], (const uint64_t*)d[5], encoding,
        (uint32_t)digest_len, bip66 ? 1 : 0, (flags & S2K_ECDSA_REJECT_MALLEABLE) ? 1 : 0, io + o_xy, io + o_dg, io + o_r, io + o_s,
        recoverable ? io + o_id : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    if (recoverable) {
      rc = s2k_ecdsa_recover_batch_device(ctx, n, io + o_dg, io + o_r, io + o_s, io + o_id, 0u, io + o_rec, io + o_ok, ctx->s_comp);
      if (rc) return rc;
      k_recovered_equals<<<(unsigned)((n + 255) / 256), 256, 0, ctx->s_comp>>>((uint32_t)n, io + o_xy, io + o_rec, io + o_ok, d[6]);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      rc = s2k_ecdsa_verify_batch_device(ctx, n, io + o_xy, io + o_dg, io + o_r, io + o_s, flags & S2K_ECDSA_REJECT_MALLEABLE, d[6], ctx->s_comp);
      if (rc) return rc;
    }
    *small_out = h[6];
    return S2K_OK;
  }
  // the offset arrays go up whole (24 bytes per item); the byte strings follow chunk by chunk, each
  // copy overlapping the parse + verification of the previous chunk (same scheme as
  // s2k_ecdsa_verify_batch; chunks are two full rounds of k_verify_fast)
  HIP_TRY(ctx, hipMemcpyAsync(io + o_po, pub_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_copy));
  HIP_TRY(ctx, hipMemcpyAsync(io + o_do, dig_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_copy));
  HIP_TRY(ctx, hipMemcpyAsync(io + o_so, sig_off, (n + 1) * 8, hipMemcpyHostToDevice, ctx->s_copy));
  const size_t round = (size_t)3 * 4 * (size_t)ctx->cu_count * 64;
  // with key grouping on, every chunk groups (and builds tables) on its own, so fewer and larger chunks: two halves
  // (s2k_ecdsa_verify_batch does the same; three chunks of a 2^20 batch left under 6 signatures per key and chunk,
  // below the table threshold: the whole batch went through the general ladder)
  const size_t chunk = (one_shot || n <= 3 * round) ? n : ((ctx->kg_mode != S2K_KEYS_OFF && !recoverable) ? ((n / 2 + 255) & ~(size_t)255) : 2 * round);
  int k = 0;
  for (size_t lo = 0; lo < n; lo += chunk, ++k) {
    const size_t cnt = n - lo < chunk ? n - lo : chunk, hi = lo + cnt;
    HIP_TRY(ctx, hipMemcpyAsync(io + o_pub + pub_off[lo], pubs + pub_off[lo], pub_off[hi] - pub_off[lo], hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipMemcpyAsync(io + o_dig + dig_off[lo], digests + dig_off[lo], dig_off[hi] - dig_off[lo], hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipMemcpyAsync(io + o_sig + sig_off[lo], sigs + sig_off[lo], sig_off[hi] - sig_off[lo], hipMemcpyHostToDevice, ctx->s_copy));
    HIP_TRY(ctx, hipEventRecord(ctx->ev_copied[k & 1], ctx->s_copy));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->s_comp, ctx->ev_copied[k & 1], 0));
    k_parse_encoded<<<(unsigned)((cnt + 255) / 256), 256, 0, ctx->s_comp>>>(
        (uint32_t)lo, (uint32_t)cnt, io + o_pub, (const uint64_t*)(io + o_po), io + o_dig, (const uint64_t*)(io + o_do), io + o_sig,
        (const uint64_t*)(io + o_so), encoding, (uint32_t)digest_len, bip66 ? 1 : 0, (flags & S2K_ECDSA_REJECT_MALLEABLE) ? 1 : 0,
        io + o_xy, io + o_dg, io + o_r, io + o_s, recoverable ? io + o_id : nullptr);
    HIP_TRY(ctx, hipGetLastError());
    if (recoverable) {
      // RecoverPublicKey (ecdsa.go:244-282) on the shared ladder, then k.Equal(q) on the device
      rc = s2k_ecdsa_recover_batch_device(ctx, cnt, io + o_dg + lo * 32, io + o_r + lo * 32, io + o_s + lo * 32, io + o_id + lo,
                                          flags & S2K_ECDSA_FORCE_COMPLETE, io + o_rec + lo * 65, io + o_ok + lo, ctx->s_comp);
      if (rc) return rc;
      k_recovered_equals<<<(unsigned)((cnt + 255) / 256), 256, 0, ctx->s_comp>>>((uint32_t)cnt, io + o_xy + lo * 64, io + o_rec + lo * 65,
                                                                                 io + o_ok + lo, io + o_v + lo);
      HIP_TRY(ctx, hipGetLastError());
    } else {
      rc = s2k_ecdsa_verify_batch_device(ctx, cnt, io + o_xy + lo * 64, io + o_dg + lo * 32, io + o_r + lo * 32, io + o_s + lo * 32,
                                         flags & (S2K_ECDSA_REJECT_MALLEABLE | S2K_ECDSA_FORCE_COMPLETE | S2K_ECDSA_FORCE_WORKLIST), io + o_v + lo, ctx->s_comp);
      if (rc) return rc;
    }
  }
  HIP_TRY(ctx, hipMemcpyAsync(h_out, io + o_v, n, hipMemcpyDeviceToHost, ctx->s_comp));
  return S2K_OK;
}

static int encoded_check_args(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off, const uint8_t* digests,
                              const uint64_t* dig_off, const uint8_t* sigs, const uint64_t* sig_off, int encoding, size_t* digest_len,
                              uint32_t* flags, const uint8_t* valid) {
  if (!pubs || !pub_off || !digests || !dig_off || !sigs || !sig_off || !valid) return fail(ctx, S2K_ERR_ARG, "null buffer");
  if (encoding != S2K_ENCODING_ASN1 && encoding != S2K_ENCODING_COMPACT && encoding != S2K_ENCODING_COMPACT_RECOVERABLE)
    return fail(ctx, S2K_ERR_ARG, "unknown encoding");
  if (n >= S2K_MAX_BATCH) return fail(ctx, S2K_ERR_ARG, "batch too large (at most 2^30 - 1 items per call)");
  if (*flags & S2K_ECDSA_BIP0066) {
    if (encoding != S2K_ENCODING_ASN1) return fail(ctx, S2K_ERR_ARG, "BIP-0066 needs the ASN.1 encoding");
    *digest_len = 32;                                 // optsShitcoin: SHA-256
    *flags |= S2K_ECDSA_REJECT_MALLEABLE;
  }
  return S2K_OK;
}

int s2k_ecdsa_verify_encoded_batch(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off,
                                   const uint8_t* digests, const uint64_t* dig_off, const uint8_t* sigs,
                                   const uint64_t* sig_off, int encoding, size_t digest_len, uint32_t flags,
                                   uint8_t* valid) {
  if (!ctx) return S2K_ERR_ARG;
  if (n == 0) return S2K_OK;
  int rc = encoded_check_args(ctx, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, &digest_len, &flags, valid);
  if (rc) return rc;
  const uint8_t* small_out = nullptr;
  rc = encoded_enqueue_inner(ctx, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, digest_len, flags, valid, /*one_shot=*/false,
                             &small_out);
  if (rc) {
    s2k_internal_drain(ctx);
    return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
  if (small_out) memcpy(valid, small_out, n);
  return S2K_OK;
}

// the same without the wait at the end (s2k_wait / s2k_wait_all; see s2k_ecdsa_verify_batch_submit)
int s2k_ecdsa_verify_encoded_batch_submit(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off,
                                          const uint8_t* digests, const uint64_t* dig_off, const uint8_t* sigs,
                                          const uint64_t* sig_off, int encoding, size_t digest_len, uint32_t flags,
                                          uint8_t* valid, s2k_ticket* ticket) {
  if (!ctx || !ticket) return fail(ctx, S2K_ERR_ARG, "null argument");
  *ticket = 0;
  if (n) {
    int rc = encoded_check_args(ctx, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, &digest_len, &flags, valid);
    if (rc) return rc;
  }
  s2k_ctx::pipe_slot* sl = nullptr;
  int rc = s2k_internal_pipe_slot(ctx, n, valid, &sl);
  if (rc) return rc;
  if (n) {
    rc = encoded_enqueue_inner(sl->ctx, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, digest_len, flags,
                               sl->direct ? valid : sl->h_valid, /*one_shot=*/true);
    if (rc) {
      s2k_internal_drain(sl->ctx);
      return fail(ctx, rc, "%s", sl->ctx->err);
    }
  }
  s2k_internal_pipe_issue(ctx, sl, ticket);
  return S2K_OK;
}

}  // extern "C"

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

// one lane per item: bytes -> pub X||Y (64), digest (32), r (32), s (32), all big-endian as the
// verification entry point takes them.  Items that fail any pre-check get r = 0, which the
// verifier's range check rejects.
__global__ void __launch_bounds__(256)
k_parse_encoded(uint32_t first, uint32_t n, const uint8_t* __restrict__ pubs, const uint64_t* __restrict__ pub_off,
                const uint8_t* __restrict__ digests, const uint64_t* __restrict__ dig_off, const uint8_t* __restrict__ sigs,
                const uint64_t* __restrict__ sig_off, int encoding, uint32_t digest_len, int bip66, int low_s, uint8_t* __restrict__ xy,
                uint8_t* __restrict__ dg, uint8_t* __restrict__ rr, uint8_t* __restrict__ ss, uint8_t* __restrict__ recid) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  i += first;                                            // items [first, first + n) of the batch
  const uint8_t* pk = pubs + pub_off[i];
  size_t pk_len = (size_t)(pub_off[i + 1] - pub_off[i]);
  const uint8_t* d = digests + dig_off[i];
  size_t d_len = (size_t)(dig_off[i + 1] - dig_off[i]);
  const uint8_t* sg = sigs + sig_off[i];
  size_t sg_len = (size_t)(sig_off[i + 1] - sig_off[i]);
  uint8_t r[32], s[32], q[64];
  for (int j = 0; j < 32; ++j) r[j] = s[j] = 0;
  for (int j = 0; j < 64; ++j) q[j] = 0;
  bool ok = !(digest_len && d_len != digest_len);        // ecdsa.go:186-188
  ok = ok && d_len >= 32;                                // hashToScalar, ecdsa.go:478-480
  if (ok && bip66) {
    ok = s2k_der::is_valid_signature_encoding_bip0066(sg, sg_len) != 0;
    --sg_len;                                            // drop the sighash byte
  }
  uint8_t v = 0;
  if (ok) {
    int rc;
    if (encoding == S2K_ENCODING_ASN1) {
      rc = s2k_der::parse_asn1_signature(sg, sg_len, r, s);
    } else if (encoding == S2K_ENCODING_COMPACT) {
      rc = s2k_der::parse_compact_signature(sg, sg_len, r, s);
    } else {   // ParseCompactRecoverableSignature (s11n.go:156-168): [R | S | V], 65 bytes
      rc = sg_len == 65 ? s2k_der::parse_compact_signature(sg, 64, r, s) : 1;
      if (rc == 0) v = sg[64];
      // the recovery path takes no options: the low-s rule of Verify (ecdsa.go:212) is applied here
      if (rc == 0 && low_s) {
        // (n - 1) / 2, big-endian (scalar.go:190 IsGreaterThanHalfN)
        const uint8_t half[32] = {0x7f, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff,
                                  0x5d, 0x57, 0x6e, 0x73, 0x57, 0xa4, 0x50, 0x1d, 0xdf, 0xe9, 0x2f, 0x46, 0x68, 0x1b, 0x20, 0xa0};
        int cmp = 0;
        for (int j = 0; j < 32; ++j)
          if (cmp == 0 && s[j] != half[j]) cmp = s[j] < half[j] ? -1 : 1;
        if (cmp > 0) rc = 2;
      }
    }
    ok = rc == 0;
  }
  if (ok) {
    if (pk_len == 65 && pk[0] == 0x04) {
      for (int j = 0; j < 64; ++j) q[j] = pk[1 + j];     // canonical coordinates and the curve equation: k_verify_fast
    } else if (pk_len == 33 && (pk[0] == 0x02 || pk[0] == 0x03)) {
      // SetCompressedBytes (point_s11n.go:140-176)
      uint32_t xw[8];
      load_be32_unaligned(xw, pk + 1);
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
        for (int j = 0; j < 32; ++j) q[j] = pk[1 + j];
        store_be32_unaligned(q + 32, yw);
      }
    } else {
      ok = false;                                        // bad length / prefix, or the identity (secec.go:206-209)
    }
  }
  uint8_t* oq = xy + i * 64;
  for (int j = 0; j < 64; ++j) oq[j] = q[j];
  for (int j = 0; j < 32; ++j) {
    dg[i * 32 + j] = (ok && d_len >= 32) ? d[j] : 0;
    rr[i * 32 + j] = ok ? r[j] : 0;
    ss[i * 32 + j] = ok ? s[j] : 0;
  }
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
static int encoded_enqueue_inner(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off, const uint8_t* digests,
                                 const uint64_t* dig_off, const uint8_t* sigs, const uint64_t* sig_off, int encoding, size_t digest_len,
                                 uint32_t flags, uint8_t* h_out, bool one_shot) {
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
  if (n > 0x7fffffffu) return fail(ctx, S2K_ERR_ARG, "batch too large");
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
  rc = encoded_enqueue_inner(ctx, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, digest_len, flags, valid, /*one_shot=*/false);
  if (rc) {
    s2k_internal_drain(ctx);
    return rc;
  }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->s_comp));
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

// host.cpp — host side above the device entry points: the byte-level work the reference does
// in Go before `verify` runs (signature and public-key (de)serialisation), for whole batches.
//
//   ParseASN1Signature / ParseCompactSignature          secec/s11n.go:83-108, :129-144
//   bytesToCanonicalScalar                               secec/s11n.go:203-218
//   IsValidSignatureEncodingBIP0066                      secec/bitcoin/asn1_shitcoin.go:13-115
//   PublicKey.Verify option handling                     secec/ecdsa.go:171-228
//   NewPublicKey length / prefix dispatch                secec/secec.go:188-216, point_s11n.go:215-230
//
// DER parsing restates golang.org/x/crypto v0.11.0 `cryptobyte` (go.mod:8 — a dependency that
// is not vendored in the reference): String.ReadASN1 with DER length rules and
// ReadASN1Integer(*[]byte) with minimal-encoding and sign checks.  The arithmetic (point
// decompression, on-curve checks, verification) stays on the GPU: this file only routes bytes.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/secp256k1_voi_amd.h"

namespace {

// n, big-endian
const uint8_t ORDER_BE[32] = {0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xfe,
                              0xba, 0xae, 0xdc, 0xe6, 0xaf, 0x48, 0xa0, 0x3b, 0xbf, 0xd2, 0x5e, 0x8c, 0xd0, 0x36, 0x41, 0x41};

struct cb_str {
  const uint8_t* p;
  size_t n;
};

// cryptobyte String.ReadASN1(out, tag): any-tag read with DER length checks, then tag compare
bool cb_read_asn1(cb_str& s, cb_str& out, uint8_t tag) {
  if (s.n < 2) return false;
  uint8_t t = s.p[0], lb = s.p[1];
  if ((t & 0x1f) == 0x1f) return false;   // high-tag-number form is not supported
  size_t hdr, len;
  if ((lb & 0x80) == 0) {
    hdr = 2;
    len = lb;
  } else {
    unsigned ll = lb & 0x7f;
    if (ll == 0 || ll > 4 || s.n < 2 + (size_t)ll) return false;
    uint32_t l32 = 0;
    for (unsigned i = 0; i < ll; ++i) l32 = (l32 << 8) | s.p[2 + i];
    if (l32 < 128) return false;                        // should have used the short form
    if ((l32 >> ((ll - 1) * 8)) == 0) return false;     // leading zero octet in the length
    hdr = 2 + ll;
    len = l32;
  }
  if (s.n < hdr + len) return false;
  if (t != tag) return false;
  out.p = s.p + hdr;
  out.n = len;
  s.p += hdr + len;
  s.n -= hdr + len;
  return true;
}
// cryptobyte String.ReadASN1Integer(*[]byte)
bool cb_read_asn1_integer(cb_str& s, cb_str& out) {
  cb_str b;
  if (!cb_read_asn1(s, b, 0x02)) return false;
  if (b.n == 0) return false;
  if (b.n > 1 && ((b.p[0] == 0x00 && (b.p[1] & 0x80) == 0) || (b.p[0] == 0xff && (b.p[1] & 0x80) == 0x80))) return false;
  if (b.p[0] & 0x80) return false;   // negative
  while (b.n > 1 && b.p[0] == 0) {
    ++b.p;
    --b.n;
  }
  out = b;
  return true;
}
// bytesToCanonicalScalar (s11n.go:203-218) followed by the IsZero test of the callers
bool canonical_nonzero_scalar(uint8_t out[32], const uint8_t* p, size_t n) {
  if (n > 32 || n == 0) return false;
  memset(out, 0, 32);
  memcpy(out + 32 - n, p, n);
  if (memcmp(out, ORDER_BE, 32) >= 0) return false;   // SetCanonicalBytes (scalar.go:136)
  uint8_t acc = 0;
  for (int i = 0; i < 32; ++i) acc |= out[i];
  return acc != 0;
}

}  // namespace

extern "C" {

int s2k_parse_asn1_signature(const uint8_t* der, size_t len, uint8_t r[32], uint8_t s[32]) {
  if (!der || !r || !s) return S2K_ERR_ARG;
  cb_str in{der, len}, inner, rb, sb;
  if (!cb_read_asn1(in, inner, 0x30) || in.n != 0 || !cb_read_asn1_integer(inner, rb) || !cb_read_asn1_integer(inner, sb) ||
      inner.n != 0)
    return 1;   // errInvalidAsn1Sig
  if (!canonical_nonzero_scalar(r, rb.p, rb.n)) return 2;   // errInvalidScalar
  if (!canonical_nonzero_scalar(s, sb.p, sb.n)) return 2;
  return 0;
}

int s2k_parse_compact_signature(const uint8_t* sig, size_t len, uint8_t r[32], uint8_t s[32]) {
  if (!sig || !r || !s) return S2K_ERR_ARG;
  if (len != 64) return 1;   // errInvalidCompactSig
  if (!canonical_nonzero_scalar(r, sig, 32)) return 2;
  if (!canonical_nonzero_scalar(s, sig + 32, 32)) return 2;
  return 0;
}

// 1 = well-formed (with the trailing sighash byte), 0 = not
int s2k_is_valid_signature_encoding_bip0066(const uint8_t* d, size_t n) {
  if (!d) return 0;
  if (n < 9 || n > 73) return 0;
  if (d[0] != 0x30) return 0;
  if ((size_t)d[1] != n - 3) return 0;
  size_t len_r = d[3];
  if (5 + len_r >= n) return 0;
  size_t len_s = d[5 + len_r];
  if (len_r + len_s + 7 != n) return 0;
  if (d[2] != 0x02) return 0;
  if (len_r == 0) return 0;
  if (d[4] & 0x80) return 0;
  if (len_r > 1 && d[4] == 0x00 && !(d[5] & 0x80)) return 0;
  if (d[len_r + 4] != 0x02) return 0;
  if (len_s == 0) return 0;
  if (d[len_r + 6] & 0x80) return 0;
  if (len_s > 1 && d[len_r + 6] == 0x00 && !(d[len_r + 7] & 0x80)) return 0;
  return 1;
}

// PublicKey.Verify(digest, sig, opts) for n encoded items (ecdsa.go:171-228).
//   pubs / digests / sigs: concatenated byte strings with n+1 offsets each
//   encoding: S2K_ENCODING_ASN1 or S2K_ENCODING_COMPACT (EncodingCompactRecoverable is not a
//             batch verification: it is public-key recovery, ecdsa.go:220-226)
//   digest_len: 0 = opts == nil (any length >= 32 is taken, leftmost 32 bytes used);
//               otherwise opts.Hash.Size(): other lengths verify false (ecdsa.go:184-188)
//   flags: S2K_ECDSA_REJECT_MALLEABLE, S2K_ECDSA_BIP0066 (bitcoin.VerifyASN1,
//          ecdsa_shitcoin.go:29-35: shape check, strip the sighash byte, low-s, 32-byte digest)
// Public keys are any SEC1 encoding NewPublicKey accepts (33 or 65 bytes); compressed keys are
// decompressed on the device.  A malformed key makes that item false (the reference could not
// have constructed the PublicKey).
int s2k_ecdsa_verify_encoded_batch(s2k_ctx* ctx, size_t n, const uint8_t* pubs, const uint64_t* pub_off,
                                   const uint8_t* digests, const uint64_t* dig_off, const uint8_t* sigs,
                                   const uint64_t* sig_off, int encoding, size_t digest_len, uint32_t flags,
                                   uint8_t* valid) {
  if (!ctx) return S2K_ERR_ARG;
  if (n == 0) return S2K_OK;
  if (!pubs || !pub_off || !digests || !dig_off || !sigs || !sig_off || !valid) return S2K_ERR_ARG;
  if (encoding != S2K_ENCODING_ASN1 && encoding != S2K_ENCODING_COMPACT) return S2K_ERR_ARG;
  const bool bip66 = (flags & S2K_ECDSA_BIP0066) != 0;
  if (bip66) {
    if (encoding != S2K_ENCODING_ASN1) return S2K_ERR_ARG;
    digest_len = 32;                                  // optsShitcoin: SHA-256
    flags |= S2K_ECDSA_REJECT_MALLEABLE;
  }
  std::vector<uint8_t> xy(n * 64, 0), dg(n * 32, 0), rr(n * 32, 0), ss(n * 32, 0), pre(n, 0);
  std::vector<uint8_t> comp;       // compressed keys to decompress on the device
  std::vector<size_t> comp_idx;
  for (size_t i = 0; i < n; ++i) {
    const uint8_t* pk = pubs + pub_off[i];
    size_t pk_len = (size_t)(pub_off[i + 1] - pub_off[i]);
    const uint8_t* d = digests + dig_off[i];
    size_t d_len = (size_t)(dig_off[i + 1] - dig_off[i]);
    const uint8_t* sg = sigs + sig_off[i];
    size_t sg_len = (size_t)(sig_off[i + 1] - sig_off[i]);
    if (digest_len && d_len != digest_len) continue;          // ecdsa.go:186-188
    if (d_len < 32) continue;                                  // hashToScalar, ecdsa.go:478-480
    if (bip66) {
      if (!s2k_is_valid_signature_encoding_bip0066(sg, sg_len)) continue;
      --sg_len;                                                // drop the sighash byte
    }
    int rc = encoding == S2K_ENCODING_ASN1 ? s2k_parse_asn1_signature(sg, sg_len, &rr[i * 32], &ss[i * 32])
                                           : s2k_parse_compact_signature(sg, sg_len, &rr[i * 32], &ss[i * 32]);
    if (rc != 0) {
      memset(&rr[i * 32], 0, 32);
      memset(&ss[i * 32], 0, 32);
      continue;
    }
    if (pk_len == 65 && pk[0] == 0x04) {
      memcpy(&xy[i * 64], pk + 1, 64);                         // validated on the device
    } else if (pk_len == 33 && (pk[0] == 0x02 || pk[0] == 0x03)) {
      comp.insert(comp.end(), pk, pk + 33);
      comp_idx.push_back(i);
    } else {
      continue;                                                // bad length / prefix, or the identity (secec.go:206-209)
    }
    memcpy(&dg[i * 32], d, 32);
    pre[i] = 1;
  }
  if (!comp_idx.empty()) {
    std::vector<uint8_t> dec(comp_idx.size() * 65), ok(comp_idx.size());
    int rc = s2k_point_decode_batch(ctx, comp_idx.size(), 33, comp.data(), dec.data(), ok.data());
    if (rc) return rc;
    for (size_t j = 0; j < comp_idx.size(); ++j) {
      size_t i = comp_idx[j];
      if (ok[j]) memcpy(&xy[i * 64], &dec[j * 65 + 1], 64);
      else pre[i] = 0;
    }
  }
  for (size_t i = 0; i < n; ++i)
    if (!pre[i]) memset(&rr[i * 32], 0, 32);                   // r = 0 is rejected by the device's range check
  int rc = s2k_ecdsa_verify_batch(ctx, n, xy.data(), dg.data(), rr.data(), ss.data(),
                                  flags & (S2K_ECDSA_REJECT_MALLEABLE | S2K_ECDSA_FORCE_COMPLETE), valid);
  if (rc) return rc;
  for (size_t i = 0; i < n; ++i) valid[i] = (valid[i] && pre[i]) ? 1 : 0;
  return S2K_OK;
}

}  // extern "C"

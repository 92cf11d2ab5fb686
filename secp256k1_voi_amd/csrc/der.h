// der.h — byte-level parsing of signatures, shared by the host entry points and the device ingest
// kernel (ingest.hip):
//   ParseASN1Signature / ParseCompactSignature          secec/s11n.go:83-108, :129-144
//   bytesToCanonicalScalar                               secec/s11n.go:203-218
//   IsValidSignatureEncodingBIP0066                      secec/bitcoin/asn1_shitcoin.go:13-115
// DER parsing restates golang.org/x/crypto v0.11.0 `cryptobyte` (go.mod:8 — a dependency that is
// not vendored in the reference): String.ReadASN1 with DER length rules and
// ReadASN1Integer(*[]byte) with minimal-encoding and sign checks.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#define S2K_HD __host__ __device__ inline

namespace s2k_der {

struct cb_str {
  const uint8_t* p;
  size_t n;
};

// cryptobyte String.ReadASN1(out, tag): any-tag read with DER length checks, then tag compare
S2K_HD bool cb_read_asn1(cb_str& s, cb_str& out, uint8_t tag) {
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
S2K_HD bool cb_read_asn1_integer(cb_str& s, cb_str& out) {
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
S2K_HD bool canonical_nonzero_scalar(uint8_t out[32], const uint8_t* p, size_t n) {
  // n, big-endian
  const uint8_t order[32] = {0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xff, 0xfe,
                             0xba, 0xae, 0xdc, 0xe6, 0xaf, 0x48, 0xa0, 0x3b, 0xbf, 0xd2, 0x5e, 0x8c, 0xd0, 0x36, 0x41, 0x41};
  if (n > 32 || n == 0) return false;
  for (int i = 0; i < 32; ++i) out[i] = 0;
  for (size_t i = 0; i < n; ++i) out[32 - n + i] = p[i];
  int cmp = 0;   // SetCanonicalBytes (scalar.go:136): value < n
  uint8_t acc = 0;
  for (int i = 0; i < 32; ++i) {
    if (cmp == 0 && out[i] != order[i]) cmp = out[i] < order[i] ? -1 : 1;
    acc |= out[i];
  }
  return cmp < 0 && acc != 0;
}

// 0 ok, 1 malformed encoding, 2 scalar out of range or zero
S2K_HD int parse_asn1_signature(const uint8_t* der, size_t len, uint8_t r[32], uint8_t s[32]) {
  cb_str in{der, len}, inner, rb, sb;
  if (!cb_read_asn1(in, inner, 0x30) || in.n != 0 || !cb_read_asn1_integer(inner, rb) || !cb_read_asn1_integer(inner, sb) ||
      inner.n != 0)
    return 1;   // errInvalidAsn1Sig
  if (!canonical_nonzero_scalar(r, rb.p, rb.n)) return 2;   // errInvalidScalar
  if (!canonical_nonzero_scalar(s, sb.p, sb.n)) return 2;
  return 0;
}
S2K_HD int parse_compact_signature(const uint8_t* sig, size_t len, uint8_t r[32], uint8_t s[32]) {
  if (len != 64) return 1;   // errInvalidCompactSig
  if (!canonical_nonzero_scalar(r, sig, 32)) return 2;
  if (!canonical_nonzero_scalar(s, sig + 32, 32)) return 2;
  return 0;
}
// 1 = well-formed (with the trailing sighash byte), 0 = not
S2K_HD int is_valid_signature_encoding_bip0066(const uint8_t* d, size_t n) {
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

}  // namespace s2k_der

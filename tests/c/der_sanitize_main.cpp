// The DER / compact / BIP-0066 parsers of csrc/der.h (host side) under AddressSanitizer + UBSan: every prefix,
// every single-byte mutation and random garbage derived from a few seed encodings, each in a heap buffer of the
// exact length (so that one byte read past the end is a report).  Built host-only by tests/test_host_parsing.py.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "der.h"

static void feed(const uint8_t* p, size_t n) {
  uint8_t* buf = (uint8_t*)malloc(n ? n : 1);   // exact size: overreads trip ASan
  memcpy(buf, p, n);
  uint8_t r[32], s[32];
  (void)s2k_der::parse_asn1_signature(buf, n, r, s);
  (void)s2k_der::parse_compact_signature(buf, n, r, s);
  (void)s2k_der::is_valid_signature_encoding_bip0066(buf, n);
  free(buf);
}

int main() {
  std::vector<std::vector<uint8_t>> seeds;
  // (r, s) = (1, 1); a 71-byte signature with high bits set; long-form length; BIP-0066 style with sighash byte
  seeds.push_back({0x30, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01});
  {
    std::vector<uint8_t> v = {0x30, 0x45, 0x02, 0x21, 0x00};
    for (int i = 0; i < 32; ++i) v.push_back((uint8_t)(0x80 + i));
    v.push_back(0x02);
    v.push_back(0x20);
    for (int i = 0; i < 32; ++i) v.push_back((uint8_t)(0x10 + i));
    seeds.push_back(v);
    v.push_back(0x01);   // sighash byte
    seeds.push_back(v);
  }
  seeds.push_back({0x30, 0x81, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01});
  seeds.push_back({0x30, 0x84, 0x00, 0x00, 0x00, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01});
  seeds.push_back(std::vector<uint8_t>(64, 0x7f));   // compact
  uint64_t st = 0x243F6A8885A308D3ull;
  auto next = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  size_t runs = 0;
  for (const auto& sd : seeds) {
    for (size_t n = 0; n <= sd.size(); ++n) feed(sd.data(), n), ++runs;                 // every prefix
    for (size_t i = 0; i < sd.size(); ++i)                                               // every byte, several values
      for (int v : {0x00, 0x01, 0x02, 0x30, 0x7f, 0x80, 0x81, 0x84, 0xff}) {
        std::vector<uint8_t> m = sd;
        m[i] = (uint8_t)v;
        feed(m.data(), m.size()), ++runs;
      }
    for (int it = 0; it < 2000; ++it) {                                                  // random splices
      std::vector<uint8_t> m = sd;
      int k = 1 + (int)(next() % 4);
      for (int j = 0; j < k && !m.empty(); ++j) m[next() % m.size()] = (uint8_t)next();
      if (next() & 1) m.resize(next() % (m.size() + 1));
      feed(m.data(), m.size()), ++runs;
    }
  }
  for (int it = 0; it < 20000; ++it) {                                                   // pure garbage
    uint8_t g[80];
    size_t n = next() % 80;
    for (size_t i = 0; i < n; ++i) g[i] = (uint8_t)next();
    if (n > 0 && (next() & 3)) g[0] = 0x30;
    feed(g, n), ++runs;
  }
  printf("ok %zu inputs\n", runs);
  return 0;
}

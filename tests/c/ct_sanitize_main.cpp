// Exercises the constant-time CPU functions under AddressSanitizer / UndefinedBehaviorSanitizer (host build only;
// tests/test_ct_cpu.py compiles this file together with csrc/ct_cpu.cpp).  Exit code 0 = no report, self-consistent.
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "secp256k1_voi_amd.h"

int main() {
  uint8_t k[32], d[32], q[65], out[65], out2[65], x1[32], x2[32], r[32], s[32], rid;
  uint64_t st = 0x9E3779B97F4A7C15ull;
  auto next = [&]() {
    st ^= st << 13;
    st ^= st >> 7;
    st ^= st << 17;
    return st;
  };
  int bad = 0;
  for (int it = 0; it < 40; ++it) {
    for (int i = 0; i < 32; ++i) {
      k[i] = (uint8_t)next();
      d[i] = (uint8_t)next();
    }
    if (it == 0) memset(k, 0, 32);
    if (it == 1) memset(k, 0xFF, 32);
    k[0] &= 0x7f;   // keep below n for the ECDH / signing entry points
    d[0] &= 0x7f;
    d[31] |= 1;
    bad += s2k_ct_scalar_base_mult(d, q) != 0;
    bad += s2k_ct_scalar_mult(k, q, out) != 0;
    // (k * d) G two ways: k * (d G) and the x-coordinate through ECDH with swapped roles
    uint8_t kg[65];
    bad += s2k_ct_scalar_base_mult(k, kg) != 0;
    if (it >= 2) {
      bad += s2k_ct_scalar_mult(d, kg, out2) != 0;
      bad += memcmp(out, out2, 65) != 0;
      k[31] |= 1;
      bad += s2k_ct_scalar_base_mult(k, kg) != 0;
      bad += s2k_ct_ecdh(k, q, x1) != 0;
      bad += s2k_ct_ecdh(d, kg, x2) != 0;
      bad += memcmp(x1, x2, 32) != 0;
      bad += s2k_ct_ecdsa_sign_raw(d, k, k, r, s, &rid) != 0;
    }
  }
  (void)s2k_ct_debug_fe_mul_count();
  printf("%s\n", bad ? "FAILED" : "ok");
  return bad != 0;
}

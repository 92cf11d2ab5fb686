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
  // Point.MultiScalarMult: 0, 1, 2, 5 and 33 terms; sum_i k_i (d_i G) against (sum_i k_i d_i) G is checked in
  // tests/test_ct_cpu.py, here: k*P + (n-k)*P... kept simple: k_i * P_i with P_1 = P_0 and k_1 = -k_0 (mod 2^256 wrap
  // avoided by using small scalars) gives the remaining terms' sum
  {
    const int sizes[5] = {0, 1, 2, 5, 33};
    static uint8_t ks[33 * 32], ps[33 * 65];
    for (int si = 0; si < 5; ++si) {
      int n = sizes[si];
      for (int j = 0; j < n; ++j) {
        for (int i = 0; i < 32; ++i) {
          ks[32 * j + i] = (uint8_t)next();
          d[i] = (uint8_t)next();
        }
        bad += s2k_ct_scalar_base_mult(d, ps + 65 * j) != 0;
      }
      bad += s2k_ct_multi_scalar_mult((size_t)n, n ? ks : nullptr, n ? ps : nullptr, out) != 0;
      if (n == 0) {
        memset(out2, 0, 65);
        bad += memcmp(out, out2, 65) != 0;
      }
      if (n == 1) {
        bad += s2k_ct_scalar_mult(ks, ps, out2) != 0;
        bad += memcmp(out, out2, 65) != 0;
      }
      if (n == 2) {   // k0 P0 + k1 P1 against the two single products added through a third term of scalar 1
        uint8_t a[65], b[65], one[3 * 32] = {0}, three[3 * 65];
        bad += s2k_ct_scalar_mult(ks, ps, a) != 0;
        bad += s2k_ct_scalar_mult(ks + 32, ps + 65, b) != 0;
        one[31] = one[63] = 1;   // 1*a + 1*b + 0*P0
        memcpy(three, a, 65);
        memcpy(three + 65, b, 65);
        memcpy(three + 130, ps, 65);
        bad += s2k_ct_multi_scalar_mult(3, one, three, out2) != 0;
        bad += memcmp(out, out2, 65) != 0;
      }
    }
    ps[0] = 0x05;   // malformed record: rejected, nothing leaked or left allocated
    bad += s2k_ct_multi_scalar_mult(33, ks, ps, out) != S2K_ERR_ARG;
    bad += s2k_ct_multi_scalar_mult(2, nullptr, ps, out) != S2K_ERR_ARG;
  }
  (void)s2k_ct_debug_fe_mul_count();
  printf("%s\n", bad ? "FAILED" : "ok");
  return bad != 0;
}

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
  // the single operations: every entry point on exact-size heap buffers (an overread is a report), group identities as the check
  {
    auto heap = [](const uint8_t* src, size_t n) {
      uint8_t* p = new uint8_t[n];
      memcpy(p, src, n);
      return p;
    };
    uint8_t ident[65] = {0};
    for (int it = 0; it < 30; ++it) {
      for (int i = 0; i < 32; ++i) {
        k[i] = (uint8_t)next();
        d[i] = (uint8_t)next();
      }
      k[0] &= 0x7f;
      d[0] &= 0x7f;
      if (it == 0) memset(k, 0, 32);
      uint8_t pa[65], pb[65];
      bad += s2k_ct_scalar_base_mult(k, pa) != 0;
      bad += s2k_ct_scalar_base_mult(d, pb) != 0;
      uint8_t *A = heap(pa, 65), *B = heap(pb, 65), *O = new uint8_t[65], *O2 = new uint8_t[65], *K = heap(k, 32), *D = heap(d, 32), *S32 = new uint8_t[32],
              *T32 = new uint8_t[32];
      uint64_t f = 9, g = 9;
      // (k + d) G = k G + d G; a - b + b = a; 2a = a + a; -(-a) = a
      bad += s2k_ct_scalar_op(S2K_OP_ADD, K, D, S32) != 0;
      bad += s2k_ct_scalar_base_mult(S32, out) != 0;
      bad += s2k_ct_point_add(A, B, O) != 0 || memcmp(O, out, 65) != 0;
      bad += s2k_ct_point_subtract(O, B, O2) != 0 || memcmp(O2, A, 65) != 0;
      bad += s2k_ct_point_double(A, O) != 0 || s2k_ct_point_add(A, A, O2) != 0 || memcmp(O, O2, 65) != 0;
      bad += s2k_ct_point_negate(A, O) != 0 || s2k_ct_point_conditional_negate(O, 1, O2) != 0 || memcmp(O2, A, 65) != 0;
      bad += s2k_ct_point_add(A, O, O2) != 0 || memcmp(O2, ident, 65) != 0;
      bad += s2k_ct_point_conditional_select(A, B, 0, O) != 0 || memcmp(O, A, 65) != 0;
      bad += s2k_ct_point_conditional_select(A, B, 7, O) != 0 || memcmp(O, B, 65) != 0;
      bad += s2k_ct_point_equal(A, A, &f) != 0 || f != 1 || s2k_ct_point_equal(A, B, &f) != 0 || f != (uint64_t)(memcmp(A, B, 65) == 0);
      bad += s2k_ct_point_is_identity(A, &f) != 0 || f != (uint64_t)(it == 0) || s2k_ct_point_is_y_odd(A, &g) != 0 || g != (uint64_t)(A[64] & 1);
      // scalars: a * a^-1 = 1 (a != 0), a - a = 0, a + (-a) = 0, select / negate / predicates
      bad += s2k_ct_scalar_op(S2K_OP_INV, D, nullptr, S32) != 0 || s2k_ct_scalar_op(S2K_OP_MUL, D, S32, T32) != 0;
      bad += s2k_ct_scalar_predicate(S2K_SCALAR_IS_ZERO, D, nullptr, &f) != 0;
      uint8_t one[32] = {0};
      one[31] = 1;
      if (!f) bad += memcmp(T32, one, 32) != 0;
      bad += s2k_ct_scalar_op(S2K_OP_SUB, D, D, S32) != 0 || s2k_ct_scalar_predicate(S2K_SCALAR_IS_ZERO, S32, nullptr, &f) != 0 || f != 1;
      bad += s2k_ct_scalar_op(S2K_OP_NEG, D, nullptr, S32) != 0 || s2k_ct_scalar_op(S2K_OP_ADD, D, S32, T32) != 0;
      bad += s2k_ct_scalar_predicate(S2K_SCALAR_IS_ZERO, T32, nullptr, &f) != 0 || f != 1;
      bad += s2k_ct_scalar_conditional_negate(D, 1, T32) != 0 || memcmp(T32, S32, 32) != 0;
      bad += s2k_ct_scalar_conditional_select(K, D, 1, T32) != 0 || memcmp(T32, D, 32) != 0;
      bad += s2k_ct_scalar_predicate(S2K_SCALAR_EQUAL, K, D, &f) != 0 || f != (uint64_t)(memcmp(K, D, 32) == 0);
      bad += s2k_ct_scalar_predicate(S2K_SCALAR_IS_GT_HALF_N, D, nullptr, &f) != 0 || s2k_ct_scalar_predicate(S2K_SCALAR_IS_GT_HALF_N, S32, nullptr, &g) != 0;
      bad += s2k_ct_scalar_op(S2K_OP_SQR, D, nullptr, S32) != 0 || s2k_ct_scalar_op(S2K_OP_MUL, D, D, T32) != 0 || memcmp(S32, T32, 32) != 0;
      bad += s2k_ct_scalar_set_bytes(D, S32, &f) != 0 || f != 0 || memcmp(S32, D, 32) != 0;
      // field: x(A) * x(A)^-1 = 1; y(A)^2 has a root and it is +-y(A)
      if (it) {
        bad += s2k_ct_fe_op(S2K_OP_INV, A + 1, nullptr, S32, nullptr) != 0 || s2k_ct_fe_op(S2K_OP_MUL, A + 1, S32, T32, &f) != 0 || memcmp(T32, one, 32) != 0;
        bad += s2k_ct_fe_op(S2K_OP_SQR, A + 33, nullptr, S32, nullptr) != 0 || s2k_ct_fe_op(S2K_OP_SQRT, S32, nullptr, T32, &f) != 0 || f != 1;
        bad += s2k_ct_fe_op(S2K_OP_NEG, T32, nullptr, S32, nullptr) != 0 || (memcmp(T32, A + 33, 32) != 0 && memcmp(S32, A + 33, 32) != 0);
        bad += s2k_ct_fe_op(S2K_OP_ADD, A + 1, A + 33, S32, nullptr) != 0 || s2k_ct_fe_op(S2K_OP_SUB, S32, A + 33, T32, nullptr) != 0 || memcmp(T32, A + 1, 32) != 0;
      }
      // malformed operands: rejected
      O[0] = 0x04;
      memset(O + 1, 0xFF, 64);
      bad += s2k_ct_point_add(A, O, O2) != S2K_ERR_ARG || s2k_ct_point_is_identity(O, &f) != S2K_ERR_ARG;
      memset(S32, 0xFF, 32);
      bad += s2k_ct_scalar_op(S2K_OP_INV, S32, nullptr, T32) != S2K_ERR_ARG || s2k_ct_fe_op(S2K_OP_INV, S32, nullptr, T32, nullptr) != S2K_ERR_ARG;
      bad += s2k_ct_scalar_set_bytes(S32, T32, &f) != 0 || f != 1;
      delete[] A; delete[] B; delete[] O; delete[] O2; delete[] K; delete[] D; delete[] S32; delete[] T32;
    }
  }
  (void)s2k_ct_debug_fe_mul_count();
  printf("%s\n", bad ? "FAILED" : "ok");
  return bad != 0;
}

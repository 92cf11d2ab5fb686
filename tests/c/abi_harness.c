/* abi_harness.c — the C-ABI used the way a cgo shim uses it: plain C, no Python, no torch.
 *
 *   gcc -O2 -I include tests/c/abi_harness.c -o /tmp/abi_harness -L secp256k1_voi_amd -lsecp256k1_voi_amd \
 *       -Wl,-rpath,$PWD/secp256k1_voi_amd
 *   /tmp/abi_harness cpu      host-only entry points (parsers, constant-time twins): runs anywhere
 *   /tmp/abi_harness gpu      context + batch verification of signatures made with the CT signer
 *   /tmp/abi_harness group    (built with -DWITH_ORACLE and the oracle library) submit / wait on one context and a
 *                             two-member group on device 0, verdicts against the synchronous call AND the CPU oracle
 *
 * Exit code 0 = every check passed.  tests/test_c_harness.py builds and runs it. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "secp256k1_voi_amd.h"

static int failures = 0;
#define CHECK(cond, msg)                                        \
  do {                                                          \
    if (!(cond)) {                                              \
      fprintf(stderr, "FAIL %s (line %d)\n", msg, __LINE__);    \
      ++failures;                                               \
    }                                                           \
  } while (0)

static void hex(uint8_t* out, const char* s) {
  for (size_t i = 0; s[2 * i]; ++i) {
    unsigned v;
    sscanf(s + 2 * i, "%2x", &v);
    out[i] = (uint8_t)v;
  }
}

/* generator (point.go:18-21) and 2G, 3G */
static const char* GX = "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798";
static const char* GY = "483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8";
static const char* G2X = "c6047f9441ed7d6d3045406e95c07cd85c778e4b8cef3ca7abac09b95c709ee5";
static const char* G3X = "f9308a019258c31049344f85f89d5229b531c845836f99b08601f113bce036f9";

static int cpu_part(void) {
  uint8_t k[32] = {0}, out[65], g[65], x[32];
  g[0] = 4;
  hex(g + 1, GX);
  hex(g + 33, GY);
  k[31] = 2;
  CHECK(s2k_ct_scalar_base_mult(k, out) == S2K_OK, "ct base mult");
  hex(x, G2X);
  CHECK(out[0] == 4 && memcmp(out + 1, x, 32) == 0, "2*G (ScalarBaseMult)");
  k[31] = 3;
  CHECK(s2k_ct_scalar_mult(k, g, out) == S2K_OK, "ct scalar mult");
  hex(x, G3X);
  CHECK(out[0] == 4 && memcmp(out + 1, x, 32) == 0, "3*G (ScalarMult)");
  /* ECDH is symmetric: x(a * (b G)) == x(b * (a G)) */
  uint8_t a[32], b[32], A[65], B[65], s1[32], s2[32];
  for (int i = 0; i < 32; ++i) {
    a[i] = (uint8_t)(17 * i + 3);
    b[i] = (uint8_t)(29 * i + 5);
  }
  a[0] &= 0x7f;
  b[0] &= 0x7f;
  CHECK(s2k_ct_scalar_base_mult(a, A) == S2K_OK && s2k_ct_scalar_base_mult(b, B) == S2K_OK, "keys");
  CHECK(s2k_ct_ecdh(a, B, s1) == S2K_OK && s2k_ct_ecdh(b, A, s2) == S2K_OK && memcmp(s1, s2, 32) == 0, "ECDH symmetry");
  /* malformed point is refused */
  uint8_t bad[65];
  memcpy(bad, g, 65);
  bad[64] ^= 1;
  CHECK(s2k_ct_scalar_mult(k, bad, out) == S2K_ERR_ARG, "off-curve point refused");
  /* the single operations a Go Point / Scalar method binds to: G + G = 2G = Double(G), 2G + G = 3G, 3G - G = 2G; 2 * 2^-1 = 1 */
  {
    uint8_t g2[65], g3[65], t[65], two[32] = {0}, inv[32], one[32] = {0}, prod[32];
    uint64_t f = 9;
    CHECK(s2k_ct_point_add(g, g, g2) == S2K_OK && s2k_ct_point_double(g, t) == S2K_OK && memcmp(g2, t, 65) == 0, "G + G = Double(G)");
    hex(x, G2X);
    CHECK(g2[0] == 4 && memcmp(g2 + 1, x, 32) == 0, "G + G = 2G");
    CHECK(s2k_ct_point_add(g2, g, g3) == S2K_OK, "2G + G");
    hex(x, G3X);
    CHECK(memcmp(g3 + 1, x, 32) == 0, "2G + G = 3G");
    CHECK(s2k_ct_point_subtract(g3, g, t) == S2K_OK && s2k_ct_point_equal(t, g2, &f) == S2K_OK && f == 1, "3G - G = 2G");
    CHECK(s2k_ct_point_subtract(g, g, t) == S2K_OK && s2k_ct_point_is_identity(t, &f) == S2K_OK && f == 1, "G - G = 0");
    CHECK(s2k_ct_point_add(g, bad, t) == S2K_ERR_ARG, "off-curve operand refused");
    two[31] = 2;
    one[31] = 1;
    CHECK(s2k_ct_scalar_op(S2K_OP_INV, two, NULL, inv) == S2K_OK && s2k_ct_scalar_op(S2K_OP_MUL, two, inv, prod) == S2K_OK &&
              memcmp(prod, one, 32) == 0, "2 * Invert(2) = 1");
    CHECK(s2k_ct_scalar_predicate(S2K_SCALAR_IS_GT_HALF_N, inv, NULL, &f) == S2K_OK && f == 1, "1/2 = (n + 1) / 2 > n / 2");
  }
  /* DER: 30 06 02 01 01 02 01 01 is (r, s) = (1, 1) */
  const uint8_t der[8] = {0x30, 0x06, 0x02, 0x01, 0x01, 0x02, 0x01, 0x01};
  uint8_t r[32], s[32];
  CHECK(s2k_parse_asn1_signature(der, sizeof der, r, s) == 0 && r[31] == 1 && s[31] == 1, "ParseASN1Signature");
  const uint8_t der_bad[9] = {0x30, 0x07, 0x02, 0x02, 0x00, 0x01, 0x02, 0x01, 0x01}; /* non-minimal INTEGER */
  CHECK(s2k_parse_asn1_signature(der_bad, sizeof der_bad, r, s) != 0, "non-minimal DER refused");
  printf("cpu: %s\n", failures ? "FAILED" : "ok");
  return failures;
}

static int gpu_part(void) {
  s2k_ctx* ctx = NULL;
  int rc = s2k_ctx_create(0, &ctx);
  if (rc != S2K_OK) {
    fprintf(stderr, "s2k_ctx_create: %d %s\n", rc, s2k_last_error(NULL));
    return 1;
  }
  enum { N = 4096 };
  uint8_t *pub = malloc(N * 64), *dig = malloc(N * 32), *r = malloc(N * 32), *s = malloc(N * 32), *valid = malloc(N);
  uint8_t d[32], k[32], Q[65];
  for (int i = 0; i < N; ++i) {
    for (int j = 0; j < 32; ++j) {
      d[j] = (uint8_t)(i * 131 + j * 7 + 1);
      k[j] = (uint8_t)(i * 17 + j * 23 + 9);
      dig[i * 32 + j] = (uint8_t)(i * 5 + j * 3);
    }
    d[0] &= 0x7f;
    k[0] &= 0x7f;
    uint8_t rid;
    s2k_ct_scalar_base_mult(d, Q);
    memcpy(pub + i * 64, Q + 1, 64);
    CHECK(s2k_ct_ecdsa_sign_raw(d, dig + i * 32, k, r + i * 32, s + i * 32, &rid) == S2K_OK, "sign");
  }
  /* corrupt every 8th digest */
  for (int i = 0; i < N; i += 8) dig[i * 32 + 5] ^= 0x40;
  /* 4096 signatures would take the wave-per-signature ladder; this part is about the per-key tables (the small-batch
   * ladder runs the same batch further down) */
  CHECK(s2k_ctx_set_small_batch_max(ctx, 0) == S2K_OK && s2k_ctx_set_mid_batch_max(ctx, 0) == S2K_OK, "s2k_ctx_set_small_batch_max");
  rc = s2k_ecdsa_verify_batch(ctx, N, pub, dig, r, s, S2K_ECDSA_REJECT_MALLEABLE, valid);
  CHECK(rc == S2K_OK, "s2k_ecdsa_verify_batch");
  int good = 0, wrong = 0;
  for (int i = 0; i < N; ++i) {
    int expect = (i % 8) != 0;
    good += valid[i];
    wrong += valid[i] != expect;
  }
  CHECK(wrong == 0 && good == N - N / 8, "verdicts");
  /* the keys repeat (256 of them, 16 signatures each): the call above grouped them and went through per-key
   * tables; the same batch with the grouping off must give the same verdicts */
  uint32_t st[4] = {0, 0, 0, 0};
  CHECK(s2k_ctx_key_grouping_stats(ctx, st) == S2K_OK && st[0] == N && st[1] == 256 && st[2] == 0, "grouping statistics");
  uint8_t* valid2 = malloc(N);
  CHECK(s2k_ctx_set_key_grouping(ctx, S2K_KEYS_OFF, 0, 0, 0) == S2K_OK, "s2k_ctx_set_key_grouping");
  rc = s2k_ecdsa_verify_batch(ctx, N, pub, dig, r, s, S2K_ECDSA_REJECT_MALLEABLE, valid2);
  CHECK(rc == S2K_OK && memcmp(valid, valid2, N) == 0, "same verdicts without the grouping");
  CHECK(s2k_ctx_set_small_batch_max(ctx, 8192) == S2K_OK, "s2k_ctx_set_small_batch_max");
  memset(valid2, 0xff, N);
  rc = s2k_ecdsa_verify_batch(ctx, N, pub, dig, r, s, S2K_ECDSA_REJECT_MALLEABLE, valid2);
  CHECK(rc == S2K_OK && memcmp(valid, valid2, N) == 0, "same verdicts from the wave-per-signature ladder");
  CHECK(s2k_ctx_set_small_batch_max(ctx, 0) == S2K_OK && s2k_ctx_set_mid_batch_max(ctx, 32768) == S2K_OK, "s2k_ctx_set_mid_batch_max");
  memset(valid2, 0xff, N);
  rc = s2k_ecdsa_verify_batch(ctx, N, pub, dig, r, s, S2K_ECDSA_REJECT_MALLEABLE, valid2);
  CHECK(rc == S2K_OK && memcmp(valid, valid2, N) == 0, "same verdicts from the four-lanes-per-signature ladder");
  CHECK(s2k_ctx_set_small_batch_max(ctx, 3072) == S2K_OK, "s2k_ctx_set_small_batch_max");
  CHECK(s2k_ctx_key_grouping_stats(ctx, st) == S2K_OK && st[0] == 0 && st[1] == 0, "grouping statistics (off)");
  CHECK(s2k_ctx_set_key_grouping(ctx, 7, 0, 0, 0) == S2K_ERR_ARG, "bad grouping mode refused");
  CHECK(s2k_ctx_set_key_grouping(ctx, S2K_KEYS_AUTO, 0, 0, 0) == S2K_OK, "grouping back on");
  free(valid2);
  /* recovery gives the signer's key back */
  uint8_t* rec = malloc(N * 65), *ok = malloc(N), *rids = calloc(N, 1);
  for (int i = 0; i < N; i += 8) dig[i * 32 + 5] ^= 0x40; /* undo */
  for (int i = 0; i < N; ++i) { /* recovery ids again (cheap) */
    for (int j = 0; j < 32; ++j) {
      d[j] = (uint8_t)(i * 131 + j * 7 + 1);
      k[j] = (uint8_t)(i * 17 + j * 23 + 9);
    }
    d[0] &= 0x7f;
    k[0] &= 0x7f;
    uint8_t r2[32], s2[32];
    s2k_ct_ecdsa_sign_raw(d, dig + i * 32, k, r2, s2, rids + i);
  }
  rc = s2k_ecdsa_recover_batch(ctx, N, dig, r, s, rids, 0, rec, ok);
  CHECK(rc == S2K_OK, "s2k_ecdsa_recover_batch");
  int rec_bad = 0;
  for (int i = 0; i < N; ++i) rec_bad += !(ok[i] && memcmp(rec + i * 65 + 1, pub + i * 64, 64) == 0);
  CHECK(rec_bad == 0, "recovered keys");
  printf("gpu: %s (%d/%d valid, build: %s)\n", failures ? "FAILED" : "ok", good, N, s2k_build_config());
  s2k_ctx_destroy(ctx);
  return failures;
}

/* What a cgo BatchVerifier over several devices does: pinned packed arrays, two batches in flight. */
#ifdef WITH_ORACLE
#include "secp256k1_oracle.h"
#endif
static void make_batch(int n, int salt, uint8_t* pub, uint8_t* dig, uint8_t* r, uint8_t* s) {
  uint8_t d[32], k[32], Q[65], rid;
  for (int i = 0; i < n; ++i) {
    for (int j = 0; j < 32; ++j) {
      d[j] = (uint8_t)((i % 97) * 131 + j * 7 + 1 + salt);      /* 97 keys */
      k[j] = (uint8_t)(i * 17 + j * 23 + 9 + salt * 3);
      dig[i * 32 + j] = (uint8_t)(i * 5 + j * 3 + salt);
    }
    d[0] &= 0x7f;
    k[0] &= 0x7f;
    s2k_ct_scalar_base_mult(d, Q);
    memcpy(pub + i * 64, Q + 1, 64);
    if (s2k_ct_ecdsa_sign_raw(d, dig + i * 32, k, r + i * 32, s + i * 32, &rid) != S2K_OK) ++failures;
    if (i % 5 == salt % 5) dig[i * 32 + (i % 32)] ^= 0x10;      /* every fifth one damaged */
    if (i % 11 == 3) s[i * 32 + 31] ^= 1;
  }
}
static int group_part(void) {
  enum { N = 6000, B = 5 };   /* more batches than a context keeps in flight: the oldest is retired by a later submit */
  const int devices[2] = {0, 0};
  s2k_ctx* ctx = NULL;
  s2k_group* grp = NULL;
  CHECK(s2k_device_count() >= 1, "s2k_device_count");
  if (s2k_ctx_create(0, &ctx) != S2K_OK) {
    fprintf(stderr, "s2k_ctx_create: %s\n", s2k_last_error(NULL));
    return 1;
  }
  CHECK(s2k_group_create(devices, 2, &grp) == S2K_OK && s2k_group_size(grp) == 2, "s2k_group_create (two members on device 0)");
  if (!grp) return 1;
  uint8_t *pub[B], *dig[B], *r[B], *s[B], *sync[B], *piped[B], *grouped[B], *expect[B];
  for (int b = 0; b < B; ++b) {
    /* page-locked like a long-lived verifier's packed arrays; the verdict arrays are plain heap memory */
    pub[b] = s2k_host_alloc(N * 64);
    dig[b] = s2k_host_alloc(N * 32);
    r[b] = s2k_host_alloc(N * 32);
    s[b] = s2k_host_alloc(N * 32);
    sync[b] = malloc(N);
    piped[b] = malloc(N);
    grouped[b] = malloc(N);
    expect[b] = malloc(N);
    CHECK(pub[b] && dig[b] && r[b] && s[b], "s2k_host_alloc");
    make_batch(N - 100 * b, b, pub[b], dig[b], r[b], s[b]);
    memset(piped[b], 9, N);
    memset(grouped[b], 9, N);
    CHECK(s2k_ecdsa_verify_batch(ctx, N - 100 * b, pub[b], dig[b], r[b], s[b], S2K_ECDSA_REJECT_MALLEABLE, sync[b]) == S2K_OK, "synchronous call");
#ifdef WITH_ORACLE
    orc_ecdsa_verify_batch(N - 100 * b, pub[b], dig[b], r[b], s[b], 1, expect[b], 8);
    CHECK(memcmp(sync[b], expect[b], N - 100 * b) == 0, "synchronous call against the oracle");
#endif
  }
  /* one context, submit / wait: five batches, three in flight */
  s2k_ticket t[B];
  for (int b = 0; b < B; ++b)
    CHECK(s2k_ecdsa_verify_batch_submit(ctx, N - 100 * b, pub[b], dig[b], r[b], s[b], S2K_ECDSA_REJECT_MALLEABLE, piped[b], &t[b]) == S2K_OK, "submit");
  for (int b = 0; b < B; ++b) {
    CHECK(s2k_wait(ctx, t[b]) == S2K_OK, "wait");
    CHECK(memcmp(piped[b], sync[b], N - 100 * b) == 0, "submit / wait verdicts");
  }
  CHECK(s2k_wait(ctx, t[B - 1] + 1) == S2K_ERR_ARG, "unknown ticket refused");
  CHECK(s2k_poll(ctx, t[0]) == S2K_OK, "poll of a retired ticket");
  CHECK(s2k_wait_all(ctx) == S2K_OK, "wait_all with nothing in flight");
  /* the group: the same batches, all submitted before the first wait (the fourth submit blocks until the first is done) */
  s2k_ticket gt[B];
  for (int b = 0; b < B; ++b)
    CHECK(s2k_group_ecdsa_verify_batch_submit(grp, N - 100 * b, pub[b], dig[b], r[b], s[b], S2K_ECDSA_REJECT_MALLEABLE, grouped[b], &gt[b]) == S2K_OK,
          "group submit");
  int good = 0;
  for (int b = 0; b < B; ++b) {
    CHECK(s2k_group_wait(grp, gt[b]) == S2K_OK, s2k_group_last_error(grp));
    CHECK(memcmp(grouped[b], sync[b], N - 100 * b) == 0, "group verdicts against the single-context call");
#ifdef WITH_ORACLE
    CHECK(memcmp(grouped[b], expect[b], N - 100 * b) == 0, "group verdicts against the oracle");
#endif
    for (int i = 0; i < N - 100 * b; ++i) good += grouped[b][i];
  }
  double st[8];
  CHECK(s2k_group_member_stats(grp, st) == S2K_OK && st[0] + st[4] == N - 100 * (B - 1) && st[1] == 0 && st[5] == st[0], "member statistics");
  CHECK(s2k_group_ecdsa_verify_batch(grp, N, pub[0], dig[0], r[0], s[0], 0, grouped[0]) == S2K_OK, "synchronous group call");
  CHECK(good > 0 && good < B * N, "mixed verdicts");
  /* key sets: the 97 keys of batch 0 built once - on the context, and on every member of the group; the batch names them
   * by index and goes through the ticket forms */
  {
    const int n0 = N;
    uint8_t* keys = malloc(97 * 64);
    uint32_t* kidx = s2k_host_alloc(n0 * sizeof(uint32_t));
    uint8_t* ksv = malloc(n0);
    CHECK(keys && kidx && ksv, "allocation");
    memcpy(keys, pub[0], 97 * 64);                      /* (signature i was made with key i mod 97) */
    for (int i = 0; i < n0; ++i) kidx[i] = (uint32_t)(i % 97);
    s2k_keyset* ks = NULL;
    s2k_group_keyset* gks = NULL;
    CHECK(s2k_keyset_create_ex(ctx, 97, keys, S2K_KEYSET_JOINT, &ks) == S2K_OK && s2k_keyset_size(ks) == 97 &&
              s2k_keyset_layout(ks) == S2K_KEYSET_JOINT, "s2k_keyset_create_ex");
    CHECK(s2k_group_keyset_create(grp, 97, keys, S2K_KEYSET_JOINT, &gks) == S2K_OK && s2k_group_keyset_size(gks) == 97, "s2k_group_keyset_create");
    s2k_ticket kt = 0, kgt = 0;
    memset(ksv, 9, n0);
    CHECK(s2k_ecdsa_verify_batch_keyset_submit(ctx, ks, n0, kidx, dig[0], r[0], s[0], S2K_ECDSA_REJECT_MALLEABLE, ksv, &kt) == S2K_OK, "key-set submit");
    CHECK(s2k_wait(ctx, kt) == S2K_OK && memcmp(ksv, sync[0], n0) == 0, "key-set submit / wait verdicts");
    memset(ksv, 9, n0);
    CHECK(s2k_group_ecdsa_verify_batch_keyset_submit(grp, gks, n0, kidx, dig[0], r[0], s[0], S2K_ECDSA_REJECT_MALLEABLE, ksv, &kgt) == S2K_OK,
          "group key-set submit");
    CHECK(s2k_group_wait(grp, kgt) == S2K_OK && memcmp(ksv, sync[0], n0) == 0, "group key-set verdicts");
    kidx[5] = 97;                                       /* names no key: invalid */
    CHECK(s2k_group_ecdsa_verify_batch_keyset(grp, gks, n0, kidx, dig[0], r[0], s[0], S2K_ECDSA_REJECT_MALLEABLE, ksv) == S2K_OK && ksv[5] == 0 &&
              memcmp(ksv + 6, sync[0] + 6, n0 - 6) == 0, "index outside the set");
    CHECK(s2k_ecdsa_verify_batch_keyset_submit(ctx, NULL, n0, kidx, dig[0], r[0], s[0], 0, ksv, &kt) == S2K_ERR_ARG, "null key set refused");
    s2k_group_keyset_destroy(gks);
    s2k_keyset_destroy(ks);
    s2k_host_free(kidx);
    free(keys);
    free(ksv);
  }
  const int bad_dev[1] = {s2k_device_count()};
  s2k_group* none = NULL;
  CHECK(s2k_group_create(bad_dev, 1, &none) == S2K_ERR_ARG && none == NULL, "unknown device refused");
#ifdef WITH_ORACLE
  printf("group: %s (%d valid of %d, checked against the oracle)\n", failures ? "FAILED" : "ok", good, B * N - 100 * (B * (B - 1) / 2));
#else
  printf("group: %s (%d valid of %d)\n", failures ? "FAILED" : "ok", good, B * N - 100 * (B * (B - 1) / 2));
#endif
  s2k_group_destroy(grp);
  s2k_ctx_destroy(ctx);
  for (int b = 0; b < B; ++b) {
    s2k_host_free(pub[b]); s2k_host_free(dig[b]); s2k_host_free(r[b]); s2k_host_free(s[b]);
    free(sync[b]); free(piped[b]); free(grouped[b]); free(expect[b]);
  }
  return failures;
}

int main(int argc, char** argv) {
  if (argc > 1 && strcmp(argv[1], "gpu") == 0) return gpu_part() ? 1 : 0;
  if (argc > 1 && strcmp(argv[1], "group") == 0) return group_part() ? 1 : 0;
  return cpu_part() ? 1 : 0;
}

/*
 * secp256k1_oracle.h — CPU restatement of the Yawning/secp256k1-voi verify / scalar-mult
 * path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the HIP engine in secp256k1_voi_amd/.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path never links, imports or calls anything under oracle/.
 *
 * Parity status: PINNED — checked against the reference's own golden vectors
 * (Wycheproof ECDSA/ECDH, BIP-340 CSV, RFC 6979 CSV, libsecp256k1 KAT, GLV boundary
 * scalars, generator-table blob hash); see tests/test_oracle_golden.py.
 *
 * All file:line citations are relative to the reference tree (Yawning/secp256k1-voi).
 * Boundary convention (matches Scalar.Bytes / Element.Bytes, scalar.go:148, field.go:151):
 * every scalar / field element is a 32-byte big-endian canonical string; points are
 * 65-byte buffers: 0x04‖X‖Y, or 0x00 followed by 64 zero bytes for the identity.
 */
#ifndef SECP256K1_ORACLE_H
#define SECP256K1_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_POINT_SIZE 65

/* ---- Fp (internal/field/field.go) ---- */
int  orc_fp_is_canonical(const uint8_t a[32]);                      /* field.go:128 */
void orc_fp_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fp_sqr(uint8_t out[32], const uint8_t a[32]);
void orc_fp_add(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fp_sub(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fp_neg(uint8_t out[32], const uint8_t a[32]);
void orc_fp_inv(uint8_t out[32], const uint8_t a[32]);              /* field_invert.go:11 */
int  orc_fp_sqrt(uint8_t out[32], const uint8_t a[32]);             /* field_sqrt_ratio.go:14 */
/* inputs reduced mod p first (SetBytes semantics, field.go:115) */
void orc_fp_reduce(uint8_t out[32], const uint8_t a[32], int *did_reduce);

/* ---- Fn (scalar.go) ---- */
int  orc_fn_is_canonical(const uint8_t a[32]);                      /* scalar.go:136 */
void orc_fn_reduce(uint8_t out[32], const uint8_t a[32], int *did_reduce); /* scalar.go:123 */
void orc_fn_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fn_add(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fn_sub(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]);
void orc_fn_neg(uint8_t out[32], const uint8_t a[32]);
void orc_fn_inv(uint8_t out[32], const uint8_t a[32]);              /* scalar_invert.go:11 */
int  orc_fn_is_gt_half_n(const uint8_t a[32]);                      /* scalar.go:190 */
void orc_fn_split_glv(uint8_t k1[32], uint8_t k2[32], const uint8_t k[32]); /* point_mul_glv.go:59 */

/* ---- group (point.go, point_projective.go, point_mul_*.go, point_s11n.go) ---- */
void orc_point_generator(uint8_t out[65]);
void orc_point_identity(uint8_t out[65]);
/* decode any SEC1 encoding (1/33/65 bytes); 0 on success (point_s11n.go:215) */
int  orc_point_from_bytes(uint8_t out[65], const uint8_t *src, size_t len);
int  orc_point_on_curve_xy(const uint8_t x[32], const uint8_t y[32]); /* point_s11n.go:298 */
void orc_point_compressed(uint8_t out[33], size_t *out_len, const uint8_t p[65]); /* :90 */
void orc_point_add(uint8_t out[65], const uint8_t a[65], const uint8_t b[65]);    /* point.go:62 */
void orc_point_double(uint8_t out[65], const uint8_t a[65]);                     /* point.go:71 */
void orc_point_neg(uint8_t out[65], const uint8_t a[65]);                        /* point.go:89 */
/* same as orc_point_add but both inputs are first re-scaled by the projective factors
 * za, zb (canonical, non-zero) — mirrors DebugMustRandomizeZ, point_test.go:359 */
void orc_point_add_randz(uint8_t out[65], const uint8_t a[65], const uint8_t za[32],
                         const uint8_t b[65], const uint8_t zb[32]);
/* Point.Equal on re-scaled projective representatives (point.go:134) */
int  orc_point_equal_randz(const uint8_t a[65], const uint8_t za[32], const uint8_t b[65], const uint8_t zb[32]);
void orc_scalar_mult_vartime(uint8_t out[65], const uint8_t k[32], const uint8_t p[65]);   /* point_mul_glv.go:203 */
void orc_scalar_mult_vartime_randz(uint8_t out[65], const uint8_t k[32], const uint8_t p[65],
                                   const uint8_t z[32]);
void orc_scalar_mult_trivial(uint8_t out[65], const uint8_t k[32], const uint8_t p[65]);   /* point_test.go:392 */
void orc_scalar_base_mult_vartime(uint8_t out[65], const uint8_t k[32]);                   /* point_mul_table.go:197 */
void orc_double_scalar_mult_basepoint_vartime(uint8_t out[65], const uint8_t u1[32],
                                              const uint8_t u2[32], const uint8_t p[65]);  /* point_mul_glv.go:307 */
/* Straus; n==0 -> identity, n==1 -> GLV single (point_mul_multi.go:73) */
void orc_multi_scalar_mult_vartime(uint8_t out[65], size_t n, const uint8_t *scalars /* n*32 */,
                                   const uint8_t *points /* n*65 */);
/* generator table entry (j+1)*2^(8i)*G as X‖Y (internal/gentable/point_mul_table.go:20-51) */
void orc_generator_table_entry(uint8_t out[64], unsigned i, unsigned j);
/* SHA-256 over all 32*255 entries in blob order */
void orc_generator_table_sha256(uint8_t out[32]);

/* ---- ECDSA (secec/ecdsa.go, secec/s11n.go, secec/secec.go) ---- */
/* 1 = valid.  q: 64-byte affine X‖Y of an already-validated public key; digest >= 32
 * bytes (leftmost 32 used, ecdsa.go:477); r,s: 32-byte BE, must be canonical and in
 * [1,n) (ParseCompactSignature, s11n.go:129 + verify, ecdsa.go:392). */
int  orc_ecdsa_verify_raw(const uint8_t q[64], const uint8_t *digest, size_t digest_len,
                          const uint8_t r[32], const uint8_t s[32], int reject_malleable);
/* batch over packed arrays; out[i] in {0,1}; nthreads<=1 => serial */
void orc_ecdsa_verify_batch(size_t n, const uint8_t *q /* n*64 */, const uint8_t *digest32 /* n*32 */,
                            const uint8_t *r /* n*32 */, const uint8_t *s /* n*32 */,
                            int reject_malleable, uint8_t *out, int nthreads);
/* RecoverPublicKey (ecdsa.go:244): 1 and the 65-byte key on success, 0 on any error */
int  orc_ecdsa_recover(uint8_t out65[65], const uint8_t *digest, size_t digest_len, const uint8_t r[32],
                       const uint8_t s[32], unsigned recovery_id);
/* ParseASN1Signature (s11n.go:83): 0 ok, 1 = bad ASN.1, 2 = bad scalar */
int  orc_parse_asn1_signature(uint8_t r[32], uint8_t s[32], const uint8_t *der, size_t len);
/* PublicKey.Verify with EncodingASN1 and opts==nil|{RejectMalleable} (ecdsa.go:171);
 * pub is any SEC1 encoding accepted by secec.NewPublicKey (secec.go:188). -1 = bad key */
int  orc_ecdsa_verify_asn1(const uint8_t *pub, size_t pub_len, const uint8_t *digest, size_t digest_len,
                           const uint8_t *sig, size_t sig_len, int reject_malleable);

/* ---- BIP-340 (secec/bitcoin/schnorr.go) ---- */
/* 1 valid, 0 invalid, -1 invalid public key (NewSchnorrPublicKey fails, schnorr.go:257) */
int  orc_schnorr_verify(const uint8_t pk[32], const uint8_t *msg, size_t msg_len,
                        const uint8_t *sig, size_t sig_len);
void orc_sha256(uint8_t out[32], const uint8_t *data, size_t len);

/* instrumentation: number of Fp / Fn Montgomery multiplications since the last reset */
void orc_counters_reset(void);
void orc_counters_get(uint64_t *fp_mul, uint64_t *fn_mul);

#ifdef __cplusplus
}
#endif
#endif

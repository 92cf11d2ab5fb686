/*
 * secp256k1_oracle.c — CPU restatement of the reference's verify / scalar-mult path.
 *
 * TEST INFRASTRUCTURE ONLY (see secp256k1_oracle.h).  Plain C, 4x64-bit saturated limbs
 * in the Montgomery domain, the same algorithms, window sizes, formulas and addition
 * chains as the reference, so that (a) it is a faithful CPU baseline and (b) every
 * canonical output (valid bit, canonical bytes) equals the reference's.  Nothing here is
 * copied from the reference: the fiat-generated straight-line code is restated as the
 * textbook word-by-word Montgomery loop it was generated from.
 *
 * Citations are file:line in the reference tree.
 */
#include "secp256k1_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t v[4]; } u256;          /* limb 0 least significant (internal/helpers/helpers.go:48-55) */

typedef struct {
    u256 m;          /* modulus */
    uint64_t minv;   /* -m^-1 mod 2^64 */
    u256 one;        /* R mod m */
    u256 r2;         /* R^2 mod m */
} mont_ctx;

static __thread uint64_t cnt_fp_mul, cnt_fn_mul;

/* p, m' : internal/fiat/secp256k1montgomery/secp256k1montgomery.go:87-406 (constants 0xd838091dd2253531, 0xfffffffefffffc2f) */
static mont_ctx FP = {
    {{0xfffffffefffffc2fULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL, 0xffffffffffffffffULL}},
    0xd838091dd2253531ULL, {{0}}, {{0}}};
/* n, m' : internal/fiat/secp256k1montgomeryscalar/secp256k1montgomeryscalar.go:11,87 */
static mont_ctx FN = {
    {{0xbfd25e8cd0364141ULL, 0xbaaedce6af48a03bULL, 0xfffffffffffffffeULL, 0xffffffffffffffffULL}},
    0x4b0dff665588b13fULL, {{0}}, {{0}}};

/* ------------------------------------------------------------------ */
/* 256-bit helpers                                                     */
/* ------------------------------------------------------------------ */

static inline uint64_t adc(uint64_t a, uint64_t b, uint64_t *carry) {
    u128 t = (u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t sbb(uint64_t a, uint64_t b, uint64_t *borrow) {
    u128 t = (u128)a - b - *borrow;
    *borrow = (uint64_t)(t >> 64) & 1;
    return (uint64_t)t;
}
static uint64_t u256_add(u256 *o, const u256 *a, const u256 *b) {
    uint64_t c = 0;
    for (int i = 0; i < 4; i++) o->v[i] = adc(a->v[i], b->v[i], &c);
    return c;
}
static uint64_t u256_sub(u256 *o, const u256 *a, const u256 *b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) o->v[i] = sbb(a->v[i], b->v[i], &br);
    return br;
}
static int u256_is_zero(const u256 *a) { return (a->v[0] | a->v[1] | a->v[2] | a->v[3]) == 0; }
static int u256_eq(const u256 *a, const u256 *b) {
    return ((a->v[0] ^ b->v[0]) | (a->v[1] ^ b->v[1]) | (a->v[2] ^ b->v[2]) | (a->v[3] ^ b->v[3])) == 0;
}
/* BytesToSaturated / PutSaturatedToBytes, internal/helpers/helpers.go:48-66 */
static void u256_from_be(u256 *o, const uint8_t b[32]) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int j = 0; j < 8; j++) w = (w << 8) | b[(3 - i) * 8 + j];
        o->v[i] = w;
    }
}
static void u256_to_be(uint8_t b[32], const u256 *a) {
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) b[(3 - i) * 8 + j] = (uint8_t)(a->v[i] >> (56 - 8 * j));
}
/* reduceSaturated: dst = src - m if src >= m (src < 2m); returns didReduce.
 * internal/field/field_reduce.go:82-102 and scalar.go (same routine for n). */
static int reduce_saturated(u256 *dst, const u256 *src, const mont_ctx *c) {
    u256 red;
    uint64_t borrow = u256_sub(&red, src, &c->m);
    int did = (borrow == 0);
    *dst = did ? red : *src;
    return did;
}

/* ------------------------------------------------------------------ */
/* Montgomery arithmetic (fiat word-by-word, R = 2^256)                */
/* ------------------------------------------------------------------ */

/* Mul: secp256k1montgomery.go:87-406 / secp256k1montgomeryscalar.go:87.  Four rounds of
 * (t += a_i*b ; q = t0*m' ; t = (t + q*m) / 2^64) and one final conditional subtract. */
static void mont_mul(u256 *out, const u256 *a, const u256 *b, const mont_ctx *c) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < 4; j++) {
            u128 x = (u128)a->v[i] * b->v[j] + t[j] + carry;
            t[j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        u128 x = (u128)t[4] + carry;
        t[4] = (uint64_t)x;
        t[5] = (uint64_t)(x >> 64);

        uint64_t q = t[0] * c->minv;
        x = (u128)q * c->m.v[0] + t[0];
        carry = (uint64_t)(x >> 64);
        for (int j = 1; j < 4; j++) {
            x = (u128)q * c->m.v[j] + t[j] + carry;
            t[j - 1] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        x = (u128)t[4] + carry;
        t[3] = (uint64_t)x;
        t[4] = t[5] + (uint64_t)(x >> 64);
        t[5] = 0;
    }
    u256 r = {{t[0], t[1], t[2], t[3]}}, red;
    uint64_t borrow = u256_sub(&red, &r, &c->m);
    /* subtract iff t >= m, i.e. overflow word set or no borrow */
    *out = (t[4] != 0 || borrow == 0) ? red : r;
}
/* Add/Sub/Opp: secp256k1montgomery.go:750,802,844 */
static void mont_add(u256 *o, const u256 *a, const u256 *b, const mont_ctx *c) {
    u256 s, red;
    uint64_t carry = u256_add(&s, a, b);
    uint64_t borrow = u256_sub(&red, &s, &c->m);
    *o = (carry != 0 || borrow == 0) ? red : s;
}
static void mont_sub(u256 *o, const u256 *a, const u256 *b, const mont_ctx *c) {
    u256 d;
    uint64_t borrow = u256_sub(&d, a, b);
    if (borrow) u256_add(&d, &d, &c->m);
    *o = d;
}
static void mont_neg(u256 *o, const u256 *a, const mont_ctx *c) {
    u256 z = {{0, 0, 0, 0}};
    mont_sub(o, &z, a, c);
}
/* ToMontgomery (x * R^2 * R^-1) / FromMontgomery (x * 1 * R^-1): secp256k1montgomery.go:1110,886 */
static void mont_to(u256 *o, const u256 *a, const mont_ctx *c) { mont_mul(o, a, &c->r2, c); }
static void mont_from(u256 *o, const u256 *a, const mont_ctx *c) {
    u256 one = {{1, 0, 0, 0}};
    mont_mul(o, a, &one, c);
}

static void mont_ctx_init(mont_ctx *c) {
    /* R mod m = 2^256 - m (m > 2^255) — SetOne, secp256k1montgomery.go:1606 */
    u256 z = {{0, 0, 0, 0}};
    u256_sub(&c->one, &z, &c->m);
    /* R^2 mod m by 256 modular doublings of R mod m */
    u256 r2 = c->one;
    for (int i = 0; i < 256; i++) mont_add(&r2, &r2, &r2, c);
    c->r2 = r2;
}

/* ------------------------------------------------------------------ */
/* field.Element (Fp) and Scalar (Fn) wrappers                         */
/* ------------------------------------------------------------------ */
typedef u256 fe; /* Montgomery domain, mod p */
typedef u256 sc; /* Montgomery domain, mod n */

static inline void fe_mul(fe *o, const fe *a, const fe *b) { cnt_fp_mul++; mont_mul(o, a, b, &FP); }   /* field.go:82 */
static inline void fe_sqr(fe *o, const fe *a) { cnt_fp_mul++; mont_mul(o, a, a, &FP); }                /* field.go:90; fiat Square has the Mul schedule */
static inline void fe_add(fe *o, const fe *a, const fe *b) { mont_add(o, a, b, &FP); }                 /* field.go:61 */
static inline void fe_sub(fe *o, const fe *a, const fe *b) { mont_sub(o, a, b, &FP); }                 /* field.go:68 */
static inline void fe_neg(fe *o, const fe *a) { mont_neg(o, a, &FP); }                                 /* field.go:75 */
static void fe_pow2k(fe *o, const fe *a, unsigned k) {                                                /* field.go:97 */
    fe t = *a;
    for (unsigned i = 0; i < k; i++) fe_sqr(&t, &t);
    *o = t;
}
static inline int fe_is_zero(const fe *a) { return u256_is_zero(a); }                                  /* field.go:183 */
static inline int fe_eq(const fe *a, const fe *b) { return u256_eq(a, b); }                            /* field.go:178 */
static int fe_is_odd(const fe *a) { u256 nm; mont_from(&nm, a, &FP); return (int)(nm.v[0] & 1); }      /* field.go:191 */
/* SetCanonicalBytes, field.go:128 — 0 on success */
static int fe_set_canonical_bytes(fe *o, const uint8_t b[32]) {
    u256 l, r;
    u256_from_be(&l, b);
    if (reduce_saturated(&r, &l, &FP)) return -1;
    mont_to(o, &l, &FP);
    return 0;
}
static void fe_get_bytes(uint8_t b[32], const fe *a) { u256 nm; mont_from(&nm, a, &FP); u256_to_be(b, &nm); } /* field.go:156 */
static void fe_from_u64(fe *o, uint64_t x) { u256 l = {{x, 0, 0, 0}}; mont_to(o, &l, &FP); }

static fe FE_ONE, FE_ZERO, FE_B, FE_B3, FE_BETA, FE_C2, FE_GX, FE_GY;

/* Invert: x^(p-2), addition chain of internal/field/field_invert.go:11-140 (255 S + 15 M) */
static void fe_invert(fe *z, const fe *x) {
    fe t0, t1, t2, t3, t4, t5;
    fe_sqr(&t0, x);            /* x^2 */
    fe_sqr(&t1, &t0);          /* x^4 */
    fe_mul(&t1, x, &t1);       /* _101 */
    fe_mul(&t0, &t0, &t1);     /* _111 */
    fe_sqr(&t2, &t0);          /* _1110 */
    fe_pow2k(&t3, &t2, 2);     /* _111000 */
    fe_mul(&t3, &t0, &t3);     /* _111111 */
    fe_pow2k(&t3, &t3, 4);
    fe_mul(&t2, &t2, &t3);     /* i13 */
    fe_pow2k(&t3, &t2, 2);
    fe_mul(&t3, &t0, &t3);     /* x12 */
    fe_pow2k(&t3, &t3, 10);
    fe_mul(&t2, &t2, &t3);
    fe_mul(&t4, x, &t2);       /* x22 */
    fe_sqr(&t2, &t4);          /* i29 */
    fe_pow2k(&t3, &t2, 2);     /* i31 */
    fe_pow2k(&t5, &t3, 22);
    fe_mul(&t3, &t3, &t5);     /* i54 */
    fe_pow2k(&t5, &t3, 20);
    fe_mul(&t2, &t2, &t5);
    fe_pow2k(&t2, &t2, 46);
    fe_mul(&t3, &t3, &t2);     /* i122 */
    fe_pow2k(&t2, &t3, 110);
    fe_mul(&t3, &t3, &t2);
    fe_mul(&t0, &t0, &t3);     /* x223 */
    fe_pow2k(&t0, &t0, 23);
    fe_mul(&t4, &t4, &t0);
    fe_pow2k(&t4, &t4, 7);
    fe_mul(&t4, &t1, &t4);
    fe_pow2k(&t4, &t4, 3);     /* i269 */
    fe_mul(z, &t1, &t4);
}

/* pow3mod4: x^((p-3)/4), chain of internal/field/field_sqrt_ratio.go:65-185 (253 S + 14 M) */
static void fe_pow3mod4(fe *z, const fe *x) {
    fe t0, t1, t2, t3, t4, t5;
    fe_sqr(&t0, x);
    fe_mul(&t0, x, &t0);       /* _11 */
    fe_pow2k(&t1, &t0, 2);
    fe_mul(&t1, &t0, &t1);     /* _1111 */
    fe_sqr(&t2, &t1);
    fe_mul(&t2, x, &t2);       /* _11111 */
    fe_pow2k(&t3, &t2, 2);
    fe_mul(&t3, &t0, &t3);     /* _1111111 */
    fe_pow2k(&t4, &t3, 4);
    fe_mul(&t1, &t1, &t4);     /* x11 */
    fe_pow2k(&t4, &t1, 11);
    fe_mul(&t1, &t1, &t4);     /* x22 */
    fe_pow2k(&t4, &t1, 5);
    fe_mul(&t2, &t2, &t4);     /* x27 */
    fe_pow2k(&t4, &t2, 27);
    fe_mul(&t2, &t2, &t4);     /* x54 */
    fe_pow2k(&t4, &t2, 54);
    fe_mul(&t2, &t2, &t4);     /* x108 */
    fe_pow2k(&t4, &t2, 108);
    fe_mul(&t2, &t2, &t4);     /* x216 */
    fe_pow2k(&t2, &t2, 7);
    fe_mul(&t3, &t3, &t2);     /* x223 */
    fe_pow2k(&t3, &t3, 23);
    fe_mul(&t1, &t1, &t3);
    fe_pow2k(&t1, &t1, 5);
    fe_mul(&t5, x, &t1);
    fe_pow2k(&t5, &t5, 3);     /* i266 */
    fe_mul(z, &t0, &t5);
}

/* SqrtRatio (RFC 9380 F.2.1.2, q = 3 mod 4): internal/field/field_sqrt_ratio.go:25-63 */
static int fe_sqrt_ratio(fe *z, const fe *u, const fe *v) {
    fe tv1, tv2, tv3, y1, y2;
    fe_sqr(&tv1, v);
    fe_mul(&tv2, u, v);
    fe_mul(&tv1, &tv1, &tv2);
    fe_pow3mod4(&y1, &tv1);
    fe_mul(&y1, &y1, &tv2);
    fe_mul(&y2, &y1, &FE_C2);
    fe_sqr(&tv3, &y1);
    fe_mul(&tv3, &tv3, v);
    int is_qr = fe_eq(&tv3, u);
    *z = is_qr ? y1 : y2;
    return is_qr;
}
/* Sqrt: field_sqrt_ratio.go:14-23 — fe = 0 when no root exists */
static int fe_sqrt(fe *o, const fe *a) {
    fe tmp;
    int ok = fe_sqrt_ratio(&tmp, a, &FE_ONE);
    *o = ok ? tmp : FE_ZERO;
    return ok;
}

static inline void sc_mul(sc *o, const sc *a, const sc *b) { cnt_fn_mul++; mont_mul(o, a, b, &FN); }   /* scalar.go:84 */
static inline void sc_sqr(sc *o, const sc *a) { cnt_fn_mul++; mont_mul(o, a, a, &FN); }
static inline void sc_add(sc *o, const sc *a, const sc *b) { mont_add(o, a, b, &FN); }                 /* scalar.go:66 */
static inline void sc_sub(sc *o, const sc *a, const sc *b) { mont_sub(o, a, b, &FN); }
static inline void sc_neg(sc *o, const sc *a) { mont_neg(o, a, &FN); }                                 /* scalar.go:78 */
static void sc_pow2k(sc *o, const sc *a, unsigned k) {
    sc t = *a;
    for (unsigned i = 0; i < k; i++) sc_sqr(&t, &t);
    *o = t;
}
static inline int sc_is_zero(const sc *a) { return u256_is_zero(a); }                                  /* scalar.go:181 */
static inline int sc_eq(const sc *a, const sc *b) { return u256_eq(a, b); }                            /* scalar.go:176 */
/* SetBytes (reduces; returns didReduce), scalar.go:123 */
static int sc_set_bytes(sc *o, const uint8_t b[32]) {
    u256 l;
    u256_from_be(&l, b);
    int did = reduce_saturated(&l, &l, &FN);
    mont_to(o, &l, &FN);
    return did;
}
/* SetCanonicalBytes, scalar.go:136 — 0 on success */
static int sc_set_canonical_bytes(sc *o, const uint8_t b[32]) {
    u256 l, r;
    u256_from_be(&l, b);
    if (reduce_saturated(&r, &l, &FN)) return -1;
    mont_to(o, &l, &FN);
    return 0;
}
static void sc_get_bytes(uint8_t b[32], const sc *a) { u256 nm; mont_from(&nm, a, &FN); u256_to_be(b, &nm); } /* scalar.go:153 */
/* IsGreaterThanHalfN, scalar.go:190-206 (halfNSat, scalar.go:33-38) */
static int sc_is_gt_half_n(const sc *a) {
    static const u256 half_n = {{0xdfe92f46681b20a0ULL, 0x5d576e7357a4501dULL, 0xffffffffffffffffULL, 0x7fffffffffffffffULL}};
    u256 nm, d;
    mont_from(&nm, a, &FN);
    uint64_t borrow = u256_sub(&d, &nm, &half_n);
    return borrow == 0 && !u256_is_zero(&d);
}

/* Invert: x^(n-2), addition chain of scalar_invert.go:11-303 (253 S + 40 M) */
static void sc_invert(sc *z, const sc *x) {
    sc t0, t1, t2, t3, t4, t5, t6, t7, t8, t9, t10, t11, t12, t13, t14;
    sc_sqr(&t0, x);             /* _10 */
    sc_mul(&t1, x, &t0);        /* _11 */
    sc_mul(&t2, &t0, &t1);      /* _101 */
    sc_mul(&t3, &t0, &t2);      /* _111 */
    sc_mul(&t4, &t0, &t3);      /* _1001 */
    sc_mul(&t5, &t0, &t4);      /* _1011 */
    sc_mul(&t0, &t0, &t5);      /* _1101 */
    sc_pow2k(&t6, &t0, 2);
    sc_mul(&t6, &t5, &t6);      /* _111111 */
    sc_sqr(&t7, &t6);
    sc_mul(&t7, x, &t7);        /* _1111111 */
    sc_sqr(&t8, &t7);
    sc_mul(&t8, x, &t8);        /* _11111111 */
    sc_pow2k(&t9, &t8, 3);      /* i17 */
    sc_pow2k(&t10, &t9, 2);     /* i19 */
    sc_sqr(&t11, &t10);         /* i20 */
    sc_sqr(&t12, &t11);         /* i21 */
    sc_pow2k(&t13, &t12, 7);
    sc_mul(&t11, &t11, &t13);
    sc_pow2k(&t11, &t11, 9);
    sc_mul(&t12, &t12, &t11);   /* i39 */
    sc_pow2k(&t11, &t12, 6);
    sc_mul(&t10, &t10, &t11);
    sc_pow2k(&t10, &t10, 26);
    sc_mul(&t12, &t12, &t10);   /* i73 */
    sc_pow2k(&t10, &t12, 4);
    sc_mul(&t9, &t9, &t10);
    sc_pow2k(&t9, &t9, 60);
    sc_mul(&t12, &t12, &t9);
    sc_mul(&t7, &t7, &t12);     /* x127 */
    sc_pow2k(&t7, &t7, 5);
    sc_mul(&t7, &t5, &t7);
    sc_pow2k(&t7, &t7, 3);
    sc_mul(&t7, &t2, &t7);
    sc_pow2k(&t7, &t7, 4);      /* i154 */
    sc_mul(&t7, &t2, &t7);
    sc_pow2k(&t7, &t7, 4);
    sc_mul(&t7, &t3, &t7);
    sc_pow2k(&t7, &t7, 5);
    sc_mul(&t7, &t0, &t7);      /* i166 */
    sc_pow2k(&t7, &t7, 2);
    sc_mul(&t7, &t1, &t7);
    sc_pow2k(&t7, &t7, 5);
    sc_mul(&t7, &t3, &t7);
    sc_pow2k(&t7, &t7, 6);      /* i181 */
    sc_mul(&t7, &t0, &t7);
    sc_pow2k(&t7, &t7, 5);
    sc_mul(&t7, &t5, &t7);
    sc_pow2k(&t7, &t7, 4);
    sc_mul(&t7, &t0, &t7);      /* i193 */
    sc_pow2k(&t7, &t7, 3);
    sc_mul(&t7, x, &t7);
    sc_pow2k(&t7, &t7, 6);
    sc_mul(&t2, &t2, &t7);
    sc_pow2k(&t2, &t2, 10);     /* i214 */
    sc_mul(&t2, &t3, &t2);
    sc_pow2k(&t2, &t2, 4);
    sc_mul(&t3, &t3, &t2);
    sc_pow2k(&t3, &t3, 9);
    sc_mul(&t8, &t8, &t3);      /* i230 */
    sc_pow2k(&t8, &t8, 5);
    sc_mul(&t8, &t4, &t8);
    sc_pow2k(&t8, &t8, 6);
    sc_mul(&t5, &t5, &t8);
    sc_pow2k(&t5, &t5, 4);      /* i247 */
    sc_mul(&t5, &t0, &t5);
    sc_pow2k(&t5, &t5, 5);
    sc_mul(&t1, &t1, &t5);
    sc_pow2k(&t1, &t1, 6);
    sc_mul(&t1, &t0, &t1);      /* i261 */
    sc_pow2k(&t1, &t1, 10);
    sc_mul(&t0, &t0, &t1);
    sc_pow2k(&t0, &t0, 4);
    sc_mul(&t4, &t4, &t0);
    sc_pow2k(&t4, &t4, 6);      /* i283 */
    sc_mul(&t14, x, &t4);
    sc_pow2k(&t14, &t14, 8);
    sc_mul(z, &t6, &t14);
}

/* ------------------------------------------------------------------ */
/* Point (point.go:31): homogeneous projective, identity (0:1:0)       */
/* ------------------------------------------------------------------ */
typedef struct { fe x, y, z; } pt;
typedef struct { fe x, y; } apt;   /* affinePoint, point_mul_table.go:67 */

static void pt_identity(pt *v) { v->x = FE_ZERO; v->y = FE_ONE; v->z = FE_ZERO; }     /* point.go:42 */
static void pt_generator(pt *v) { v->x = FE_GX; v->y = FE_GY; v->z = FE_ONE; }        /* point.go:52, feGX/feGY point.go:18-21 */
static int  pt_is_identity(const pt *v) { return fe_is_zero(&v->z); }                 /* point.go:148 */
static void pt_neg(pt *v, const pt *p) { v->x = p->x; fe_neg(&v->y, &p->y); v->z = p->z; } /* point.go:89 */

/* addComplete — RCB'15 Algorithm 7, a = 0 (point_projective.go:24-120): 12 M + 2 m3b */
static void pt_add_complete(pt *v, const pt *p, const pt *q) {
    fe t0, t1, t2, t3, t4, x3, y3, z3;
    fe_mul(&t0, &p->x, &q->x);
    fe_mul(&t1, &p->y, &q->y);
    fe_mul(&t2, &p->z, &q->z);
    fe_add(&t3, &p->x, &p->y);
    fe_add(&t4, &q->x, &q->y);
    fe_mul(&t3, &t3, &t4);
    fe_add(&t4, &t0, &t1);
    fe_sub(&t3, &t3, &t4);
    fe_add(&t4, &p->y, &p->z);
    fe_add(&x3, &q->y, &q->z);
    fe_mul(&t4, &t4, &x3);
    fe_add(&x3, &t1, &t2);
    fe_sub(&t4, &t4, &x3);
    fe_add(&x3, &p->x, &p->z);
    fe_add(&y3, &q->x, &q->z);
    fe_mul(&x3, &x3, &y3);
    fe_add(&y3, &t0, &t2);
    fe_sub(&y3, &x3, &y3);
    fe_add(&x3, &t0, &t0);
    fe_add(&t0, &x3, &t0);
    fe_mul(&t2, &FE_B3, &t2);
    fe_add(&z3, &t1, &t2);
    fe_sub(&t1, &t1, &t2);
    fe_mul(&y3, &FE_B3, &y3);
    fe_mul(&x3, &t4, &y3);
    fe_mul(&t2, &t3, &t1);
    fe_sub(&x3, &t2, &x3);
    fe_mul(&y3, &y3, &t0);
    fe_mul(&t1, &t1, &z3);
    fe_add(&y3, &t1, &y3);
    fe_mul(&t0, &t0, &t3);
    fe_mul(&z3, &z3, &t4);
    fe_add(&z3, &z3, &t0);
    v->x = x3; v->y = y3; v->z = z3;
}
/* addMixed — RCB'15 Algorithm 8 (point_projective.go:123-205): 11 M; addend must not be the identity */
static void pt_add_mixed(pt *v, const pt *p, const fe *x2, const fe *y2) {
    fe t0, t1, t2, t3, t4, x3, y3, z3;
    fe_mul(&t0, &p->x, x2);
    fe_mul(&t1, &p->y, y2);
    fe_add(&t3, x2, y2);
    fe_add(&t4, &p->x, &p->y);
    fe_mul(&t3, &t3, &t4);
    fe_add(&t4, &t0, &t1);
    fe_sub(&t3, &t3, &t4);
    fe_mul(&t4, y2, &p->z);
    fe_add(&t4, &t4, &p->y);
    fe_mul(&y3, x2, &p->z);
    fe_add(&y3, &y3, &p->x);
    fe_add(&x3, &t0, &t0);
    fe_add(&t0, &x3, &t0);
    fe_mul(&t2, &FE_B3, &p->z);
    fe_add(&z3, &t1, &t2);
    fe_sub(&t1, &t1, &t2);
    fe_mul(&y3, &FE_B3, &y3);
    fe_mul(&x3, &t4, &y3);
    fe_mul(&t2, &t3, &t1);
    fe_sub(&x3, &t2, &x3);
    fe_mul(&y3, &y3, &t0);
    fe_mul(&t1, &t1, &z3);
    fe_add(&y3, &t1, &y3);
    fe_mul(&t0, &t0, &t3);
    fe_mul(&z3, &z3, &t4);
    fe_add(&z3, &z3, &t0);
    v->x = x3; v->y = y3; v->z = z3;
}
/* doubleComplete — RCB'15 Algorithm 9 (point_projective.go:208-273): 6 M + 2 S + 1 m3b */
static void pt_double_complete(pt *v, const pt *p) {
    fe t0, t1, t2, x3, y3, z3;
    fe_sqr(&t0, &p->y);
    fe_add(&z3, &t0, &t0);
    fe_add(&z3, &z3, &z3);
    fe_add(&z3, &z3, &z3);
    fe_mul(&t1, &p->y, &p->z);
    fe_sqr(&t2, &p->z);
    fe_mul(&t2, &FE_B3, &t2);
    fe_mul(&x3, &t2, &z3);
    fe_add(&y3, &t0, &t2);
    fe_mul(&z3, &t1, &z3);
    fe_add(&t1, &t2, &t2);
    fe_add(&t2, &t1, &t2);
    fe_sub(&t0, &t0, &t2);
    fe_mul(&y3, &t0, &y3);
    fe_add(&y3, &x3, &y3);
    fe_mul(&t1, &p->x, &p->y);
    fe_mul(&x3, &t0, &t1);
    fe_add(&x3, &x3, &x3);
    v->x = x3; v->y = y3; v->z = z3;
}
/* rescale — Z = 1, identity-safe (point_projective.go:278-302) */
static void pt_rescale(pt *v, const pt *p) {
    if (pt_is_identity(p)) { pt_identity(v); return; }
    fe a;
    fe_invert(&a, &p->z);
    fe_mul(&v->x, &a, &p->x);
    fe_mul(&v->y, &a, &p->y);
    v->z = FE_ONE;
}
/* multiply all coordinates by z (DebugMustRandomizeZ, point_test.go:359-373) */
static void pt_scale_z(pt *v, const pt *p, const fe *z) {
    fe_mul(&v->x, &p->x, z);
    fe_mul(&v->y, &p->y, z);
    fe_mul(&v->z, &p->z, z);
}
/* Equal: cross-multiplied comparison (point.go:134-145) */
static int pt_equal(const pt *a, const pt *b) {
    fe x1z2, x2z1, y1z2, y2z1;
    fe_mul(&x1z2, &a->x, &b->z);
    fe_mul(&x2z1, &b->x, &a->z);
    fe_mul(&y1z2, &a->y, &b->z);
    fe_mul(&y2z1, &b->y, &a->z);
    return fe_eq(&x1z2, &x2z1) && fe_eq(&y1z2, &y2z1);
}

/* maybeYY = x^3 + 7 ; xyOnCurve (point_s11n.go:298-307) */
static void fe_maybe_yy(fe *yy, const fe *x) {
    fe_sqr(yy, x);
    fe_mul(yy, yy, x);
    fe_add(yy, yy, &FE_B);
}
static int xy_on_curve(const fe *x, const fe *y) {
    fe yy, y2;
    fe_maybe_yy(&yy, x);
    fe_sqr(&y2, y);
    return fe_eq(&yy, &y2);
}

/* --- 65-byte boundary encoding --- */
static void pt_to_buf(uint8_t out[65], const pt *p) {
    memset(out, 0, 65);
    if (pt_is_identity(p)) return;                 /* prefixIdentity, point_s11n.go:71-73 */
    pt s;
    pt_rescale(&s, p);
    out[0] = 0x04;
    fe_get_bytes(out + 1, &s.x);
    fe_get_bytes(out + 33, &s.y);
}
static int pt_from_buf(pt *p, const uint8_t in[65]) {
    if (in[0] == 0x00) { pt_identity(p); return 0; }
    if (in[0] != 0x04) return -1;
    if (fe_set_canonical_bytes(&p->x, in + 1) || fe_set_canonical_bytes(&p->y, in + 33)) return -1;
    if (!xy_on_curve(&p->x, &p->y)) return -1;
    p->z = FE_ONE;
    return 0;
}

/* SetCompressedBytes (point_s11n.go:140-172), SetUncompressedBytes (:178-209), SetBytes (:215-230) */
static int pt_set_bytes(pt *v, const uint8_t *src, size_t len) {
    if (len == 1) {
        if (src[0] != 0x00) return -1;
        pt_identity(v);
        return 0;
    }
    if (len == 33) {
        if (src[0] != 0x02 && src[0] != 0x03) return -1;
        fe x, yy, y, yneg;
        if (fe_set_canonical_bytes(&x, src + 1)) return -1;
        fe_maybe_yy(&yy, &x);
        if (!fe_sqrt(&y, &yy)) return -1;
        fe_neg(&yneg, &y);
        int tag_eq = (fe_is_odd(&y) == (src[0] & 1));
        v->x = x;
        v->y = tag_eq ? y : yneg;
        v->z = FE_ONE;
        return 0;
    }
    if (len == 65) {
        if (src[0] != 0x04) return -1;
        fe x, y;
        if (fe_set_canonical_bytes(&x, src + 1)) return -1;
        if (fe_set_canonical_bytes(&y, src + 33)) return -1;
        if (!xy_on_curve(&x, &y)) return -1;
        v->x = x; v->y = y; v->z = FE_ONE;
        return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------ */
/* scalar multiplication                                               */
/* ------------------------------------------------------------------ */

/* projectivePointMultTable [1P..15P] (point_mul_table.go:30,51-60) */
typedef struct { pt e[15]; } proj_tbl;
static void proj_tbl_init(proj_tbl *t, const pt *p) {
    t->e[0] = *p;
    for (int i = 1; i < 15; i += 2) {
        pt_double_complete(&t->e[i], &t->e[i / 2]);
        pt_add_complete(&t->e[i + 1], &t->e[i], p);
    }
}
/* SelectAndAddVartime (point_mul_table.go:43-49) */
static void proj_tbl_add_vartime(const proj_tbl *t, pt *sum, unsigned idx) {
    if (idx == 0) return;
    pt_add_complete(sum, sum, &t->e[idx - 1]);
}

/* generatorHugeAffineTable: tbl[i][j] = (j+1) * 2^(8i) * G (point_mul_table.go:73-100,
 * internal/gentable/point_mul_table.go:20-51).  Regenerated here, not read from the blob. */
static apt G_TABLE[32][255];

static void build_generator_table(void) {
    static pt row[255];
    pt base;
    pt_generator(&base);
    for (int i = 0; i < 32; i++) {
        row[0] = base;
        for (int j = 1; j < 255; j++) pt_add_complete(&row[j], &row[j - 1], &base);
        /* affine via one shared inversion (Montgomery's trick) */
        static fe prod[255];
        prod[0] = row[0].z;
        for (int j = 1; j < 255; j++) fe_mul(&prod[j], &prod[j - 1], &row[j].z);
        fe inv, zi;
        fe_invert(&inv, &prod[254]);
        for (int j = 254; j >= 0; j--) {
            if (j > 0) { fe_mul(&zi, &inv, &prod[j - 1]); fe_mul(&inv, &inv, &row[j].z); }
            else zi = inv;
            fe_mul(&G_TABLE[i][j].x, &row[j].x, &zi);
            fe_mul(&G_TABLE[i][j].y, &row[j].y, &zi);
        }
        /* next base = 256 * base */
        pt nb;
        pt_add_complete(&nb, &row[254], &base);
        base = nb;
    }
}

/* scalarBaseMultVartime (point_mul_table.go:197-211): 32 byte-indexed mixed adds, no doublings */
static void pt_scalar_base_mult_vartime(pt *v, const sc *s) {
    uint8_t b[32];
    sc_get_bytes(b, s);
    pt_identity(v);
    for (int i = 0; i < 32; i++) {
        unsigned idx = b[i];
        if (idx == 0) continue;                                   /* :105-107 */
        const apt *a = &G_TABLE[31 - i][idx - 1];
        pt_add_mixed(v, v, &a->x, &a->y);
    }
}

/* GLV constants (point_mul_glv.go:37-57) */
static sc SC_NEG_LAMBDA, SC_NEG_B1, SC_NEG_B2, SC_G1, SC_G2, SC_ONE, SC_ZERO;

/* mulGFlooredDiv (point_mul_glv.go:119-189): floor(k*g / 2^384) rounded on bit 383 */
static void sc_mul_g_floored_div(sc *o, const sc *k, const sc *g) {
    u256 a, b;
    mont_from(&a, k, &FN);
    mont_from(&b, g, &FN);
    uint64_t c[8] = {0};
    for (int i = 0; i < 4; i++) {
        uint64_t u = 0;
        for (int j = 0; j < 4; j++) {
            u128 t = (u128)a.v[i] * b.v[j] + c[i + j] + u;
            c[i + j] = (uint64_t)t;
            u = (uint64_t)(t >> 64);
        }
        c[i + 4] = u;
    }
    uint64_t should_add = (c[5] >> 63) & 1;
    u128 t = (u128)c[6] + should_add;
    u256 r = {{(uint64_t)t, c[7] + (uint64_t)(t >> 64), 0, 0}};
    mont_to(o, &r, &FN);
}
/* splitGLV (point_mul_glv.go:59-117) */
static void sc_split_glv(sc *k1, sc *k2, const sc *k) {
    sc c1, c2, tmp;
    sc_mul_g_floored_div(&c1, k, &SC_G1);
    sc_mul_g_floored_div(&c2, k, &SC_G2);
    sc_mul(k2, &c1, &SC_NEG_B1);
    sc_mul(&tmp, &c2, &SC_NEG_B2);
    sc_add(k2, k2, &tmp);
    sc_mul(k1, k2, &SC_NEG_LAMBDA);
    sc_add(k1, k, k1);
}
/* mulBeta (point_mul_glv.go:191-200) */
static void pt_mul_beta(pt *v, const pt *p) { fe_mul(&v->x, &p->x, &FE_BETA); v->y = p->y; v->z = p->z; }

/* scalarMultVartimeGLV (point_mul_glv.go:203-254) */
static void pt_scalar_mult_vartime_glv(pt *v, const sc *s, const pt *p) {
    pt pee = *p, pee_prime;
    pt_mul_beta(&pee_prime, p);
    sc k1, k2;
    sc_split_glv(&k1, &k2, s);
    if (sc_is_gt_half_n(&k1)) { sc_neg(&k1, &k1); pt_neg(&pee, &pee); }
    if (sc_is_gt_half_n(&k2)) { sc_neg(&k2, &k2); pt_neg(&pee_prime, &pee_prime); }
    proj_tbl tbl, tbl_prime;
    proj_tbl_init(&tbl, &pee);
    proj_tbl_init(&tbl_prime, &pee_prime);
    uint8_t b1[32], b2[32];
    sc_get_bytes(b1, &k1);
    sc_get_bytes(b2, &k2);
    pt acc;
    pt_identity(&acc);
    for (int i = 16; i < 32; i++) {
        if (i != 16) for (int d = 0; d < 4; d++) pt_double_complete(&acc, &acc);
        proj_tbl_add_vartime(&tbl, &acc, b1[i] >> 4);
        proj_tbl_add_vartime(&tbl_prime, &acc, b2[i] >> 4);
        for (int d = 0; d < 4; d++) pt_double_complete(&acc, &acc);
        proj_tbl_add_vartime(&tbl, &acc, b1[i] & 0xf);
        proj_tbl_add_vartime(&tbl_prime, &acc, b2[i] & 0xf);
    }
    *v = acc;
}
/* DoubleScalarMultBasepointVartime (point_mul_glv.go:307-317) */
static void pt_double_scalar_mult_basepoint_vartime(pt *v, const sc *u1, const sc *u2, const pt *p) {
    pt u1g, u2p;
    pt_scalar_base_mult_vartime(&u1g, u1);
    pt_scalar_mult_vartime_glv(&u2p, u2, p);
    pt_add_complete(v, &u1g, &u2p);
}
/* MultiScalarMultVartime — Straus (point_mul_multi.go:73-117) */
static void pt_multi_scalar_mult_vartime(pt *v, size_t l, const sc *scalars, const pt *points) {
    if (l == 1) { pt_scalar_mult_vartime_glv(v, &scalars[0], &points[0]); return; }
    proj_tbl *tbls = (proj_tbl *)malloc(sizeof(proj_tbl) * (l ? l : 1));
    uint8_t (*sb)[32] = (uint8_t (*)[32])malloc(32 * (l ? l : 1));
    for (size_t i = 0; i < l; i++) { proj_tbl_init(&tbls[i], &points[i]); sc_get_bytes(sb[i], &scalars[i]); }
    pt acc;
    pt_identity(&acc);
    for (int i = 0; i < 32; i++) {
        if (i != 0) for (int d = 0; d < 4; d++) pt_double_complete(&acc, &acc);
        for (size_t j = 0; j < l; j++) proj_tbl_add_vartime(&tbls[j], &acc, sb[j][i] >> 4);
        for (int d = 0; d < 4; d++) pt_double_complete(&acc, &acc);
        for (size_t j = 0; j < l; j++) proj_tbl_add_vartime(&tbls[j], &acc, sb[j][i] & 0xf);
    }
    *v = acc;
    free(tbls);
    free(sb);
}
/* scalarMultTrivial — MSB-first double-and-add, the reference's own test oracle (point_test.go:392-416) */
static void pt_scalar_mult_trivial(pt *v, const sc *s, const pt *p) {
    uint8_t b[32];
    sc_get_bytes(b, s);
    pt acc;
    pt_identity(&acc);
    for (int i = 0; i < 256; i++) {
        pt_double_complete(&acc, &acc);
        if ((b[i / 8] >> (7 - (i % 8))) & 1) pt_add_complete(&acc, &acc, p);
    }
    *v = acc;
}

/* ------------------------------------------------------------------ */
/* init                                                                */
/* ------------------------------------------------------------------ */
static void hex_to_be32(uint8_t out[32], const char *hex) {
    size_t n = strlen(hex);
    memset(out, 0, 32);
    for (size_t i = 0; i < n; i++) {
        char ch = hex[n - 1 - i];
        uint8_t v = (ch >= '0' && ch <= '9') ? ch - '0' : (ch >= 'a' && ch <= 'f') ? ch - 'a' + 10 : ch - 'A' + 10;
        out[31 - i / 2] |= (i & 1) ? (uint8_t)(v << 4) : v;
    }
}
static void fe_from_hex(fe *o, const char *h) { uint8_t b[32]; hex_to_be32(b, h); fe_set_canonical_bytes(o, b); }
static void sc_from_hex(sc *o, const char *h) { uint8_t b[32]; hex_to_be32(b, h); sc_set_canonical_bytes(o, b); }

static pthread_once_t init_once = PTHREAD_ONCE_INIT;
static void oracle_init_impl(void) {
    mont_ctx_init(&FP);
    mont_ctx_init(&FN);
    memset(&FE_ZERO, 0, sizeof FE_ZERO);
    FE_ONE = FP.one;
    fe_from_u64(&FE_B, 7);                     /* feB */
    fe_from_u64(&FE_B3, 21);                   /* feB3, point_projective.go:21 */
    fe_from_hex(&FE_BETA, "7ae96a2b657c07106e64479eac3434e99cf0497512f58995c1396c28719501ee");   /* point_mul_glv.go:44 */
    fe_from_hex(&FE_C2, "31fdf302724013e57ad13fb38f842afeec184f00a74789dd286729c8303c4a59");     /* field_sqrt_ratio.go:10 */
    fe_from_hex(&FE_GX, "79be667ef9dcbbac55a06295ce870b07029bfcdb2dce28d959f2815b16f81798");     /* point.go:18 */
    fe_from_hex(&FE_GY, "483ada7726a3c4655da4fbfc0e1108a8fd17b448a68554199c47d08ffb10d4b8");     /* point.go:20 */
    memset(&SC_ZERO, 0, sizeof SC_ZERO);
    SC_ONE = FN.one;
    sc_from_hex(&SC_NEG_LAMBDA, "ac9c52b33fa3cf1f5ad9e3fd77ed9ba4a880b9fc8ec739c2e0cfc810b51283cf"); /* point_mul_glv.go:41 */
    sc_from_hex(&SC_NEG_B1, "e4437ed6010e88286f547fa90abfe4c3");                                       /* :47 */
    sc_from_hex(&SC_NEG_B2, "fffffffffffffffffffffffffffffffe8a280ac50774346dd765cda83db1562c");     /* :50 */
    sc_from_hex(&SC_G1, "3086d221a7d46bcde86c90e49284eb153daa8a1471e8ca7fe893209a45dbb031");         /* :53 */
    sc_from_hex(&SC_G2, "e4437ed6010e88286f547fa90abfe4c4221208ac9df506c61571b4ae8ac47f71");         /* :56 */
    build_generator_table();
    cnt_fp_mul = cnt_fn_mul = 0;
}
static inline void oracle_init(void) { pthread_once(&init_once, oracle_init_impl); }

/* ------------------------------------------------------------------ */
/* SHA-256 (FIPS 180-4) — Go's crypto/sha256 in schnorr.go:309-320     */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t h[8]; uint8_t buf[64]; size_t buflen; uint64_t total; } sha256_ctx;
static const uint32_t SHA_K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
static inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
static void sha256_block(sha256_ctx *c, const uint8_t *p) {
    uint32_t w[64], a, b, cc, d, e, f, g, h;
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    a = c->h[0]; b = c->h[1]; cc = c->h[2]; d = c->h[3]; e = c->h[4]; f = c->h[5]; g = c->h[6]; h = c->h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + SHA_K[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & cc) ^ (b & cc);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
    }
    c->h[0] += a; c->h[1] += b; c->h[2] += cc; c->h[3] += d; c->h[4] += e; c->h[5] += f; c->h[6] += g; c->h[7] += h;
}
static void sha256_init(sha256_ctx *c) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(c->h, iv, sizeof iv);
    c->buflen = 0;
    c->total = 0;
}
static void sha256_update(sha256_ctx *c, const uint8_t *d, size_t n) {
    c->total += n;
    while (n) {
        size_t k = 64 - c->buflen;
        if (k > n) k = n;
        memcpy(c->buf + c->buflen, d, k);
        c->buflen += k; d += k; n -= k;
        if (c->buflen == 64) { sha256_block(c, c->buf); c->buflen = 0; }
    }
}
static void sha256_final(sha256_ctx *c, uint8_t out[32]) {
    uint64_t bits = c->total * 8;
    uint8_t pad = 0x80;
    sha256_update(c, &pad, 1);
    uint8_t z = 0;
    while (c->buflen != 56) sha256_update(c, &z, 1);
    uint8_t lb[8];
    for (int i = 0; i < 8; i++) lb[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha256_update(c, lb, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(c->h[i] >> 24); out[4 * i + 1] = (uint8_t)(c->h[i] >> 16); out[4 * i + 2] = (uint8_t)(c->h[i] >> 8); out[4 * i + 3] = (uint8_t)c->h[i]; }
}
void orc_sha256(uint8_t out[32], const uint8_t *data, size_t len) {
    sha256_ctx c;
    sha256_init(&c);
    sha256_update(&c, data, len);
    sha256_final(&c, out);
}

/* ------------------------------------------------------------------ */
/* ECDSA verify                                                        */
/* ------------------------------------------------------------------ */

/* verify (secec/ecdsa.go:392-470), d == nil branch.  q is a valid, non-identity point. */
static int ecdsa_verify_core(const pt *q, const uint8_t *digest, size_t digest_len, const sc *r, const sc *s) {
    if (sc_is_zero(r) || sc_is_zero(s)) return 0;                       /* :400 */
    if (digest_len < 32) return 0;                                      /* hashToScalar, :478-480 */
    sc e;
    sc_set_bytes(&e, digest);                                           /* leftmost 32 bytes, reduced (:483-484) */
    sc s_inv, u1, u2;
    sc_invert(&s_inv, s);                                               /* :428 */
    sc_mul(&u1, &e, &s_inv);
    sc_mul(&u2, r, &s_inv);
    pt R;
    pt_double_scalar_mult_basepoint_vartime(&R, &u1, &u2, q);           /* :436 */
    if (pt_is_identity(&R)) return 0;                                   /* :450 */
    pt Rs;
    pt_rescale(&Rs, &R);                                                /* XBytes, point_s11n.go:119-134 */
    uint8_t xb[32];
    fe_get_bytes(xb, &Rs.x);
    sc v;
    sc_set_bytes(&v, xb);                                               /* v = xR mod n, :460 */
    return sc_eq(&v, r);                                                /* :465 */
}

int orc_ecdsa_verify_raw(const uint8_t q[64], const uint8_t *digest, size_t digest_len,
                         const uint8_t r[32], const uint8_t s[32], int reject_malleable) {
    oracle_init();
    pt Q;
    if (fe_set_canonical_bytes(&Q.x, q) || fe_set_canonical_bytes(&Q.y, q + 32)) return 0;
    if (!xy_on_curve(&Q.x, &Q.y)) return 0;
    Q.z = FE_ONE;
    sc rs, ss;
    /* ParseCompactSignature (s11n.go:129-144): canonical and non-zero */
    if (sc_set_canonical_bytes(&rs, r) || sc_is_zero(&rs)) return 0;
    if (sc_set_canonical_bytes(&ss, s) || sc_is_zero(&ss)) return 0;
    if (reject_malleable && sc_is_gt_half_n(&ss)) return 0;            /* ecdsa.go:212 */
    return ecdsa_verify_core(&Q, digest, digest_len, &rs, &ss);
}

typedef struct {
    size_t lo, hi;
    const uint8_t *q, *d, *r, *s;
    int rm;
    uint8_t *out;
} batch_job;
static void *batch_worker(void *arg) {
    batch_job *j = (batch_job *)arg;
    for (size_t i = j->lo; i < j->hi; i++)
        j->out[i] = (uint8_t)orc_ecdsa_verify_raw(j->q + 64 * i, j->d + 32 * i, 32, j->r + 32 * i, j->s + 32 * i, j->rm);
    return NULL;
}
void orc_ecdsa_verify_batch(size_t n, const uint8_t *q, const uint8_t *digest32, const uint8_t *r, const uint8_t *s,
                            int reject_malleable, uint8_t *out, int nthreads) {
    oracle_init();
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * nthreads);
    batch_job *jobs = (batch_job *)malloc(sizeof(batch_job) * nthreads);
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = (batch_job){n * t / nthreads, n * (t + 1) / nthreads, q, digest32, r, s, reject_malleable, out};
        if (nthreads == 1) batch_worker(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    if (nthreads > 1) for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    free(th);
    free(jobs);
}

/* RecoverPoint (point_s11n.go:245-282) + RecoverPublicKey (secec/ecdsa.go:244-282).
 * Returns 1 and the uncompressed key, or 0 when the reference returns an error. */
int orc_ecdsa_recover(uint8_t out65[65], const uint8_t *digest, size_t digest_len, const uint8_t r[32],
                      const uint8_t s[32], unsigned recovery_id) {
    oracle_init();
    memset(out65, 0, 65);
    sc rs, ss;
    if (sc_set_canonical_bytes(&rs, r) || sc_set_canonical_bytes(&ss, s)) return 0;
    if (sc_is_zero(&rs) || sc_is_zero(&ss)) return 0;                      /* ecdsa.go:245 */
    if (recovery_id >= 4) return 0;                                        /* point_s11n.go:246 */
    unsigned y_is_odd = recovery_id & 1, x_gt_n = (recovery_id >> 1) & 1;
    fe xfe, xfen, fen;
    fe_set_canonical_bytes(&xfe, r);                                       /* n < p: cannot fail */
    { uint8_t nb[32]; u256_to_be(nb, &FN.m); fe_set_canonical_bytes(&fen, nb); }
    fe_add(&xfen, &xfe, &fen);
    if (x_gt_n) xfe = xfen;
    uint8_t xb[32];
    fe_get_bytes(xb, &xfe);
    sc chk;
    int did = sc_set_bytes(&chk, xb);
    if (!((unsigned)did == x_gt_n && sc_eq(&chk, &rs))) return 0;          /* sanity check, point_s11n.go:266-269 */
    uint8_t comp[33];
    comp[0] = (uint8_t)(0x02 + y_is_odd);
    memcpy(comp + 1, xb, 32);
    pt R;
    if (pt_set_bytes(&R, comp, 33)) return 0;
    if (digest_len < 32) return 0;
    sc e, neg_e, r_inv, u1, u2;
    sc_set_bytes(&e, digest);
    sc_neg(&neg_e, &e);
    sc_invert(&r_inv, &rs);
    sc_mul(&u1, &neg_e, &r_inv);
    sc_mul(&u2, &ss, &r_inv);
    pt Q;
    pt_double_scalar_mult_basepoint_vartime(&Q, &u1, &u2, &R);
    if (pt_is_identity(&Q)) return 0;                                      /* NewPublicKeyFromPoint, secec.go:206-209 */
    pt_to_buf(out65, &Q);
    return 1;
}

/* --- strict DER, restating golang.org/x/crypto v0.11.0 cryptobyte (go.mod:8), which is
 * not vendored in the reference: String.ReadASN1 / ReadASN1Integer(*[]byte) as used at
 * secec/s11n.go:89-96.  Pinned by the Wycheproof flag classes (wycheproof_test.go:349-352). */
typedef struct { const uint8_t *p; size_t n; } cb_str;
static int cb_read_asn1(cb_str *s, cb_str *out, uint8_t tag) {
    if (s->n < 2) return 0;
    uint8_t t = s->p[0], lb = s->p[1];
    if ((t & 0x1f) == 0x1f) return 0;                     /* high-tag-number form unsupported */
    size_t hdr, len;
    if ((lb & 0x80) == 0) { hdr = 2; len = lb; }
    else {
        unsigned ll = lb & 0x7f;
        if (ll == 0 || ll > 4 || s->n < 2 + ll) return 0;
        uint32_t l32 = 0;
        for (unsigned i = 0; i < ll; i++) l32 = (l32 << 8) | s->p[2 + i];
        if (l32 < 128) return 0;                          /* should have used short form */
        if ((l32 >> ((ll - 1) * 8)) == 0) return 0;       /* leading zero octet in length */
        hdr = 2 + ll;
        len = l32;
    }
    if (s->n < hdr + len) return 0;
    if (t != tag) return 0;
    out->p = s->p + hdr;
    out->n = len;
    s->p += hdr + len;
    s->n -= hdr + len;
    return 1;
}
static int cb_read_asn1_integer_bytes(cb_str *s, cb_str *out) {
    cb_str b;
    if (!cb_read_asn1(s, &b, 0x02)) return 0;
    if (b.n == 0) return 0;                                               /* checkASN1Integer */
    if (b.n > 1 && ((b.p[0] == 0x00 && (b.p[1] & 0x80) == 0) || (b.p[0] == 0xff && (b.p[1] & 0x80) == 0x80))) return 0;
    if (b.p[0] & 0x80) return 0;                                          /* negative */
    while (b.n > 1 && b.p[0] == 0) { b.p++; b.n--; }
    *out = b;
    return 1;
}
/* bytesToCanonicalScalar (s11n.go:203-218) */
static int bytes_to_canonical_scalar(sc *o, uint8_t out_be[32], const cb_str *b) {
    if (b->n > 32 || b->n == 0) return -1;
    uint8_t tmp[32] = {0};
    memcpy(tmp + 32 - b->n, b->p, b->n);
    if (sc_set_canonical_bytes(o, tmp)) return -1;
    memcpy(out_be, tmp, 32);
    return 0;
}
/* ParseASN1Signature (s11n.go:83-108) */
int orc_parse_asn1_signature(uint8_t r[32], uint8_t s[32], const uint8_t *der, size_t len) {
    oracle_init();
    cb_str in = {der, len}, inner, rb, sb;
    if (!cb_read_asn1(&in, &inner, 0x30) || in.n != 0 || !cb_read_asn1_integer_bytes(&inner, &rb) ||
        !cb_read_asn1_integer_bytes(&inner, &sb) || inner.n != 0)
        return 1;
    sc rs, ss;
    if (bytes_to_canonical_scalar(&rs, r, &rb) || sc_is_zero(&rs)) return 2;
    if (bytes_to_canonical_scalar(&ss, s, &sb) || sc_is_zero(&ss)) return 2;
    return 0;
}
/* PublicKey.Verify, EncodingASN1 (ecdsa.go:171-228) on top of NewPublicKey (secec.go:188-216) */
int orc_ecdsa_verify_asn1(const uint8_t *pub, size_t pub_len, const uint8_t *digest, size_t digest_len,
                          const uint8_t *sig, size_t sig_len, int reject_malleable) {
    oracle_init();
    pt Q;
    if (pt_set_bytes(&Q, pub, pub_len)) return -1;
    if (pt_is_identity(&Q)) return -1;                    /* newPublicKeyFromPoint, secec.go:206-209 */
    uint8_t rb[32], sb[32];
    if (orc_parse_asn1_signature(rb, sb, sig, sig_len)) return 0;
    sc r, s;
    sc_set_canonical_bytes(&r, rb);
    sc_set_canonical_bytes(&s, sb);
    if (reject_malleable && sc_is_gt_half_n(&s)) return 0;
    return ecdsa_verify_core(&Q, digest, digest_len, &r, &s);
}

/* ------------------------------------------------------------------ */
/* BIP-340 verify (secec/bitcoin/schnorr.go:221-253, :420-478)         */
/* ------------------------------------------------------------------ */
static void schnorr_tagged_hash(uint8_t out[32], const char *tag, const uint8_t *a, size_t an, const uint8_t *b, size_t bn,
                                const uint8_t *c, size_t cn) {                         /* schnorr.go:309-320 */
    uint8_t th[32];
    orc_sha256(th, (const uint8_t *)tag, strlen(tag));
    sha256_ctx h;
    sha256_init(&h);
    sha256_update(&h, th, 32);
    sha256_update(&h, th, 32);
    sha256_update(&h, a, an);
    sha256_update(&h, b, bn);
    sha256_update(&h, c, cn);
    sha256_final(&h, out);
}
int orc_schnorr_verify(const uint8_t pk[32], const uint8_t *msg, size_t msg_len, const uint8_t *sig, size_t sig_len) {
    oracle_init();
    /* NewSchnorrPublicKey: lift_x via compressed decode with prefix 0x02 (schnorr.go:257-275) */
    uint8_t comp[33];
    comp[0] = 0x02;
    memcpy(comp + 1, pk, 32);
    pt P;
    if (pt_set_bytes(&P, comp, 33)) return -1;
    if (sig_len != 64) return 0;                                                       /* :222 */
    /* parseSchnorrSignature (:420-449) */
    u256 rl, tmp;
    u256_from_be(&rl, sig);
    if (reduce_saturated(&tmp, &rl, &FP)) return 0;                                    /* r >= p */
    sc s, e;
    if (sc_set_canonical_bytes(&s, sig + 32)) return 0;                                /* s >= n */
    uint8_t eb[32];
    schnorr_tagged_hash(eb, "BIP0340/challenge", sig, 32, pk, 32, msg, msg_len);
    sc_set_bytes(&e, eb);
    sc_neg(&e, &e);                                                                    /* :244 */
    pt R;
    pt_double_scalar_mult_basepoint_vartime(&R, &s, &e, &P);                           /* :245 */
    /* verifySchnorrSignatureR (:451-478) */
    if (pt_is_identity(&R)) return 0;
    pt Rs;
    pt_rescale(&Rs, &R);
    if (fe_is_odd(&Rs.y)) return 0;
    uint8_t xb[32];
    fe_get_bytes(xb, &Rs.x);
    return memcmp(xb, sig, 32) == 0;
}

/* ------------------------------------------------------------------ */
/* exported byte-level wrappers                                        */
/* ------------------------------------------------------------------ */
static void fe_in(fe *o, const uint8_t b[32]) { u256 l; u256_from_be(&l, b); reduce_saturated(&l, &l, &FP); mont_to(o, &l, &FP); }
static void sc_in(sc *o, const uint8_t b[32]) { sc_set_bytes(o, b); }

int orc_fp_is_canonical(const uint8_t a[32]) { oracle_init(); fe t; return fe_set_canonical_bytes(&t, a) == 0; }
void orc_fp_reduce(uint8_t out[32], const uint8_t a[32], int *did) {
    oracle_init();
    u256 l; u256_from_be(&l, a);
    int d = reduce_saturated(&l, &l, &FP);
    if (did) *did = d;
    u256_to_be(out, &l);
}
#define FP_BINOP(name, op) void name(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) { \
    oracle_init(); fe x, y, z; fe_in(&x, a); fe_in(&y, b); op(&z, &x, &y); fe_get_bytes(out, &z); }
FP_BINOP(orc_fp_mul, fe_mul)
FP_BINOP(orc_fp_add, fe_add)
FP_BINOP(orc_fp_sub, fe_sub)
void orc_fp_sqr(uint8_t out[32], const uint8_t a[32]) { oracle_init(); fe x, z; fe_in(&x, a); fe_sqr(&z, &x); fe_get_bytes(out, &z); }
void orc_fp_neg(uint8_t out[32], const uint8_t a[32]) { oracle_init(); fe x, z; fe_in(&x, a); fe_neg(&z, &x); fe_get_bytes(out, &z); }
void orc_fp_inv(uint8_t out[32], const uint8_t a[32]) { oracle_init(); fe x, z; fe_in(&x, a); fe_invert(&z, &x); fe_get_bytes(out, &z); }
int orc_fp_sqrt(uint8_t out[32], const uint8_t a[32]) { oracle_init(); fe x, z; fe_in(&x, a); int ok = fe_sqrt(&z, &x); fe_get_bytes(out, &z); return ok; }

int orc_fn_is_canonical(const uint8_t a[32]) { oracle_init(); sc t; return sc_set_canonical_bytes(&t, a) == 0; }
void orc_fn_reduce(uint8_t out[32], const uint8_t a[32], int *did) {
    oracle_init(); sc t; int d = sc_set_bytes(&t, a); if (did) *did = d; sc_get_bytes(out, &t);
}
#define FN_BINOP(name, op) void name(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) { \
    oracle_init(); sc x, y, z; sc_in(&x, a); sc_in(&y, b); op(&z, &x, &y); sc_get_bytes(out, &z); }
FN_BINOP(orc_fn_mul, sc_mul)
FN_BINOP(orc_fn_add, sc_add)
FN_BINOP(orc_fn_sub, sc_sub)
void orc_fn_neg(uint8_t out[32], const uint8_t a[32]) { oracle_init(); sc x, z; sc_in(&x, a); sc_neg(&z, &x); sc_get_bytes(out, &z); }
void orc_fn_inv(uint8_t out[32], const uint8_t a[32]) { oracle_init(); sc x, z; sc_in(&x, a); sc_invert(&z, &x); sc_get_bytes(out, &z); }
int orc_fn_is_gt_half_n(const uint8_t a[32]) { oracle_init(); sc x; sc_in(&x, a); return sc_is_gt_half_n(&x); }
void orc_fn_split_glv(uint8_t k1[32], uint8_t k2[32], const uint8_t k[32]) {
    oracle_init(); sc x, a, b; sc_in(&x, k); sc_split_glv(&a, &b, &x); sc_get_bytes(k1, &a); sc_get_bytes(k2, &b);
}

void orc_point_generator(uint8_t out[65]) { oracle_init(); pt g; pt_generator(&g); pt_to_buf(out, &g); }
void orc_point_identity(uint8_t out[65]) { memset(out, 0, 65); }
int orc_point_from_bytes(uint8_t out[65], const uint8_t *src, size_t len) {
    oracle_init(); pt p; if (pt_set_bytes(&p, src, len)) return -1; pt_to_buf(out, &p); return 0;
}
int orc_point_on_curve_xy(const uint8_t x[32], const uint8_t y[32]) {
    oracle_init(); fe fx, fy;
    if (fe_set_canonical_bytes(&fx, x) || fe_set_canonical_bytes(&fy, y)) return 0;
    return xy_on_curve(&fx, &fy);
}
void orc_point_compressed(uint8_t out[33], size_t *out_len, const uint8_t p[65]) {      /* point_s11n.go:90-117 */
    oracle_init();
    if (p[0] == 0x00) { out[0] = 0x00; *out_len = 1; return; }
    pt a; pt_from_buf(&a, p);
    out[0] = fe_is_odd(&a.y) ? 0x03 : 0x02;
    memcpy(out + 1, p + 1, 32);
    *out_len = 33;
}
void orc_point_add(uint8_t out[65], const uint8_t a[65], const uint8_t b[65]) {
    oracle_init(); pt x, y, z; pt_from_buf(&x, a); pt_from_buf(&y, b); pt_add_complete(&z, &x, &y); pt_to_buf(out, &z);
}
void orc_point_add_randz(uint8_t out[65], const uint8_t a[65], const uint8_t za[32], const uint8_t b[65], const uint8_t zb[32]) {
    oracle_init(); pt x, y, z; fe fa, fb;
    pt_from_buf(&x, a); pt_from_buf(&y, b); fe_in(&fa, za); fe_in(&fb, zb);
    pt_scale_z(&x, &x, &fa); pt_scale_z(&y, &y, &fb);
    pt_add_complete(&z, &x, &y); pt_to_buf(out, &z);
}
int orc_point_equal_randz(const uint8_t a[65], const uint8_t za[32], const uint8_t b[65], const uint8_t zb[32]) {
    oracle_init(); pt x, y; fe fa, fb;
    pt_from_buf(&x, a); pt_from_buf(&y, b); fe_in(&fa, za); fe_in(&fb, zb);
    pt_scale_z(&x, &x, &fa); pt_scale_z(&y, &y, &fb);
    return pt_equal(&x, &y);
}
void orc_point_double(uint8_t out[65], const uint8_t a[65]) {
    oracle_init(); pt x, z; pt_from_buf(&x, a); pt_double_complete(&z, &x); pt_to_buf(out, &z);
}
void orc_point_neg(uint8_t out[65], const uint8_t a[65]) {
    oracle_init(); pt x, z; pt_from_buf(&x, a); pt_neg(&z, &x); pt_to_buf(out, &z);
}
void orc_scalar_mult_vartime(uint8_t out[65], const uint8_t k[32], const uint8_t p[65]) {
    oracle_init(); pt x, z; sc s; pt_from_buf(&x, p); sc_in(&s, k); pt_scalar_mult_vartime_glv(&z, &s, &x); pt_to_buf(out, &z);
}
void orc_scalar_mult_vartime_randz(uint8_t out[65], const uint8_t k[32], const uint8_t p[65], const uint8_t zb[32]) {
    oracle_init(); pt x, z; sc s; fe fz;
    pt_from_buf(&x, p); sc_in(&s, k); fe_in(&fz, zb); pt_scale_z(&x, &x, &fz);
    pt_scalar_mult_vartime_glv(&z, &s, &x); pt_to_buf(out, &z);
}
void orc_scalar_mult_trivial(uint8_t out[65], const uint8_t k[32], const uint8_t p[65]) {
    oracle_init(); pt x, z; sc s; pt_from_buf(&x, p); sc_in(&s, k); pt_scalar_mult_trivial(&z, &s, &x); pt_to_buf(out, &z);
}
void orc_scalar_base_mult_vartime(uint8_t out[65], const uint8_t k[32]) {
    oracle_init(); pt z; sc s; sc_in(&s, k); pt_scalar_base_mult_vartime(&z, &s); pt_to_buf(out, &z);
}
void orc_double_scalar_mult_basepoint_vartime(uint8_t out[65], const uint8_t u1[32], const uint8_t u2[32], const uint8_t p[65]) {
    oracle_init(); pt x, z; sc a, b; pt_from_buf(&x, p); sc_in(&a, u1); sc_in(&b, u2);
    pt_double_scalar_mult_basepoint_vartime(&z, &a, &b, &x); pt_to_buf(out, &z);
}
void orc_multi_scalar_mult_vartime(uint8_t out[65], size_t n, const uint8_t *scalars, const uint8_t *points) {
    oracle_init();
    sc *s = (sc *)malloc(sizeof(sc) * (n ? n : 1));
    pt *p = (pt *)malloc(sizeof(pt) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) { sc_in(&s[i], scalars + 32 * i); pt_from_buf(&p[i], points + 65 * i); }
    pt z;
    pt_multi_scalar_mult_vartime(&z, n, s, p);
    pt_to_buf(out, &z);
    free(s); free(p);
}
void orc_generator_table_entry(uint8_t out[64], unsigned i, unsigned j) {
    oracle_init();
    fe_get_bytes(out, &G_TABLE[i][j].x);
    fe_get_bytes(out + 32, &G_TABLE[i][j].y);
}
void orc_generator_table_sha256(uint8_t out[32]) {
    oracle_init();
    sha256_ctx c;
    sha256_init(&c);
    uint8_t e[64];
    for (unsigned i = 0; i < 32; i++)
        for (unsigned j = 0; j < 255; j++) { orc_generator_table_entry(e, i, j); sha256_update(&c, e, 64); }
    sha256_final(&c, out);
}

void orc_counters_reset(void) { cnt_fp_mul = cnt_fn_mul = 0; }
void orc_counters_get(uint64_t *fp_mul, uint64_t *fn_mul) { *fp_mul = cnt_fp_mul; *fn_mul = cnt_fn_mul; }

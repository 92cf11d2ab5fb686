"""The constant-time CPU twins (s2k_ct_*) against the oracle and the reference's
vectors: Wycheproof ECDH (secec/wycheproof_test.go:303-305), the libsecp256k1 KAT
(point_test.go:242-261), GLV boundary scalars (point_mul_glv_test.go:25-45), RFC 6979 signatures
(ecdsa_k_test.go:244-278).  Runs without a GPU; the library has to be built (hipcc).
"""
import os
import random

import pytest

import pyref as R
from conftest import load_golden

b32 = R.b32
H = bytes.fromhex


@pytest.fixture(scope="module")
def S():
    import secp256k1_voi_amd as S
    if not os.path.exists(S.LIB_PATH):
        pytest.skip("library not built")
    S.load_library()
    return S


def test_ct_scalar_mult_kats(S, oracle):
    kat = load_golden("kats.json")["libsecp256k1_ecmult_const"]
    assert S.ct_scalar_mult(H(kat["xn"]), H(kat["a"])).hex() == kat["b"]
    d = load_golden("wycheproof_ecdh.json")
    n = 0
    for c in d["cases"]:
        pt = oracle.point_from_bytes(H(c["point"]))
        out = S.ct_scalar_mult(H(c["private"]), pt)
        assert out[1:33] == H(c["shared"]), c["tcId"]
        if len(c["point"]) == 130 and int(c["private"], 16) < R.N and int(c["private"], 16) > 0:
            assert S.ct_ecdh(H(c["private"]), H(c["point"])) == H(c["shared"]), c["tcId"]
            n += 1
    assert n > 100


def test_ct_scalar_mult_random_and_edges(S, oracle):
    rnd = random.Random(201)
    g = load_golden("kats.json")["glv"]
    ks = [0, 1, 2, 3, R.N - 1, R.N - 2, R.N, R.N + 1, 2**256 - 1, R.LAMBDA, R.N - R.LAMBDA, 2**128, 2**128 - 1, 2**127, 15, 16, 17] + \
         [int(s, 16) for s in g["boundary_scalars"]] + [rnd.randrange(2**256) for _ in range(150)]
    for v in ks:
        p = R.enc65(R.mul(rnd.randrange(1, R.N), R.G))
        assert S.ct_scalar_mult(b32(v), p) == oracle.scalar_mult_trivial(b32(v % R.N) if v >= R.N else b32(v), p), hex(v)
    # identity in, identity out; malformed records are refused
    assert S.ct_scalar_mult(b32(5), bytes(65)) == bytes(65)
    good = R.enc65(R.G)
    off = bytearray(good); off[64] ^= 1
    assert S.ct_scalar_mult(b32(5), bytes(off)) is None
    assert S.ct_scalar_mult(b32(5), b"\x04" + b32(R.P) + b32(7)) is None
    assert S.ct_scalar_mult(b32(5), b"\x00" + bytes(63) + b"\x01") is None


def test_ct_scalar_base_mult(S, oracle):
    rnd = random.Random(202)
    ks = [0, 1, 2, R.N - 1, R.N, R.N + 5, 2**256 - 1, 0xF, 0x10, 0xFF, 2**255, 2**252] + [1 << (4 * i) for i in range(64)] + \
         [15 << (4 * i) for i in range(64)] + [rnd.randrange(2**256) for _ in range(200)]
    for v in ks:
        assert S.ct_scalar_base_mult(b32(v)) == oracle.scalar_base_mult_vartime(b32(v)), hex(v)


def test_ct_ecdh_rejects_bad_inputs(S):
    q = R.enc65(R.mul(12345, R.G))
    assert S.ct_ecdh(b32(0), q) is None
    assert S.ct_ecdh(b32(R.N), q) is None
    assert S.ct_ecdh(b32(7), bytes(65)) is None          # identity is not a public key
    x = S.ct_ecdh(b32(7), q)
    assert x == b32(R.mul(7 * 12345, R.G)[0])


def test_ct_sign_raw(S, oracle):
    """signatures made by the CT primitive verify with the oracle (and an independent big-int check of r, s,
    low-s and the recovery id); the RFC 6979 vectors are reproduced when fed their nonces."""
    rnd = random.Random(203)
    for _ in range(60):
        d, k = rnd.randrange(1, R.N), rnd.randrange(1, R.N)
        digest = rnd.randbytes(32)
        r, s, rid = S.ct_ecdsa_sign_raw(b32(d), digest, b32(k))
        Rp = R.mul(k, R.G)
        e = int.from_bytes(digest, "big") % R.N
        s_ref = pow(k, -1, R.N) * (e + (Rp[0] % R.N) * d) % R.N
        flipped = s_ref > R.N // 2
        if flipped:
            s_ref = R.N - s_ref
        assert (int.from_bytes(r, "big"), int.from_bytes(s, "big")) == (Rp[0] % R.N, s_ref)
        assert rid == ((Rp[1] & 1) ^ int(flipped)) | (2 if Rp[0] >= R.N else 0)
        Q = R.mul(d, R.G)
        assert oracle.ecdsa_verify_raw(b32(Q[0]) + b32(Q[1]), digest, r, s, reject_malleable=True)
        assert oracle.ecdsa_recover(digest, r, s, rid) == R.enc65(Q)
    assert S.ct_ecdsa_sign_raw(b32(0), bytes(32), b32(1)) is None
    assert S.ct_ecdsa_sign_raw(b32(1), bytes(32), b32(0)) is None
    assert S.ct_ecdsa_sign_raw(b32(R.N), bytes(32), b32(1)) is None
    assert S.ct_ecdsa_sign_raw(b32(1), bytes(32), b32(R.N)) is None


def test_ct_multi_scalar_mult_vs_oracle(S, oracle):
    """Point.MultiScalarMult (point_mul_multi.go:25-67), constant-time form: equal to the oracle's Straus
    (its restatement of MultiScalarMultVartime :73-117 — same group element) for 0, 1, 2, 32, 64 terms, with the
    edge scalars and the identity / repeated / opposite points the reference's tests mix in
    (point_mul_multi_test.go:14-72)."""
    rnd = random.Random(515)
    assert S.ct_multi_scalar_mult([], []) == bytes(65)
    for n in (1, 2, 3, 32, 64):
        pts = [R.enc65(R.mul(rnd.randrange(1, R.N), R.G)) for _ in range(n)]
        ks = [b32(rnd.randrange(R.N)) for _ in range(n)]
        if n >= 3:
            ks[0], ks[1] = b32(0), b32(R.N - 1)
            pts[2] = bytes(65)                              # the identity as a term
        if n >= 32:
            pts[5] = pts[4]                                 # the same point twice
            pts[7] = oracle.point_neg(pts[6])               # and a pair of opposites with equal scalars: they cancel
            ks[7] = ks[6]
            ks[8] = b32((1 << 256) - 1)                     # SetBytes semantics: reduced mod n
            ks[9] = b32(0x0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f0f)
        exp = oracle.multi_scalar_mult_vartime([oracle.fn_reduce(k)[0] for k in ks], pts)
        assert S.ct_multi_scalar_mult(ks, pts) == exp, n
    # one term: exactly Point.ScalarMult (:31-33)
    p, k = R.enc65(R.mul(77, R.G)), b32(rnd.randrange(R.N))
    assert S.ct_multi_scalar_mult([k], [p]) == S.ct_scalar_mult(k, p)
    # known discrete logs: sum k_i (d_i G) == (sum k_i d_i) G
    ds = [rnd.randrange(1, R.N) for _ in range(17)]
    kk = [rnd.randrange(R.N) for _ in range(17)]
    pts = [S.ct_scalar_base_mult(b32(d)) for d in ds]
    assert S.ct_multi_scalar_mult([b32(k) for k in kk], pts) == S.ct_scalar_base_mult(b32(sum(a * b for a, b in zip(ds, kk)) % R.N))
    # malformed record -> None; length mismatch -> the reference's panic
    assert S.ct_multi_scalar_mult([k, k], [p, b"\x05" + p[1:]]) is None
    with pytest.raises(ValueError):
        S.ct_multi_scalar_mult([k], [p, p])


def test_ct_multi_scalar_mult_operation_count(S):
    """The sequence of field operations of the constant-time MultiScalarMult depends on the number of terms only:
    same multiplication count for zero, sparse, dense and random scalars, and the count is what Straus with full
    table scans costs — 15 l table entries (7 doublings + 7 additions + the copy), 252 shared doublings, 64 l
    additions (point_mul_multi.go:35-66); the variable-time form would skip the zero digits."""
    lib = S.load_library()
    rnd = random.Random(516)
    S.ct_scalar_base_mult(b32(1))
    for n in (2, 5, 32):
        pts = [S.ct_scalar_base_mult(b32(rnd.randrange(1, R.N))) for _ in range(n)]
        lib.s2k_ct_debug_fe_mul_count()
        counts = set()
        for fill in (lambda: 0, lambda: 1, lambda: R.N - 1, lambda: (1 << 256) - 1, lambda: 1 << 255, lambda: 0x10001,
                     lambda: rnd.randrange(R.N), lambda: rnd.randrange(R.N), lambda: rnd.randrange(1 << 64)):
            assert S.ct_multi_scalar_mult([b32(fill()) for _ in range(n)], pts) is not None
            counts.add(lib.s2k_ct_debug_fe_mul_count())
        assert len(counts) == 1, (n, counts)
        # per operation in ct_cpu.cpp: addition 14 products (12 + two by the constant 21), doubling 9 (8 + one by 21);
        # output: one inversion (255 squarings + 15 products) + 2; input validation: 3 per record
        add, dbl = 14, 9
        expect = n * (7 * dbl + 7 * add) + 252 * dbl + 64 * n * add + (255 + 15 + 2) + 3 * n
        assert counts == {expect}, (n, counts, expect)


def test_ct_operation_count_is_scalar_independent(S):
    """The variable-time paths skip zero digits and pick formulas by value; the constant-time twins must
    execute the same sequence whatever the scalar: same number of field multiplications for 0, 1, sparse,
    dense, boundary and random scalars (table scans and selections are masks, not branches)."""
    lib = S.load_library()
    rnd = random.Random(204)
    q = R.enc65(R.mul(rnd.randrange(1, R.N), R.G))
    S.ct_scalar_base_mult(b32(1))             # builds the generator tables once
    lib.s2k_ct_debug_fe_mul_count()
    scalars = [0, 1, 2, R.N - 1, 2**128, 2**255, (1 << 256) - 1, R.LAMBDA, 0x1111111111111111, 1 << 252] + \
        [rnd.randrange(R.N) for _ in range(20)]
    counts = set()
    for v in scalars:
        S.ct_scalar_mult(b32(v), q)
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts
    counts = set()
    for v in scalars:
        S.ct_scalar_base_mult(b32(v))
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts
    counts = set()
    for v in scalars[1:4] + scalars[10:]:
        assert S.ct_ecdsa_sign_raw(b32(v % R.N or 1), bytes(range(32)), b32((v * 7 + 1) % R.N or 1)) is not None
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts


# ---- single operations (s2k_ct_point_* / s2k_ct_scalar_* / s2k_ct_fe_op): what Point / Scalar / field.Element methods bind to ----
N_HEX = "fffffffffffffffffffffffffffffffebaaedce6af48a03bbfd25e8cd0364141"
HALF_N = 0x7fffffffffffffffffffffffffffffff5d576e7357a4501ddfe92f46681b20a0
IDENT = bytes(65)


def rec(pt):
    return IDENT if pt is None else b"\x04" + b32(pt[0]) + b32(pt[1])


def test_ct_scalar_single_ops_boundaries_and_oracle(S, oracle):
    """scalar_test.go:27-41 (SetBytes / SetCanonicalBytes on N, N+1, N+2, N+2^128), :76-95 (IsGreaterThanHalfN around N/2),
    and every arithmetic method against the oracle on boundary and random scalars (scalar.go:66-93, scalar_invert.go:11)."""
    geq_n = [R.N, R.N + 1, R.N + 2, R.N + 2**128]
    for v, red in zip(geq_n, [0, 1, 2, 2**128]):
        assert S.ct_scalar_set_bytes(b32(v)) == (b32(red), 1)
        for fn in (lambda: S.ct_scalar_op(S.OP_NEG, b32(v)), lambda: S.ct_scalar_op(S.OP_ADD, b32(1), b32(v)),
                   lambda: S.ct_scalar_predicate(0, b32(v)), lambda: S.ct_scalar_conditional_select(b32(1), b32(v), 0)):
            with pytest.raises(ValueError):            # SetCanonicalBytes: errNonCanonicalEncoding
                fn()
    assert S.ct_scalar_set_bytes(b32(R.N - 1)) == (b32(R.N - 1), 0)
    for v, gt in ((HALF_N, 0), (HALF_N - 1, 0), (HALF_N + 1, 1), (HALF_N + 2, 1), (0, 0), (R.N - 1, 1)):
        assert S.ct_scalar_predicate(1, b32(v)) == gt == int(oracle.fn_is_gt_half_n(b32(v)))
    assert S.ct_scalar_predicate(0, b32(0)) == 1 and S.ct_scalar_predicate(0, b32(1)) == 0
    rnd = random.Random(61)
    vals = [0, 1, 2, R.N - 1, R.N - 2, HALF_N, HALF_N + 1, 2**128, 2**128 - 1, 2**255, R.LAMBDA] + [rnd.randrange(R.N) for _ in range(40)]
    for a in vals:
        A = b32(a)
        assert S.ct_scalar_op(S.OP_NEG, A) == oracle.fn_neg(A) == b32(-a % R.N)
        assert S.ct_scalar_op(S.OP_SQR, A) == oracle.fn_mul(A, A)
        assert S.ct_scalar_op(S.OP_INV, A) == oracle.fn_inv(A) == b32(pow(a, -1, R.N) if a else 0)
        assert S.ct_scalar_conditional_negate(A, 0) == A and S.ct_scalar_conditional_negate(A, 1) == b32(-a % R.N)
        assert S.ct_scalar_conditional_negate(A, 2**63) == b32(-a % R.N)          # ctrl: anything but 0
        for b in rnd.sample(vals, 6):
            B = b32(b)
            assert S.ct_scalar_op(S.OP_MUL, A, B) == oracle.fn_mul(A, B) == b32(a * b % R.N)
            assert S.ct_scalar_op(S.OP_ADD, A, B) == oracle.fn_add(A, B) == b32((a + b) % R.N)
            assert S.ct_scalar_op(S.OP_SUB, A, B) == oracle.fn_sub(A, B) == b32((a - b) % R.N)
            assert S.ct_scalar_conditional_select(A, B, 0) == A and S.ct_scalar_conditional_select(A, B, 1) == B
            assert S.ct_scalar_predicate(2, A, B) == int(a == b)
    # Sum / Product of scalar_test.go:43-64 are loops over Add / Multiply: 1 + 1 + 1 = 3, 2 * 3 = 6
    three = S.ct_scalar_op(S.OP_ADD, S.ct_scalar_op(S.OP_ADD, b32(1), b32(1)), b32(1))
    assert three == b32(3) and S.ct_scalar_op(S.OP_MUL, b32(2), b32(3)) == b32(6)
    with pytest.raises(ValueError):
        S.ct_scalar_op(99, b32(1), b32(1))
    with pytest.raises(ValueError):
        S.ct_scalar_op(S.OP_MUL, b32(1), None)


def test_ct_fe_single_ops_vs_oracle(S, oracle):
    """field.Element methods (field.go:61-104, Invert, Sqrt) on the field's boundary values (field_test.go:29-41: p is not a
    canonical encoding) and random elements, against the oracle and big integers."""
    rnd = random.Random(62)
    for v in (R.P, R.P + 1, 2**256 - 1):
        with pytest.raises(ValueError):
            S.ct_fe_op(S.OP_SQR, b32(v))
    vals = [0, 1, 2, 7, R.P - 1, R.P - 2, 2**255, 0x1000003D0, 0x1000003D1, 2**256 - 0x1000003D1 - 1] + [rnd.randrange(R.P) for _ in range(40)]
    squares = 0
    for a in vals:
        A = b32(a)
        assert S.ct_fe_op(S.OP_SQR, A) == (oracle.fp_sqr(A), 1)
        assert S.ct_fe_op(S.OP_NEG, A) == (oracle.fp_neg(A), 1)
        assert S.ct_fe_op(S.OP_INV, A) == (oracle.fp_inv(A), 1) == (b32(pow(a, -1, R.P) if a else 0), 1)
        root, ok = S.ct_fe_op(S.OP_SQRT, A)
        o_root, o_ok = oracle.fp_sqrt(A)
        assert (root, ok) == (bytes(o_root), int(o_ok))
        if ok:
            squares += 1
            assert pow(int.from_bytes(root, "big"), 2, R.P) == a
        else:
            assert root == b32(0)
        for b in rnd.sample(vals, 5):
            B = b32(b)
            assert S.ct_fe_op(S.OP_MUL, A, B)[0] == oracle.fp_mul(A, B) == b32(a * b % R.P)
            assert S.ct_fe_op(S.OP_ADD, A, B)[0] == oracle.fp_add(A, B)
            assert S.ct_fe_op(S.OP_SUB, A, B)[0] == oracle.fp_sub(A, B)
    assert 10 < squares < len(vals) - 10


def test_ct_point_single_ops_properties_and_oracle(S, oracle):
    """point_test.go:136-213: a + 0 = 0 + a = a, a + a = 2a, a + b = b + a, 2 * 0 = 0, a - 0 = a, 0 - a = -a, a - a = 0,
    a - b = a + (-b); every result also against the oracle's complete formulas and the affine big-integer group law;
    Equal / IsIdentity / IsYOdd / ConditionalSelect / ConditionalNegate (point.go:102-160); malformed records are errors."""
    rnd = random.Random(63)
    pts = [R.mul(rnd.randrange(1, R.N), R.G) for _ in range(12)] + [R.G, R.mul(R.N - 1, R.G), R.mul(2, R.G)]
    for a in pts:
        A, negA = rec(a), rec((a[0], R.P - a[1]))
        assert S.ct_point_add(A, IDENT) == A == S.ct_point_add(IDENT, A)
        assert S.ct_point_add(A, A) == S.ct_point_double(A) == oracle.point_double(A) == rec(R.add(a, a))
        assert S.ct_point_subtract(A, IDENT) == A and S.ct_point_subtract(IDENT, A) == negA == S.ct_point_negate(A)
        assert S.ct_point_subtract(A, A) == IDENT == S.ct_point_add(A, negA)
        assert S.ct_point_negate(A) == oracle.point_neg(A)
        assert S.ct_point_conditional_negate(A, 0) == A and S.ct_point_conditional_negate(A, 5) == negA
        assert S.ct_point_equal(A, A) == 1 and S.ct_point_equal(A, negA) == 0 and S.ct_point_equal(A, IDENT) == 0
        assert S.ct_point_is_identity(A) == 0 and S.ct_point_is_y_odd(A) == (a[1] & 1)
        for b in rnd.sample(pts, 4):
            B = rec(b)
            ab = S.ct_point_add(A, B)
            assert ab == S.ct_point_add(B, A) == oracle.point_add(A, B) == rec(R.add(a, b))
            assert S.ct_point_subtract(A, B) == S.ct_point_add(A, S.ct_point_negate(B)) == rec(R.add(a, (b[0], R.P - b[1])))
            assert S.ct_point_conditional_select(A, B, 0) == A and S.ct_point_conditional_select(A, B, 1) == B
            assert S.ct_point_equal(A, B) == int(a == b)
    assert S.ct_point_double(IDENT) == IDENT == S.ct_point_negate(IDENT) == S.ct_point_add(IDENT, IDENT)
    assert S.ct_point_is_identity(IDENT) == 1 and S.ct_point_is_y_odd(IDENT) == 0 and S.ct_point_equal(IDENT, IDENT) == 1
    g = rec(R.G)
    bad = [b"\x04" + b32(R.G[0]) + b32(R.G[1] ^ 1),            # off the curve
           b"\x04" + b32(R.G[0]) + b32(R.P + 1),               # non-canonical y
           b"\x04" + b32(R.P) + b32(0),                         # non-canonical x
           b"\x02" + g[1:], b"\x00" + g[1:], b"\x04" + bytes(64),   # wrong tag, identity tag with a body, (0, 0)
           b"\x00" * 64 + b"\x01"]
    for rec_bad in bad:
        for fn in (lambda: S.ct_point_add(g, rec_bad), lambda: S.ct_point_add(rec_bad, g), lambda: S.ct_point_double(rec_bad),
                   lambda: S.ct_point_negate(rec_bad), lambda: S.ct_point_equal(g, rec_bad), lambda: S.ct_point_is_identity(rec_bad),
                   lambda: S.ct_point_conditional_select(g, rec_bad, 0), lambda: S.ct_point_subtract(rec_bad, g)):
            with pytest.raises(ValueError):
                fn()


def test_ct_single_ops_operation_count_is_value_independent(S):
    """The field multiplications of a single operation do not depend on the operands' values - identity or not, equal or
    opposite points, zero or dense scalars."""
    rnd = random.Random(64)
    a, b = rec(R.mul(rnd.randrange(1, R.N), R.G)), rec(R.mul(rnd.randrange(1, R.N), R.G))
    nb = rec((int.from_bytes(b[1:33], "big"), R.P - int.from_bytes(b[33:], "big")))
    lib = S.load_library()
    counts = set()
    for x, y in ((a, b), (a, a), (b, nb), (IDENT, a), (a, IDENT), (IDENT, IDENT)):
        lib.s2k_ct_debug_fe_mul_count()
        S.ct_point_add(x, y)
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts
    counts = set()
    for x in (a, IDENT, rec(R.G)):
        lib.s2k_ct_debug_fe_mul_count()
        S.ct_point_double(x)
        S.ct_point_equal(x, a)
        S.ct_point_is_identity(x)
        counts.add(lib.s2k_ct_debug_fe_mul_count())
    assert len(counts) == 1, counts


def test_ct_scalar_mult_vs_trivial_double_and_add_through_the_single_ops(S, oracle):
    """The reference's own cross-check (point_test.go:392-416: every scalar multiplication against the trivially correct
    double-and-add `scalarMultTrivial`), with the double-and-add run through s2k_ct_point_double / s2k_ct_point_add: the
    single operations and the table-driven multiplications must describe the same group."""
    rnd = random.Random(65)
    for it in range(12):
        k = [0, 1, 2, R.N - 1, R.LAMBDA, 2**128 + 1][it] if it < 6 else rnd.randrange(R.N)
        P = rec(R.mul(rnd.randrange(1, R.N), R.G))
        acc = IDENT
        for bit in range(k.bit_length() - 1, -1, -1):
            acc = S.ct_point_double(acc)
            if (k >> bit) & 1:
                acc = S.ct_point_add(acc, P)
        want = S.ct_scalar_mult(b32(k), P)
        assert acc == want == oracle.scalar_mult_vartime(b32(k), P), (it, k)
        # ... and the base multiplication is the same map on G
        if it % 3 == 0:
            g = rec(R.G)
            accg = IDENT
            for bit in range(k.bit_length() - 1, -1, -1):
                accg = S.ct_point_double(accg)
                if (k >> bit) & 1:
                    accg = S.ct_point_add(accg, g)
            assert accg == S.ct_scalar_base_mult(b32(k))


def test_ct_single_ops_argument_checks(S):
    """NULL pointers and unknown op codes are S2K_ERR_ARG (-3 ... the library's argument error), never a crash."""
    lib = S.load_library()
    import ctypes as C
    g, one, out65, out32, flag = rec(R.G), b32(1), C.create_string_buffer(65), C.create_string_buffer(32), C.c_uint64(0)
    ERR = lib.s2k_ct_point_add(None, g, out65)
    assert ERR != 0
    for rc in (lib.s2k_ct_point_add(g, None, out65), lib.s2k_ct_point_add(g, g, None), lib.s2k_ct_point_double(None, out65),
               lib.s2k_ct_point_double(g, None), lib.s2k_ct_point_subtract(g, g, None), lib.s2k_ct_point_negate(None, out65),
               lib.s2k_ct_point_conditional_select(g, None, 0, out65), lib.s2k_ct_point_equal(g, g, None),
               lib.s2k_ct_point_is_identity(None, C.byref(flag)), lib.s2k_ct_point_is_y_odd(g, None),
               lib.s2k_ct_scalar_op(S.OP_ADD, one, None, out32), lib.s2k_ct_scalar_op(S.OP_ADD, None, one, out32),
               lib.s2k_ct_scalar_op(S.OP_NEG, one, None, None), lib.s2k_ct_scalar_op(S.OP_SQRT, one, None, out32), lib.s2k_ct_scalar_op(-1, one, one, out32),
               lib.s2k_ct_scalar_conditional_select(one, None, 1, out32), lib.s2k_ct_scalar_conditional_negate(None, 1, out32),
               lib.s2k_ct_scalar_predicate(2, one, None, C.byref(flag)), lib.s2k_ct_scalar_predicate(3, one, one, C.byref(flag)),
               lib.s2k_ct_scalar_predicate(0, one, None, None), lib.s2k_ct_scalar_set_bytes(None, out32, None),
               lib.s2k_ct_fe_op(S.OP_SQRT, one, None, out32, None), lib.s2k_ct_fe_op(S.OP_MUL, one, None, out32, None), lib.s2k_ct_fe_op(42, one, one, out32, None)):
        assert rc == ERR, rc
    # and the forms that are allowed to leave optional arguments out
    assert lib.s2k_ct_scalar_set_bytes(one, out32, None) == 0 and out32.raw == one
    assert lib.s2k_ct_fe_op(S.OP_INV, one, None, out32, None) == 0 and out32.raw == one


def test_ct_machine_code_has_no_data_dependent_branches(S):
    """The constant-time claim at the level that matters: the COMPILED functions.  Optimisers turn
    `mask = (w == j)` scans back into compare-and-branch chains and secret-indexed loads (clang did, before
    the masks were routed through value_barrier()); this disassembles the object the library was linked
    from and bounds the conditional branches of every secret-handling function by its loop counters and
    argument checks."""
    import re
    import shutil
    import subprocess
    obj = os.path.join(os.path.dirname(S.LIB_PATH), "build", "ct_cpu.o")
    objdump = shutil.which("objdump")
    if not (os.path.exists(obj) and objdump):
        pytest.skip("object file or objdump not available")
    asm = subprocess.run([objdump, "-d", "--no-show-raw-insn", "-C", obj], capture_output=True, text=True).stdout
    counts, fn = {}, None
    for line in asm.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
        if m:
            fn = re.sub(r"\(anonymous namespace\)::", "", m.group(1)).split("(")[0]
            counts.setdefault(fn, 0)
        elif fn and re.search(r"\tj[a-z]+\s", line) and "\tjmp" not in line:
            counts[fn] += 1
    assert "ct_scalar_mult" in counts and "ct_scalar_base_mult" in counts, sorted(counts)
    # allowed: loop back-edges (fixed trip counts), the call_once flag, NULL / range checks of the arguments, the
    # bits of the PUBLIC exponent n - 2 in the scalar inversion
    limits = {"ct_scalar_mult": 2, "ct_scalar_base_mult": 4, "lookup_projective": 1, "make_table": 1, "pt_add": 0, "pt_add_mixed": 0,
              "pt_double": 0, "pt_to_record": 0, "sc_split_glv": 0, "fe_inv": 2, "fe_sqr_n": 1, "sc_inv": 3, "sc_reduce_wide": 0,
              "s2k_ct_scalar_mult": 3, "s2k_ct_scalar_base_mult": 2, "s2k_ct_ecdh": 6, "s2k_ct_ecdsa_sign_raw": 6,
              # loops over the (public) number of terms and over the 32 scalar bytes; allocation / argument checks
              "ct_multi_scalar_mult": 8, "s2k_ct_multi_scalar_mult": 24,
              # single operations: NULL checks, the validity of the operands' ENCODING, the public op code, fixed loops
              "pt_from_record_ct": 2, "fe_sqrt": 2, "s2k_ct_point_add": 4, "s2k_ct_point_double": 3, "s2k_ct_point_subtract": 4,
              "s2k_ct_point_conditional_negate": 4, "s2k_ct_point_negate": 4, "s2k_ct_point_conditional_select": 6,
              "s2k_ct_point_equal": 4, "s2k_ct_point_is_identity": 3, "s2k_ct_point_is_y_odd": 3, "s2k_ct_scalar_op": 16,
              "s2k_ct_scalar_conditional_select": 4, "s2k_ct_scalar_conditional_negate": 3, "s2k_ct_scalar_predicate": 10,
              "s2k_ct_scalar_set_bytes": 3, "s2k_ct_fe_op": 20}
    for name, lim in limits.items():
        assert counts.get(name, 0) <= lim, (name, counts.get(name))
    # nothing that looks like a switch over a 4-bit window value
    for name in ("ct_scalar_mult", "ct_scalar_base_mult", "lookup_projective"):
        assert counts.get(name, 0) < 8
    assert "ct_multi_scalar_mult" in counts or "s2k_ct_multi_scalar_mult" in counts, sorted(counts)

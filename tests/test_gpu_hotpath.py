"""Unit-level parity of the arithmetic the BENCH actually runs: the 9x29
lazy field with its fused products, the Jacobian doubling / mixed addition, the complete 9x29
formulas of the multiscalar kernels, the odd GLV split, the safegcd inversion, and the
verification ladder itself (table, signed-digit recoding, generator additions, worklist
fallback) on arbitrary (u1, u2, P) — all through the C-ABI (s2k_fp_op_batch_ex,
s2k_fn_split_glv_batch_ex, s2k_double_scalar_mult_basepoint_batch_ex), against big integers,
the oracle and the reference's vectors.  Operands are put into LAZY form on the device (same
value, unreduced limbs): the counterpart of the reference tests' random-Z trick
(point_test.go:359-373).  Needs a real MI355X.
"""
import random

import numpy as np
import pytest

import pyref as R
from conftest import load_golden

pytestmark = pytest.mark.gpu
b32 = R.b32
H = bytes.fromhex
P, N = R.P, R.N


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available()
    torch.cuda.init()
    import secp256k1_voi_amd as S
    return S.Engine(0)


def rows(a):
    return [bytes(x) for x in np.asarray(a)]


def ints(a):
    return [int.from_bytes(bytes(x), "big") for x in np.asarray(a)]


def lazy(*codes):
    """4-bit lazy code per operand: bits 1:0 = multiples of p added limb-wise, bit 2 = borrow-spread."""
    v = 0
    for j, c in enumerate(codes):
        v |= c << (4 * j)
    return v


def units(code):
    return 1 + (code & 3) + (1 if code & 4 else 0)


FP_EDGE = [0, 1, 2, 3, P - 1, P - 2, P, P + 1, 2**256 - 1, 2**255, 2**32 + 977, 2**32 + 976, 2**256 - 2**32 - 978,
           (P + 1) // 2, 2**128 - 1, 2**128, 2**224, 0xFFFFFFFF, 0xFFFFFFFF00000000, N, N - 1, N + 1,
           2**29 - 1, 2**29, 2**58 - 1, 2**232, 2**232 - 1, 2**256 - 2**232, 2**261 % P]


def structured_29(rnd, count):
    """Values whose 29-bit limbs (limb 8: 24 bits) are drawn from carry-path patterns."""
    pats = [0, 1, 2**29 - 1, 2**28, 2**28 - 1, 2**29 - 2, 0x1FFFFC2F, 0x1FFFFFF7]
    top = [0, 1, 2**24 - 1, 2**24 - 2, 2**23, 2**23 - 1]
    out = []
    for _ in range(count):
        v = 0
        for i in range(8):
            v |= rnd.choice(pats) << (29 * i)
        v |= rnd.choice(top) << 232
        out.append(v)
    return out


def field_operands(seed, extra=0):
    rnd = random.Random(seed)
    vals = FP_EDGE + structured_29(rnd, 4096) + [rnd.randrange(P - 2**40, 2**256) for _ in range(300)] + \
        [rnd.randrange(2**256) for _ in range(1500 + extra)]
    return vals


def shuffled(vals, seed):
    v = list(vals)
    random.Random(seed).shuffle(v)
    return v


# ---- field: plain and fused products over every lazy-form combination the unit budget allows ----
@pytest.mark.parametrize("codes", [(0, 0), (1, 1), (4, 4), (5, 0), (2, 1), (1, 2), (7, 0), (0, 6), (4, 1), (3, 0)])
def test_fe29_mul_lazy(eng, codes):
    import secp256k1_voi_amd as S
    assert units(codes[0]) * units(codes[1]) <= 7
    a = field_operands(101)
    b = shuffled(a, 102)
    out, _, _ = eng.fp_op_batch_ex(S.HP_MUL, [[b32(v) for v in a], [b32(v) for v in b]], lazy(*codes))
    assert ints(out) == [x * y % P for x, y in zip(a, b)]


@pytest.mark.parametrize("code", [0, 1, 4])
def test_fe29_sqr_lazy(eng, code):
    import secp256k1_voi_amd as S
    assert units(code) ** 2 <= 7
    a = field_operands(103)
    out, _, _ = eng.fp_op_batch_ex(S.HP_SQR, [[b32(v) for v in a]], lazy(code))
    assert ints(out) == [x * x % P for x in a]


@pytest.mark.parametrize("codes", [(0, 0, 0), (1, 1, 2), (4, 4, 2), (1, 4, 2), (0, 0, 6), (2, 1, 0), (0, 5, 1)])
def test_fe29_mul_plus_lazy(eng, codes):
    import secp256k1_voi_amd as S
    assert units(codes[0]) * units(codes[1]) + units(codes[2]) <= 7
    a = field_operands(104)
    b, c = shuffled(a, 105), shuffled(a, 106)
    out, _, _ = eng.fp_op_batch_ex(S.HP_MUL_PLUS, [[b32(v) for v in x] for x in (a, b, c)], lazy(*codes))
    assert ints(out) == [(x * y + z) % P for x, y, z in zip(a, b, c)]


@pytest.mark.parametrize("codes", [(0, 0), (1, 2), (4, 2), (1, 1), (0, 6), (0, 5)])
def test_fe29_sqr_plus_lazy(eng, codes):
    import secp256k1_voi_amd as S
    assert units(codes[0]) ** 2 + units(codes[1]) <= 7
    a = field_operands(107)
    b = shuffled(a, 108)
    out, _, _ = eng.fp_op_batch_ex(S.HP_SQR_PLUS, [[b32(v) for v in x] for x in (a, b)], lazy(*codes))
    assert ints(out) == [(x * x + y) % P for x, y in zip(a, b)]


@pytest.mark.parametrize("codes", [(0, 0, 0, 0), (2, 1, 0, 0), (1, 1, 1, 0), (0, 2, 1, 0), (1, 0, 0, 1), (4, 1, 4, 0),
                                   (1, 4, 0, 2), (5, 0, 1, 0), (0, 0, 4, 1)])
def test_fe29_mul_add_mul_lazy(eng, codes):
    """the shapes the group formulas use: [2]*[1]+[1]*[2] (Jacobian Y3), [1]*[3]+[2]*[1], [3]*[2]+[1]*[1] (pt29 tail)"""
    import secp256k1_voi_amd as S
    assert units(codes[0]) * units(codes[1]) + units(codes[2]) * units(codes[3]) <= 7
    a = field_operands(109)
    b, c, d = shuffled(a, 110), shuffled(a, 111), shuffled(a, 112)
    out, _, _ = eng.fp_op_batch_ex(S.HP_MUL_ADD_MUL, [[b32(v) for v in x] for x in (a, b, c, d)], lazy(*codes))
    assert ints(out) == [(w * x + y * z) % P for w, x, y, z in zip(a, b, c, d)]


@pytest.mark.parametrize("codes", [(0, 0, 0), (1, 1, 0), (0, 2, 1), (4, 1, 0), (2, 0, 4), (1, 0, 1)])
def test_fe29_mul_add_sqr_lazy(eng, codes):
    import secp256k1_voi_amd as S
    assert units(codes[0]) * units(codes[1]) + units(codes[2]) ** 2 <= 7
    a = field_operands(113)
    b, c = shuffled(a, 114), shuffled(a, 115)
    out, _, _ = eng.fp_op_batch_ex(S.HP_MUL_ADD_SQR, [[b32(v) for v in x] for x in (a, b, c)], lazy(*codes))
    assert ints(out) == [(x * y + z * z) % P for x, y, z in zip(a, b, c)]


def test_fe29_linear_and_predicates(eng, oracle):
    import secp256k1_voi_amd as S
    a = field_operands(116)
    b = shuffled(a, 117)
    A, B = [b32(v) for v in a], [b32(v) for v in b]
    for ca, cb in [(0, 0), (3, 2), (7, 1), (4, 4)]:      # units add up to at most 7 (32-bit limbs)
        out, _, _ = eng.fp_op_batch_ex(S.HP_ADD, [A, B], lazy(ca, cb))
        assert ints(out) == [(x + y) % P for x, y in zip(a, b)]
    for code in (0, 1, 4, 7, 3):
        out, _, _ = eng.fp_op_batch_ex(S.HP_NEGATE, [A], lazy(code))
        assert ints(out) == [-x % P for x in a]
        out, _, _ = eng.fp_op_batch_ex(S.HP_HALF, [A], lazy(code))
        assert ints(out) == [x * ((P + 1) // 2) % P for x in a]
        out, _, _ = eng.fp_op_batch_ex(S.HP_NORMALIZE_WEAK, [A], lazy(code))
        assert ints(out) == [x % P for x in a]
        out, _, flag = eng.fp_op_batch_ex(S.HP_NORMALIZE, [A], lazy(code))
        assert ints(out) == [x % P for x in a]
        assert list(flag) == [1 if x % P == 0 else 0 for x in a]
        out, _, _ = eng.fp_op_batch_ex(S.HP_MUL_SMALL21, [A], lazy(code))
        assert ints(out) == [21 * x % P for x in a]
    # equality: values equal mod p in different representations, and near misses
    eq_b = [(x % P) if i % 3 else ((x + 1) % 2**256) for i, x in enumerate(a)]
    for ca in (0, 1, 4, 5):
        _, _, flag = eng.fp_op_batch_ex(S.HP_EQ, [A, [b32(v) for v in eq_b]], lazy(ca, 0))
        assert list(flag) == [1 if (x - y) % P == 0 else 0 for x, y in zip(a, eq_b)]
    # conditional negation of 1-unit values (table y-coordinates)
    out, _, _ = eng.fp_op_batch_ex(S.HP_COND_NEGATE1, [A, B], 0)
    assert ints(out) == [(-x if y & 1 else x) % P for x, y in zip(a, b)]
    # inversion and square root on the 9x29 field (Element.Invert: 0 -> 0)
    sub = a[:len(FP_EDGE) + 600]
    out, _, _ = eng.fp_op_batch_ex(S.HP_INV, [[b32(v) for v in sub]], 0)
    assert rows(out) == [oracle.fp_inv(b32(x % P)) for x in sub]
    # the safegcd inversion mod p (per-key tables) gives the same values, lazy inputs included
    for code in (0, 1, 4):
        out, _, _ = eng.fp_op_batch_ex(S.HP_INV_GCD, [[b32(v) for v in sub]], lazy(code))
        assert rows(out) == [oracle.fp_inv(b32(x % P)) for x in sub]
    sq = [x * x % P for x in sub[:300]] + sub[300:]
    out, _, flag = eng.fp_op_batch_ex(S.HP_SQRT, [[b32(v) for v in sq]], 0)
    for x, o, f in zip(sq, ints(out), flag):
        root = R.sqrt_p(x % P)
        assert bool(f) == (root is not None)
        assert o * o % P == x % P if f else o == 0


# ---- group formulas with random Z and lazy coordinates ----
def curve_points(rnd, count):
    pts = []
    base = R.mul(rnd.randrange(1, N), R.G)
    for _ in range(count):
        pts.append(base)
        base = R.add(base, R.mul(rnd.randrange(1, 1 << 40), R.G))
    return pts


def wycheproof_points(oracle):
    d = load_golden("wycheproof_ecdh.json")
    return [R.dec65(oracle.point_from_bytes(H(c["point"]))) for c in d["cases"]]


@pytest.mark.parametrize("codes", [(0, 0, 0, 0, 0), (1, 1, 4, 0, 1), (4, 0, 1, 0, 0), (0, 4, 4, 0, 1), (0, 1, 8, 0, 1),
                                   (0, 4, 8, 0, 4)])
def test_jacobian_double_and_add_random_z(eng, oracle, codes):
    """jpt29_double / jpt29_add_affine on P lifted to a random Z (x = X/Z^2, y = Y/Z^3), including every
    exceptional input of the incomplete addition (P + P, P - P): those must come back with Z3 = 0 (flag 0)
    — that is what sends a verification lane to the complete kernel."""
    import secp256k1_voi_amd as S
    rnd = random.Random(118)
    pts = wycheproof_points(oracle)[:200] + curve_points(rnd, 600)
    qs = shuffled(pts, 119)
    for i in range(0, len(pts), 7):      # exceptional pairs
        qs[i] = pts[i] if (i // 7) % 2 == 0 else R.neg(pts[i])
    z = [rnd.randrange(1, 2**256) for _ in pts]
    z[:4] = [1, P - 1, 2**256 - 1, 2]
    cols = [[b32(p[0]) for p in pts], [b32(p[1]) for p in pts], [b32(v) for v in z],
            [b32(q[0]) for q in qs], [b32(q[1]) for q in qs]]
    x, y, flag = eng.fp_op_batch_ex(S.HP_JDBL, cols[:3], lazy(*codes[:3]))
    assert flag.all()
    assert list(zip(ints(x), ints(y))) == [R.add(p, p) for p in pts]
    for op in (S.HP_JADD, S.HP_JADD_FULL):      # mixed addition; Jacobian + Jacobian (jpt29_add, Q at Z2 = Z^2)
        x, y, flag = eng.fp_op_batch_ex(op, cols, lazy(*codes))
        for p, q, xi, yi, f in zip(pts, qs, ints(x), ints(y), flag):
            if p[0] == q[0]:
                assert f == 0
            else:
                assert f == 1 and (xi, yi) == R.add(p, q)


@pytest.mark.parametrize("codes", [(0, 0, 0, 0, 0), (0, 0, 4, 0, 0), (0, 0, 1, 0, 0)])
def test_pt29_complete_formulas_random_z(eng, oracle, codes):
    """pt29_add / pt29_add_mixed / pt29_double (the multiscalar kernels' group law): no exceptions —
    P + P, P - P, identity + Q all give the reference's result."""
    import secp256k1_voi_amd as S
    rnd = random.Random(120)
    pts = wycheproof_points(oracle)[:200] + curve_points(rnd, 600)
    qs = shuffled(pts, 121)
    for i in range(0, len(pts), 5):
        qs[i] = pts[i] if (i // 5) % 2 == 0 else R.neg(pts[i])
    z = [rnd.randrange(1, P) for _ in pts]
    for i in range(3, len(pts), 11):
        z[i] = 0                       # P = identity
    cols = [[b32(p[0]) for p in pts], [b32(p[1]) for p in pts], [b32(v) for v in z],
            [b32(q[0]) for q in qs], [b32(q[1]) for q in qs]]

    def expect(op):
        out = []
        for p, q, zz in zip(pts, qs, z):
            pp = None if zz == 0 else p
            out.append(R.add(pp, pp) if op == "dbl" else R.add(pp, q))
        return out

    def check(x, y, flag, exp):
        for xi, yi, f, e in zip(ints(x), ints(y), flag, exp):
            if e is None:
                assert f == 0
            else:
                assert f == 1 and (xi, yi) == e
    check(*eng.fp_op_batch_ex(S.HP_PT29_DBL, cols[:3], lazy(*codes[:3])), expect("dbl"))
    check(*eng.fp_op_batch_ex(S.HP_PT29_ADD, cols, lazy(*codes)), expect("add"))
    check(*eng.fp_op_batch_ex(S.HP_PT29_ADD_MIXED, cols, lazy(*codes)), expect("add"))


@pytest.mark.parametrize("codes", [(0, 0, 0, 0, 0), (0, 0, 4, 0, 0), (0, 0, 1, 0, 0)])
def test_pt29q_quad_formulas(eng, oracle, codes):
    """pt29q_add / pt29q_double (pt29q.h: the complete formulas spread over the four lanes of a quad, the serial tail of
    the multi-scalar multiplication): same cases as the single-lane forms - P + P, P - P, identity + Q - and chained
    on their own outputs (P + 5 Q, 2^7 P), against big-integer arithmetic."""
    import secp256k1_voi_amd as S
    rnd = random.Random(130)
    pts = wycheproof_points(oracle)[:200] + curve_points(rnd, 600)
    qs = shuffled(pts, 131)
    for i in range(0, len(pts), 5):
        qs[i] = pts[i] if (i // 5) % 2 == 0 else R.neg(pts[i])
    z = [rnd.randrange(1, P) for _ in pts]
    for i in range(3, len(pts), 11):
        z[i] = 0                       # P = identity
    cols = [[b32(p[0]) for p in pts], [b32(p[1]) for p in pts], [b32(v) for v in z],
            [b32(q[0]) for q in qs], [b32(q[1]) for q in qs]]

    def expect(op, reps):
        out = []
        for p, q, zz in zip(pts, qs, z):
            r = None if zz == 0 else p
            for _ in range(reps):
                r = R.add(r, r) if op == "dbl" else R.add(r, q)
            out.append(r)
        return out

    def check(x, y, flag, exp):
        for xi, yi, f, e in zip(ints(x), ints(y), flag, exp):
            if e is None:
                assert f == 0
            else:
                assert f == 1 and (xi, yi) == e
    for reps in (1, 5):
        check(*eng.fp_op_batch_ex(S.HP_PT29Q_ADD, cols, lazy(*codes) | reps << 20), expect("add", reps))
    for reps in (1, 7):
        check(*eng.fp_op_batch_ex(S.HP_PT29Q_DBL, cols, lazy(*codes) | reps << 20), expect("dbl", reps))


@pytest.mark.parametrize("codes", [(0, 0, 0, 0, 0), (0, 0, 4, 0, 0), (1, 0, 1, 0, 0)])
def test_xyzz_mixed_addition_random_zz(eng, oracle, codes):
    """xyzz29_add_affine (the bucket pass's incomplete mixed addition, x = X/ZZ, y = Y/ZZZ) on P lifted to a random ZZ = c^2,
    ZZZ = c^3, through the projective form the bucket pass stores: exact for every generic pair, and EXACTLY the exceptional
    inputs (P + P, P - P: same x) come back with ZZ3 = 0 (flag 0) - that is what sends a piece of a bucket to the
    complete-formula redo kernel."""
    import secp256k1_voi_amd as S
    rnd = random.Random(140)
    pts = wycheproof_points(oracle)[:200] + curve_points(rnd, 600)
    qs = shuffled(pts, 141)
    for i in range(0, len(pts), 7):      # exceptional pairs
        qs[i] = pts[i] if (i // 7) % 2 == 0 else R.neg(pts[i])
    z = [rnd.randrange(1, 2**256) for _ in pts]
    z[:4] = [1, P - 1, 2**256 - 1, 2]
    cols = [[b32(p[0]) for p in pts], [b32(p[1]) for p in pts], [b32(v) for v in z],
            [b32(q[0]) for q in qs], [b32(q[1]) for q in qs]]
    x, y, flag = eng.fp_op_batch_ex(S.HP_XYZZ_ADD, cols, lazy(*codes))
    for p, q, xi, yi, f in zip(pts, qs, ints(x), ints(y), flag):
        if p[0] == q[0]:
            assert f == 0
        else:
            assert f == 1 and (xi, yi) == R.add(p, q)


@pytest.mark.parametrize("codes", [(0, 0, 0, 0, 0), (0, 0, 4, 0, 0), (1, 0, 1, 0, 0)])
def test_xyzz_jacobian_round_border(eng, oracle, codes):
    """The keyed ladder's round border as the kernel runs it: XYZZ addition, XYZZ -> Jacobian (X ZZ, Y ZZZ, ZZ), two Jacobian
    doublings, Jacobian -> XYZZ (X, Y, Z^2, Z^3), XYZZ addition, XYZZ -> Jacobian: 4 P + 5 Q for every generic pair, and Z = 0
    at the end for exactly the pairs whose FIRST addition is exceptional (P = +-Q): ZZ = 0 has to survive both changes of
    form, the doublings and the second addition, because that is what sends the signature to the complete formulas."""
    import secp256k1_voi_amd as S
    rnd = random.Random(150)
    pts = wycheproof_points(oracle)[:150] + curve_points(rnd, 500)
    qs = shuffled(pts, 151)
    for i in range(0, len(pts), 9):
        qs[i] = pts[i] if (i // 9) % 2 == 0 else R.neg(pts[i])
    z = [rnd.randrange(1, 2**256) for _ in pts]
    z[:4] = [1, P - 1, 2**256 - 1, 2]
    cols = [[b32(p[0]) for p in pts], [b32(p[1]) for p in pts], [b32(v) for v in z],
            [b32(q[0]) for q in qs], [b32(q[1]) for q in qs]]
    x, y, flag = eng.fp_op_batch_ex(S.HP_XYZZ_ROUND, cols, lazy(*codes))
    for p, q, xi, yi, f in zip(pts, qs, ints(x), ints(y), flag):
        if p[0] == q[0]:
            assert f == 0
        else:
            s = R.add(p, q)
            s4 = R.add(R.add(s, s), R.add(s, s))
            assert f == 1 and (xi, yi) == R.add(s4, q)


# ---- scalars: odd GLV split and safegcd inversion ----
def test_split_glv_odd(eng):
    """sc_split_glv_odd (hot path): k == +-k1 +- k2*lambda (mod n), k1 and k2 odd and below 2^129, for the
    reference's 20 boundary scalars (point_mul_glv_test.go:25-45) and random ones."""
    rnd = random.Random(122)
    g = load_golden("kats.json")["glv"]
    ks = [int(s, 16) for s in g["boundary_scalars"]] + [0, 1, 2, 3, N - 1, N - 2, N, N + 1, 2**256 - 1, R.LAMBDA, N - R.LAMBDA,
                                                        2**128, 2**128 - 1, 2**127, 2**129] + [rnd.randrange(2**256) for _ in range(20000)]
    k1, k2, sg = eng.fn_split_glv_odd_batch([b32(v) for v in ks])
    for v, a, b, s in zip(ks, ints(k1), ints(k2), sg):
        assert a & 1 and b & 1 and a < 2**129 and b < 2**129
        a = -a if s & 1 else a
        b = -b if s & 2 else b
        assert (a + b * R.LAMBDA - v) % N == 0, hex(v)


def test_modinv_one_million(eng):
    """Scalar.Invert through the safegcd division steps (modinv30.h) on 2^20 random + structured inputs;
    every result compared with Python's modular inverse (0 -> 0, scalar_invert.go:11)."""
    import secp256k1_voi_amd as S
    rng = np.random.default_rng(123)
    n = 1 << 20
    a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    rnd = random.Random(124)
    structured = [0, 1, 2, N - 1, N - 2, N, N + 1, 2**256 - 1, (N + 1) // 2, 2**255, 2**30, 2**30 - 1, 2**60, 2**240 - 1] + \
        [1 << i for i in range(256)] + [(1 << i) - 1 for i in range(1, 257)] + [N - (1 << i) for i in range(250)] + \
        [sum(rnd.choice([0, 2**30 - 1, 2**29, 1]) << (30 * i) for i in range(9)) % 2**256 for _ in range(4096)]
    for i, v in enumerate(structured):
        a[i] = np.frombuffer(b32(v), np.uint8)
    out, _ = eng.fn_op_batch(S.OP_INV, a)
    raw = a.tobytes()
    got = out.tobytes()
    for i in range(n):
        v = int.from_bytes(raw[32 * i:32 * i + 32], "big") % N
        w = int.from_bytes(got[32 * i:32 * i + 32], "big")
        assert w == (pow(v, -1, N) if v else 0), i


# ---- the verification ladder on arbitrary inputs, both implementations ----
@pytest.mark.parametrize("impl", ["fast", "complete"])
def test_ladder_kats_both_implementations(eng, oracle, impl):
    """libsecp256k1 KAT (point_test.go:242-261), every Wycheproof ECDH point/scalar, the GLV boundary
    scalars and edge scalars through the fast ladder (k_verify_fast<POINT>) and the complete path."""
    import secp256k1_voi_amd as S
    sel = S.IMPL_FAST if impl == "fast" else S.IMPL_COMPLETE
    kat = load_golden("kats.json")["libsecp256k1_ecmult_const"]
    out = eng.double_scalar_mult_basepoint_batch_ex(sel, None, [H(kat["xn"])], [H(kat["a"])])
    assert rows(out)[0].hex() == kat["b"]
    d = load_golden("wycheproof_ecdh.json")
    pts = [oracle.point_from_bytes(H(c["point"])) for c in d["cases"]]
    out = eng.double_scalar_mult_basepoint_batch_ex(sel, None, [H(c["private"]) for c in d["cases"]], pts)
    for c, o in zip(d["cases"], rows(out)):
        assert o[1:33] == H(c["shared"]), c["tcId"]
    rnd = random.Random(125)
    g = load_golden("kats.json")["glv"]
    ks = [0, 1, 2, 3, N - 1, N - 2, N, N + 1, 2**256 - 1, R.LAMBDA, N - R.LAMBDA, 2**128, 2**128 - 1, 2**127, 15, 16, 17] + \
         [int(s, 16) for s in g["boundary_scalars"]] + [rnd.randrange(2**256) for _ in range(400)]
    Pt = [R.enc65(R.mul(rnd.randrange(1, N), R.G)) for _ in ks]
    Pt[5] = bytes(65)                      # identity inputs
    Pt[40] = bytes(65)
    out = eng.double_scalar_mult_basepoint_batch_ex(sel, None, [b32(v) for v in ks], Pt)
    for v, p, o in zip(ks, Pt, rows(out)):
        assert o == oracle.scalar_mult_trivial(b32(v % N), p), hex(v)
    u1 = [0, 1, N - 1, 2**256 - 1] + [rnd.randrange(2**256) for _ in ks[4:]]
    out = eng.double_scalar_mult_basepoint_batch_ex(sel, [b32(v) for v in u1], [b32(v) for v in ks], Pt)
    for a, v, p, o in zip(u1, ks, Pt, rows(out)):
        assert o == oracle.double_scalar_mult_basepoint_vartime(b32(a % N), b32(v % N), p)


@pytest.mark.parametrize("impl", ["fast", "complete"])
def test_ladder_exceptional_sums(eng, oracle, impl):
    """inputs that drive the ladder into its exceptional cases: u1*G + u2*P = identity, P = +-G with
    colliding table entries, small multiples (the accumulator meets a table entry), u2 = 0."""
    import secp256k1_voi_amd as S
    sel = S.IMPL_FAST if impl == "fast" else S.IMPL_COMPLETE
    rnd = random.Random(126)
    u1s, u2s, pts = [], [], []
    for _ in range(64):                    # u1 G + u2 (dG) = identity
        d, u2 = rnd.randrange(1, N), rnd.randrange(1, N)
        u1s.append((-u2 * d) % N); u2s.append(u2); pts.append(R.enc65(R.mul(d, R.G)))
    for k in list(range(0, 40)) + [N - k for k in range(1, 40)]:   # P = G: u1 G + k G collides inside the generator part
        u1s.append(rnd.randrange(N)); u2s.append(k); pts.append(R.enc65(R.G))
        u1s.append(k); u2s.append((N - k) % N); pts.append(R.enc65(R.G))            # k G - k G
        u1s.append(0); u2s.append(k); pts.append(R.enc65(R.neg(R.G)))
    for j in range(1, 33):                 # P = 2^(22 j) G: collides with the generator table's window bases
        u1s.append(1 << (22 * (j % 11))); u2s.append(1); pts.append(R.enc65(R.mul(1 << (22 * (j % 11)), R.G)))
    out = eng.double_scalar_mult_basepoint_batch_ex(sel, [b32(v) for v in u1s], [b32(v) for v in u2s], pts)
    for a, v, p, o in zip(u1s, u2s, pts, rows(out)):
        assert o == oracle.double_scalar_mult_basepoint_vartime(b32(a), b32(v), p), (hex(a), hex(v))


def test_malformed_point_records_are_errors(eng):
    """A record that is neither 0x04||X||Y on the curve nor the identity record is S2K_ERR_ARG for every
    group entry point (the reference cannot construct such Points: point_s11n.go:178-209), as for the MSM."""
    import secp256k1_voi_amd as S
    good = R.enc65(R.G)
    off = bytearray(good); off[64] ^= 1
    bad_prefix = bytearray(good); bad_prefix[0] = 0x05
    noncanon = b"\x04" + b32(P) + b32(7)
    one = [b32(1)]
    for rec in (bytes(off), bytes(bad_prefix), noncanon):
        for impl in (S.IMPL_FAST, S.IMPL_COMPLETE):
            with pytest.raises(S.EngineError):
                eng.double_scalar_mult_basepoint_batch_ex(impl, one, one, [rec])
        with pytest.raises(S.EngineError):
            eng.scalar_mult_batch(one, [rec])
        with pytest.raises(S.EngineError):
            eng.point_add_batch([good], [rec])
        with pytest.raises(S.EngineError):
            eng.point_double_batch([rec])
    assert rows(eng.point_add_batch([good], [bytes(65)]))[0] == good

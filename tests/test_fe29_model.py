"""CPU model of the 9x29 lazy field (secp256k1_voi_amd/csrc/fe29.h, tools/gen_fe29_mul.py) and of
the formulas built on it (jacobian29.h, the table build of k_verify_fast).

Two layers:
  * integers: the multiplication schedule, its tail, normalisation, halving and negation are
    executed limb by limb on random and extremal lazy inputs and compared with arithmetic mod p
    (with every 32-/64-bit width the device code relies on asserted);
  * intervals: upper bounds of every limb are pushed through the exact operation sequences of
    jacobian29.h to show that no 32-bit limb, no 64-bit column sum and no negate() bias
    overflows, and that the invariants (x 1 unit, y <= 2, z 1) are closed.

This is a model of the device code's arithmetic structure, not of its instructions; the GPU
parity tests check the values the kernels actually produce.
"""
import random

P = 2**256 - 2**32 - 977
L, W = 9, 29
M = (1 << W) - 1
M8 = (1 << 24) - 1
R0, R1 = 0x7A20, 0x100
P_LIMBS = [0x1FFFFC2F, 0x1FFFFFF7] + [M] * 6 + [M8]
U64 = 1 << 64
U32 = 1 << 32


def value(n):
    return sum(x << (W * i) for i, x in enumerate(n))


def test_constants():
    assert value(P_LIMBS) == P
    assert (1 << 261) % P == R1 * (1 << W) + R0
    assert (1 << 256) % P == 0x3D1 + 8 * (1 << W)


# ---- integer layer ------------------------------------------------------------------------------
def mul_tail(t, c, d):
    assert d < U32
    c += t[8] + d * R0
    assert c < U64
    r8 = c & M8
    c >>= 24
    assert c < U32
    k0, k1 = R0 >> 5, R1 >> 5
    e = t[0] + c * k0 + d * (k0 << 13)
    assert e < U64
    r0 = e & M
    e >>= W
    e += t[1] + c * k1 + d * (k1 << 13)
    assert e < U64
    r1 = e & M
    e >>= W
    r2 = t[2] + e
    assert r2 < U32
    return [r0, r1, r2] + t[3:8] + [r8]


def mulsum_int(pairs, addend=None):
    """sum of products (+ a lazy addend) with one reduction, as fe29_mul / fe29_sqr /
    fe29_mul_add_* / fe29_*_plus"""
    col = [0] * (2 * L - 1)
    if addend:
        for i in range(L):
            col[i] += addend[i]
    for a, b in pairs:
        for i in range(L):
            for j in range(L):
                col[i + j] += a[i] * b[j]
    t = [0] * L
    d = col[L - 1]
    assert d < U64
    t[L - 1] = d & M
    d >>= W
    c = 0
    uprev = None
    for k in range(L - 1):
        d += col[L + k]
        assert d < U64
        u = d & M
        d >>= W
        c += col[k] + u * R0 + (uprev * R1 if uprev is not None else 0)
        assert c < U64
        t[k] = c & M
        c >>= W
        uprev = u
    c += uprev * R1
    return mul_tail(t, c, d)


def negate_int(a, w):
    k = w + 1
    r = [P_LIMBS[i] * k - a[i] for i in range(L)]
    assert all(0 <= x < U32 for x in r)
    return r


def half_int(a):
    odd = a[0] & 1
    t = [a[i] + (P_LIMBS[i] if odd else 0) for i in range(L)]
    assert all(x < U32 for x in t)
    return [(t[i] >> 1) + ((t[i + 1] & 1) << 28) for i in range(L - 1)] + [t[8] >> 1]


def normalize_weak_int(a):
    t = list(a)
    x = t[8] >> 24
    t[8] &= M8
    t[0] += x * 0x3D1
    t[1] += x << 3
    assert t[0] < U32 and t[1] < U32
    for i in range(L - 1):
        t[i + 1] += t[i] >> W
        assert t[i + 1] < U32
        t[i] &= M
    return t


def normalize_int(a):
    r = normalize_weak_int(a)
    x = r[8] >> 24
    m = M
    for i in range(2, 8):
        m &= r[i]
    ge = int(r[8] == M8 and m == M and (r[1] + 8 + ((r[0] + 0x3D1) >> W)) > M)
    x |= ge
    t = list(r)
    t[0] += x * 0x3D1
    t[1] += x << 3
    for i in range(L - 1):
        t[i + 1] += t[i] >> W
        t[i] &= M
    t[8] &= M8
    return t


def from_int(v):
    return [(v >> (W * i)) & M for i in range(L - 1)] + [v >> (W * (L - 1))]


def rand_lazy(rng, units, extremal=False):
    """limbs anywhere up to `units` (floats allowed), as the device's lazy values may be"""
    hi = [int(units * (1 << W))] * 8 + [int(units * ((1 << 24) + 16))]
    if extremal:
        return [h if rng.random() < 0.7 else rng.randrange(h + 1) for h in hi]
    return [rng.randrange(h + 1) for h in hi]


def test_schedule_matches_arithmetic():
    rng = random.Random(29)
    cases = [(1, 1), (2, 2), (2, 3.9), (1, 7.8), (2.79, 2.79)]
    for wa, wb in cases:
        for it in range(200):
            a, b = rand_lazy(rng, wa, it % 2 == 0), rand_lazy(rng, wb, it % 3 == 0)
            r = mulsum_int([(a, b)])
            assert value(r) % P == value(a) * value(b) % P
            assert all(x <= M for i, x in enumerate(r) if i not in (2, 8)) and r[8] <= M8 and r[2] < M + (1 << 20)
    for it in range(200):   # fused forms at their budget: 2*2 + 1*1 (doubling), 2*1 + 1*2 (addition), 7 at most
        a, b, c, d = rand_lazy(rng, 2, True), rand_lazy(rng, 2, True), rand_lazy(rng, 1, True), rand_lazy(rng, 3.8, True)
        r = mulsum_int([(a, b), (c, d)])
        assert value(r) % P == (value(a) * value(b) + value(c) * value(d)) % P
        r = mulsum_int([(a, b), (c, c)])
        assert value(r) % P == (value(a) * value(b) + value(c) ** 2) % P
        e = rand_lazy(rng, 7.9, True)                        # *_plus: any addend whose limbs fit 32 bits
        r = mulsum_int([(a, b)], e)
        assert value(r) % P == (value(a) * value(b) + value(e)) % P
        assert all(x <= M for i, x in enumerate(r) if i not in (2, 8)) and r[8] <= M8 and r[2] < M + (1 << 20)


def test_linear_ops_match_arithmetic():
    rng = random.Random(7)
    for it in range(500):
        w = rng.choice([1, 2, 3, 4, 6])
        a = rand_lazy(rng, w, it % 2 == 0)
        assert value(negate_int(a, w)) % P == -value(a) % P
        assert value(half_int(a)) * 2 % P == value(a) % P
        nw = normalize_weak_int(a)
        assert value(nw) % P == value(a) % P and all(x <= M for x in nw[:8]) and nw[8] <= M8 + 16
        n = normalize_int(a)
        assert value(n) == value(a) % P and all(x <= M for x in n[:8]) and n[8] <= M8
    for v in (0, 1, P - 1, P, P + 1, 2 * P - 1, 2 * P, 2**256 - 1, 2**256, 2**256 + 2**32 + 976, 3 * P + 5):
        assert value(normalize_int(from_int(v))) == v % P


# ---- interval layer ------------------------------------------------------------------------------
class B:
    """upper bounds of the limbs (lower bound 0)"""

    def __init__(self, hi):
        self.hi = list(hi)
        assert len(self.hi) == L and all(h < U32 for h in self.hi), "32-bit limb overflow"

    @staticmethod
    def units(w):
        return B([int(w * (1 << W))] * 8 + [int(w * ((1 << 24) + 16))])

    def within(self, w):
        return all(a <= b for a, b in zip(self.hi, B.units(w).hi))


def add(a, b):
    return B([x + y for x, y in zip(a.hi, b.hi)])


def negate(a, w):
    bias = [p * (w + 1) for p in P_LIMBS]
    assert all(x <= b for x, b in zip(a.hi, bias)), "negate bias too small"
    return B(bias)


def mul_int(a, k):
    return B([x * k for x in a.hi])


def half(a):
    t = [a.hi[i] + P_LIMBS[i] for i in range(L)]
    assert all(x < U32 for x in t)
    return B([(t[i] >> 1) + (1 << 28) for i in range(L - 1)] + [t[8] >> 1])


def normalize_weak(a):
    x = a.hi[8] >> 24
    assert a.hi[0] + x * 0x3D1 < U32 and a.hi[1] + (x << 3) + 8 < U32
    assert all(h + 8 < U32 for h in a.hi)
    return B([M] * 8 + [M8 + 8])


def mulsum(pairs, addend=None):
    col = [0] * (2 * L - 1)
    if addend:
        for i in range(L):
            col[i] += addend.hi[i]
    for a, b in pairs:
        for i in range(L):
            for j in range(L):
                col[i + j] += a.hi[i] * b.hi[j]
    d = col[L - 1]
    assert d < U64, "column 8"
    d >>= W
    c = 0
    for k in range(L - 1):
        d += col[L + k]
        assert d < U64, ("high column", k)
        d >>= W
        c += col[k] + M * R0 + (M * R1 if k else 0)
        assert c < U64, ("low column", k)
        c >>= W
    c += M * R1
    assert d < U32, "d32 overflow"
    c += M + d * R0
    assert c < U64
    c >>= 24
    assert c < U32, "clo overflow"
    k0, k1 = R0 >> 5, R1 >> 5
    e = M + c * k0 + d * (k0 << 13)
    assert e < U64
    e >>= W
    e += M + c * k1 + d * (k1 << 13)
    assert e < U64
    e >>= W
    out = B([M, M, M + e] + [M] * 5 + [M8])
    assert out.within(1.001), "product is not 1 unit"
    return out


def mul(a, b):
    return mulsum([(a, b)])


def sqr(a):
    return mulsum([(a, a)])


ONE_UNIT = mulsum([(B.units(1), B.units(1))])   # what a product looks like


def jpt_double(x, y, z):
    z3 = mul(y, z)
    s = sqr(y)
    l = half(mul_int(sqr(x), 3))
    t = mul(negate(s, 1), x)
    x3 = mulsum([(l, l)], add(t, t))
    t = add(t, x3)
    y3 = negate(mulsum([(t, l), (s, s)]), 1)
    return x3, y3, z3


def jpt_add_affine(x, y, z, bx, by):
    zz = sqr(z)
    nx = negate(x, 1)
    h = mulsum([(bx, zz)], nx)
    ns = negate(mul(by, zz), 1)
    i = mulsum([(ns, z)], y)
    z3 = mul(z, h)
    h2 = sqr(h)
    h3 = mul(h2, negate(h, 1))
    t = mul(nx, h2)
    x3 = mulsum([(i, i)], add(add(h3, t), t))
    t = add(t, x3)
    y3 = mulsum([(t, i), (h3, y)])
    return x3, y3, z3, h


def test_jacobian_invariant_closed():
    nw = normalize_weak(B.units(4))
    X, Y, Z = nw, B.units(2), ONE_UNIT
    bx, by = ONE_UNIT, negate(ONE_UNIT, 1)           # table entry; y after a conditional negate
    assert by.within(2)
    x3, y3, z3 = jpt_double(X, Y, Z)
    assert x3.within(1.001) and y3.within(2) and z3.within(1.001)
    x3, y3, z3, h = jpt_add_affine(X, Y, Z, bx, by)
    assert x3.within(1.001) and y3.within(1.001) and z3.within(1.001) and h.within(1.001)
    # a weakly normalised value is a legal x / h / table entry everywhere a product result is
    x3, y3, z3, h = jpt_add_affine(nw, Y, nw, nw, B.units(2))
    assert x3.within(1.001) and y3.within(2)
    x3, y3, z3 = jpt_double(nw, Y, nw)
    assert x3.within(1.001) and y3.within(2)


def test_table_build_bounds():
    # k_verify_fast: d = 2Q from affine Q, entries rescaled by products of the stored H values
    q = B([M] * 8 + [M8])                               # fe29_from_words output
    dx, dy, dz = jpt_double(q, q, q)
    c2 = sqr(dz)
    c3 = mul(c2, dz)
    cur = (mul(q, c2), mul(q, c3), q)
    assert dx.within(1.001) and dy.within(2)            # used as the affine addend as they are
    x, y, z, h = jpt_add_affine(*cur, dx, dy)
    rr = mul(ONE_UNIT, h)
    r2 = sqr(rr)
    r3 = mul(r2, rr)
    assert mul(x, r2).within(1.001) and mul(y, r3).within(1.001)
    # final comparison: x(R) against r * Z^2, operands of fe29_eq
    s = add(x, negate(mul(q, sqr(z)), 1))
    normalize_weak(s)


# ---- pt29.h --------------------------------------------------------------------------------------
def mul_small_norm(a, k):
    c = 0
    for i in range(8):
        c += a.hi[i] * k
        assert c < U64
        c >>= W
    c += a.hi[8] * k
    x = c >> 24
    assert x < U32
    out = B([M + x * 0x3D1, M + (x << 3)] + [M] * 6 + [M8])
    assert out.within(1.1)
    return out


def triple_norm(a):
    return normalize_weak(mul_int(a, 3))


def pt_add_tail(t0, t1, t2, t3, t4, y3):
    t0n = triple_norm(t0)
    z3 = add(t1, t2)
    t1m = add(t1, negate(t2, 1))
    return (mulsum([(t3, t1m), (negate(t4, 1), y3)]), mulsum([(t1m, z3), (y3, t0n)]), mulsum([(z3, t4), (t0n, t3)]))


def pt_add_mixed(p, qx, qy):
    px, py, pz = p
    t0, t1 = mul(px, qx), mul(py, qy)
    t3 = mulsum([(add(qx, qy), add(px, py))], negate(add(t0, t1), 2))
    t4 = mulsum([(qy, pz)], py)
    y3 = mul_small_norm(mulsum([(qx, pz)], px), 21)
    t2 = mul_small_norm(pz, 21)
    return pt_add_tail(t0, t1, t2, t3, t4, y3)


def pt_add(p, q):
    (px, py, pz), (qx, qy, qz) = p, q
    t0, t1, t2 = mul(px, qx), mul(py, qy), mul(pz, qz)
    t3 = mulsum([(add(px, py), add(qx, qy))], negate(add(t0, t1), 2))
    t4 = mulsum([(add(py, pz), add(qy, qz))], negate(add(t1, t2), 2))
    y3 = mulsum([(add(px, pz), add(qx, qz))], negate(add(t0, t2), 2))
    return pt_add_tail(t0, t1, mul_small_norm(t2, 21), t3, t4, mul_small_norm(y3, 21))


def pt_double(p):
    px, py, pz = p
    t0 = sqr(py)
    z3 = mul_int(normalize_weak(mul_int(t0, 4)), 2)
    t1 = mul(py, pz)
    zz = sqr(pz)
    t2 = mul_small_norm(zz, 21)
    y3 = add(t0, t2)
    t0m = normalize_weak(add(t0, negate(mul_small_norm(zz, 63), 1)))
    return mul(mul_int(t0m, 2), mul(px, py)), mulsum([(t2, z3), (t0m, y3)]), mul(t1, z3)


def test_projective_invariant_closed():
    # coordinates as products leave them (limb 2 a little above 2^29), affine inputs from words
    inv = B([M, M, M + (1 << 16)] + [M] * 5 + [M8])      # the invariant, with room for limb 2's carry
    P = (inv, inv, inv)
    q = B([M] * 8 + [M8])
    for r in (pt_add(P, P), pt_double(P), pt_add_mixed(P, q, q)):
        assert all(a <= b for c in r for a, b in zip(c.hi, inv.hi))


def test_model_rejects_an_overflow():
    import pytest
    with pytest.raises(AssertionError):
        mul(B.units(3), B.units(3))
    with pytest.raises(AssertionError):
        negate(B.units(3), 1)

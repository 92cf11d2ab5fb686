"""Re-derives the limb bounds of the lazy 10x26 field (secp256k1_voi_amd/csrc/fe26.h) through
the exact operation sequences of pt26.h (the complete formulas of the multiscalar kernels), with
interval arithmetic on the limbs (tests/test_fe29_model.py does the same for the 9x29 field
of the verification ladder):

  * every 32-bit limb stays below 2^32,
  * every negate() bias dominates its operand limb by limb,
  * every 64-bit column sum of fe26_mul / fe26_sqr / the fused multiply-adds stays below 2^64,
    including the fold terms, and the tail quantities fit the widths the code assumes,
  * the loop invariant (coordinates <= 3 for projective points) is closed.

This is a CPU model of the device code's arithmetic structure, not of its instructions; the
GPU parity tests check the values.
"""
M = (1 << 26) - 1
M9 = (1 << 22) - 1
P_LIMBS = [0x3FFFC2F, 0x3FFFFBF] + [0x3FFFFFF] * 7 + [0x03FFFFF]
R0, R1 = 0x3D10, 0x400


class B:
    """upper bounds of the 10 limbs (lower bound is 0)"""

    def __init__(self, hi):
        self.hi = list(hi)
        assert len(self.hi) == 10
        assert all(h < (1 << 32) for h in self.hi), "32-bit limb overflow"

    @staticmethod
    def mag(m):
        return B([2 * m * M] * 9 + [2 * m * M9])

    def within(self, m):
        lim = B.mag(m)
        return all(a <= b for a, b in zip(self.hi, lim.hi))


def add(a, b):
    return B([x + y for x, y in zip(a.hi, b.hi)])


def negate(a, m):
    bias = [p * 2 * (m + 1) for p in P_LIMBS]
    assert all(x <= b for x, b in zip(a.hi, bias)), "negate bias too small"
    return B(bias)          # bias - a, a >= 0


def mul_int(a, k):
    return B([x * k for x in a.hi])


def half(a):
    t = [a.hi[0] + P_LIMBS[0], a.hi[1] + P_LIMBS[1]] + [a.hi[i] + M for i in range(2, 9)] + [a.hi[9] + (M >> 4)]
    assert all(x < (1 << 32) for x in t)
    return B([(t[i] >> 1) + (1 << 25) for i in range(9)] + [t[9] >> 1])


def select(a, b):
    return B([max(x, y) for x, y in zip(a.hi, b.hi)])


def _columns(pairs):
    """pairs: list of (A, B) products summed; returns per-column upper bounds of the raw sums"""
    col = [0] * 19
    for a, b in pairs:
        for i in range(10):
            for j in range(10):
                col[i + j] += a.hi[i] * b.hi[j]
    return col


def mulsum(pairs):
    """models fe26_mul / fe26_sqr / fused variants: the 10x26 schedule with R0/R1 folding"""
    col = _columns(pairs)
    d = col[9]
    assert d < (1 << 64)
    d >>= 26
    c = 0
    for k in range(9):
        d += col[10 + k]
        assert d < (1 << 64), ("high column", k)
        u = M
        d >>= 26
        c += col[k] + u * R0
        assert c < (1 << 64), ("low column", k)
        c >>= 26
        c += u * R1
    # tail (fe26_mul_tail): d32 must fit 32 bits, c the carry into column 9
    assert d < (1 << 32), "d32 overflow"
    c += d * R0 + M
    assert c < (1 << 64)
    c >>= 22
    assert c < (1 << 32), "clo overflow"
    e = c * (R0 >> 4) + d * ((R0 >> 4) << 14) + M
    assert e < (1 << 64)
    e >>= 26
    e += c * (R1 >> 4) + d * ((R1 >> 4) << 14) + M
    assert e < (1 << 64)
    e >>= 26
    out = B([M, M, M + e] + [M] * 6 + [M9])
    assert out.within(1), "product not magnitude 1"
    return out


def mul(a, b):
    return mulsum([(a, b)])


def sqr(a):
    return mulsum([(a, a)])


def mul_small_norm(a, k):
    c = 0
    for i in range(9):
        c += a.hi[i] * k
        c >>= 26
    c += a.hi[9] * k
    x = c >> 22
    out = B([M + x * 0x3D1, M + (x << 6)] + [M] * 7 + [M9])
    assert out.within(1)
    return out


# ---- pt26.h ------------------------------------------------------------------------------------
def pt_add_mixed(px, py, pz, qx, qy):
    t0 = mul(px, qx)
    t1 = mul(py, qy)
    t3 = mul(add(qx, qy), add(px, py))
    t3 = add(t3, negate(add(t0, t1), 2))
    t4 = add(mul(qy, pz), py)
    y3 = add(mul(qx, pz), px)
    t0 = mul_int(t0, 3)
    t2 = mul_small_norm(pz, 21)
    z3 = add(t1, t2)
    t1 = add(t1, negate(t2, 1))
    y3 = mul_small_norm(y3, 21)
    x3 = mul(t4, y3)
    t2 = mul(t3, t1)
    return add(t2, negate(x3, 1)), add(mul(t1, z3), mul(y3, t0)), add(mul(z3, t4), mul(t0, t3))


def pt_add(p, q):
    px, py, pz = p
    qx, qy, qz = q
    t0, t1, t2 = mul(px, qx), mul(py, qy), mul(pz, qz)
    t3 = add(mul(add(px, py), add(qx, qy)), negate(add(t0, t1), 2))
    t4 = add(mul(add(py, pz), add(qy, qz)), negate(add(t1, t2), 2))
    y3 = add(mul(add(px, pz), add(qx, qz)), negate(add(t0, t2), 2))
    t0 = mul_int(t0, 3)
    t2 = mul_small_norm(t2, 21)
    z3 = add(t1, t2)
    t1 = add(t1, negate(t2, 1))
    y3 = mul_small_norm(y3, 21)
    x3 = mul(t4, y3)
    t2 = mul(t3, t1)
    return add(t2, negate(x3, 1)), add(mul(t1, z3), mul(y3, t0)), add(mul(z3, t4), mul(t0, t3))


def pt_double(p):
    px, py, pz = p
    t0 = sqr(py)
    z3 = mul_int(t0, 8)
    t1 = mul(py, pz)
    t2 = mul_small_norm(sqr(pz), 21)
    x3 = mul(t2, z3)
    y3 = add(t0, t2)
    rz = mul(t1, z3)
    t0 = add(t0, negate(mul_int(t2, 3), 3))
    ry = add(x3, mul(t0, y3))
    rx = mul_int(mul(t0, mul(px, py)), 2)
    return rx, ry, rz


def test_projective_invariant_closed():
    P = (B.mag(3), B.mag(3), B.mag(3))
    for r in (pt_add(P, P), pt_double(P), pt_add_mixed(*P, B.mag(1), B.mag(1))):
        assert all(c.within(3) for c in r)


def test_model_rejects_an_overflow():
    # sanity of the checker itself: magnitudes beyond the documented limit must trip it
    import pytest
    with pytest.raises(AssertionError):
        mul(B.mag(24), B.mag(24))
    with pytest.raises(AssertionError):
        negate(B.mag(3), 1)

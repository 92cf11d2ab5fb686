"""The row-spread field and group law (csrc/fe29r.h, pt29r.h) on the lane-level model tests/fer_model.py: values against
arithmetic mod p and the affine group law, widths asserted inside the model on random and extremal lazy inputs.  No GPU;
the GPU tests (tests/test_gpu_round5.py) run the compiled functions through the C-ABI on the same kinds of inputs."""
import random

import fer_model as F
import pyref as R

P = F.P


# what one reduction leaves at most in the limbs ("one unit" of the row form; asserted on every product below): limb 0 takes
# the fold of the top carry (x * 0x3D1, x < 2^16), limb 1 that carry times 8, the others a carry of 17 bits
BOUND1 = [(1 << 29) + (1 << 26), (1 << 29) + (1 << 19)] + [(1 << 29) + (1 << 18)] * 6 + [(1 << 24) + (1 << 18)]


def lazy(x, rnd, units=1):
    """a representation of x with limbs up to `units` * 2^29 (limb 8: units * 2^24): x + k p spread unevenly"""
    limbs = [(x >> (29 * j)) & (F.M if j < 8 else (1 << 32) - 1) for j in range(9)]
    if units > 1:
        for j in range(9):
            limbs[j] += (units - 1) * F.P_LIMBS[j]
    # move a carry down: limb j+1 -= 1, limb j += 2^29 (keeps the value)
    for j in range(8):
        if limbs[j + 1] > 0 and rnd.random() < 0.3 and limbs[j] + (1 << 29) < units * (1 << 29) + (1 << 26):
            limbs[j + 1] -= 1
            limbs[j] += 1 << 29
    return F.lane_const(lambda j, r: limbs[j] if j <= 8 else 0)


def test_row_moves():
    v = list(range(64))
    assert F.row_shr(v, 1)[17] == 16 and F.row_shr(v, 1)[16] == 0 and F.row_shl(v, 9)[16] == 25 and F.row_shl(v, 9)[23] == 0
    assert F.row_newbcast(v, 8)[40] == 40 and F.row_newbcast(v, 8)[33] == 40
    r0, r1, r2, r3 = F.bcast_rows(v)
    for l in range(64):
        assert (r0[l], r1[l], r2[l], r3[l]) == (l & 15, 16 + (l & 15), 32 + (l & 15), 48 + (l & 15))


def test_fer_mul_random_and_lazy():
    rnd = random.Random(1)
    for it in range(300):
        vals = [(rnd.randrange(P), rnd.randrange(P)) for _ in range(4)]
        if it < 8:
            vals = [(P - 1 - it, P - 1)] * 2 + [(0, rnd.randrange(P)), (1, P - 1)]
        a = F.by_row(*[lazy(x, rnd, 1 + (it % 3)) for x, _ in vals])
        b = F.by_row(*[lazy(y, rnd, 1 + ((it // 3) % 2)) for _, y in vals])
        r = F.fer_mul(a, b)
        for row in range(4):
            assert F.fer_value(r, row) % P == vals[row][0] * vals[row][1] % P
            assert all(r[16 * row + j] <= BOUND1[j] for j in range(9))


def test_fer_mul_extremal_limbs():
    """all limbs at the top of the unit budget: [3] x [2] + [1] x [1] + a [3] addend, the largest sums the formulas form"""
    top = lambda u: F.lane_const(lambda j, r: u * BOUND1[j] if j <= 8 else 0)
    a3, b2, c1 = top(3), top(2), top(1)
    r = F.fer_mulsum([(a3, b2), (c1, c1)], addend=top(3))
    exp = (F.fer_value(a3) * F.fer_value(b2) + F.fer_value(c1) ** 2 + F.fer_value(top(3))) % P
    assert F.fer_value(r) % P == exp
    assert all(r[j] <= BOUND1[j] for j in range(9))
    r = F.fer_mulsum([(top(1), top(6))])
    assert F.fer_value(r) % P == F.fer_value(top(1)) * F.fer_value(top(6)) % P
    assert all(r[j] <= BOUND1[j] for j in range(9))


def test_small_norm_and_negate():
    rnd = random.Random(2)
    for _ in range(100):
        x = rnd.randrange(P)
        v = lazy(x, rnd, 1)
        k = F.lane_const(lambda j, r: (63, 21, 4, 3)[r])
        s = F.fer_small_norm(v, k)
        for row, kk in enumerate((63, 21, 4, 3)):
            assert F.fer_value(s, row) % P == x * kk % P
        n = F.fer_negate(v, 1)
        assert F.fer_value(n) % P == (-x) % P
        assert F.fer_value(F.fer_norm(F.fer_add(n, n))) % P == (-2 * x) % P


def proj(pt, z, rnd, yunits=1):
    if pt is None:
        return lazy(0, rnd), lazy(z % P or 1, rnd, yunits), lazy(0, rnd)
    return lazy(pt[0] * z % P, rnd), lazy(pt[1] * z % P, rnd, yunits), lazy(z % P, rnd)


def affine(X, Y, Z):
    out = []
    for row in range(4):
        z = F.fer_value(Z, row) % P
        if z == 0:
            out.append(None)
        else:
            zi = pow(z, P - 2, P)
            out.append((F.fer_value(X, row) * zi % P, F.fer_value(Y, row) * zi % P))
    assert all(o == out[0] for o in out), "the rows must agree"
    return out[0]


def test_ptr_double_and_add():
    rnd = random.Random(3)
    G = R.G
    for it in range(60):
        p = R.mul(rnd.randrange(1, R.N), G)
        q = R.mul(rnd.randrange(1, R.N), G)
        if it % 10 == 1:
            q = p
        if it % 10 == 2:
            q = (p[0], P - p[1])
        if it % 10 == 3:
            p = None
        if it % 10 == 4:
            q = None
        zp, zq = rnd.randrange(1, P), rnd.randrange(1, P)
        P1, Q1 = proj(p, zp, rnd, 1 + it % 2), proj(q, zq, rnd, 1 + (it // 2) % 2)
        assert affine(*F.ptr_add(P1, Q1)) == R.add(p, q)
        d = F.ptr_double(*P1)
        assert affine(*d) == R.add(p, p)
        # chains: the outputs feed the next operation (the Horner recurrence: 16 doublings, one addition)
        acc = P1
        ref = p
        for _ in range(5):
            acc = F.ptr_double(*acc)
            ref = R.add(ref, ref)
        acc = F.ptr_add(acc, Q1)
        ref = R.add(ref, q)
        acc = F.ptr_add(acc, acc)
        ref = R.add(ref, ref)
        assert affine(*acc) == ref

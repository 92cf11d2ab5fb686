"""Model of sc_split_glv_odd (secp256k1_voi_amd/csrc/sc.h): after the reference's splitGLV
(point_mul_glv.go:59-117) both halves are made odd by adding a short vector of the GLV lattice
{(a, b): a + b*lambda = 0 mod n}, so the signed-odd-digit ladder needs no final correction.
Checks on integers: the congruence is preserved, both halves are odd, and the magnitudes stay
below 2^129 (the ladder reads 129-bit half-scalars)."""
import random

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
LAM = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72
A1 = 0x3086D221A7D46BCDE86C90E49284EB15
B1 = -0xE4437ED6010E88286F547FA90ABFE4C3
A2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8
B2 = A1
G1 = 0x3086D221A7D46BCDE86C90E49284EB153DAA8A1471E8CA7FE893209A45DBB031
G2 = 0xE4437ED6010E88286F547FA90ABFE4C4221208AC9DF506C61571B4AE8AC47F71


def split_reference(k):
    """splitGLV as the reference computes it (rounded multiplications by g1, g2 >> 384)"""
    c1 = (k * G1 + (1 << 383)) >> 384
    c2 = (k * G2 + (1 << 383)) >> 384
    k2 = (-c1 * B1 - c2 * B2) % N
    k1 = (k - k2 * LAM) % N
    if k1 > N // 2:
        k1 -= N
    if k2 > N // 2:
        k2 -= N
    return k1, k2


def signed_digit(k, pos):
    """digit `pos` of the ladders' recoding of an odd |k| < 2^129: 2 * (bits 4 pos + 1 .. 4 pos + 4) - 15"""
    return 2 * ((abs(k) >> (4 * pos + 1)) & 15) - 15


def make_odd(k1, k2, avoid_last_addition=True):
    """the device rule: magnitudes and signs in, magnitudes and signs out"""
    p = (k1 & 1, k2 & 1)
    if p == (1, 1):
        return k1, k2
    if p == (0, 0):
        c1, c2 = k1 + A1, k2 + B1
        if avoid_last_addition and k1 == 0:
            # k = k2 lambda with k2 = 2 s d w (d the digit the ladder adds last, w its weight): + v1 would make that last
            # addition P + P; - v1 does not (test_no_last_addition_collisions)
            s2 = 1 if c2 > 0 else -1
            if k2 == 2 * s2 * signed_digit(c2, 0) or k2 == 2 * s2 * signed_digit(c2, 28) * 16 ** 28:
                c1, c2 = k1 - A1, k2 - B1
        return c1, c2
    if p == (1, 0):
        s = -1 if k1 >= 0 else 1
        return k1 + s * A2, k2 + s * B2
    s = -1 if k2 >= 0 else 1
    return k1 + s * (A2 - A1), k2 + s * (B2 - B1)


def test_lattice_vectors():
    for a, b in ((A1, B1), (A2, B2), (A2 - A1, B2 - B1)):
        assert (a + b * LAM) % N == 0
    assert (A1 & 1, B1 & 1) == (1, 1) and (A2 & 1, B2 & 1) == (0, 1) and ((A2 - A1) & 1, (B2 - B1) & 1) == (1, 0)


def test_odd_split_bounds_and_congruence():
    rnd = random.Random(2024)
    ks = [0, 1, 2, 3, N - 1, N - 2, N // 2, N // 2 + 1, LAM, N - LAM, (1 << 128) - 1, 1 << 128, (1 << 255) + 5]
    ks += [rnd.randrange(N) for _ in range(50000)]
    ks += [(rnd.randrange(1 << 16) * A1 + rnd.randrange(1 << 16) * A2) % N for _ in range(2000)]
    worst = 0
    for k in ks:
        k1, k2 = split_reference(k)
        assert (k1 + k2 * LAM - k) % N == 0 and abs(k1) < (1 << 128) and abs(k2) < (1 << 128)
        o1, o2 = make_odd(k1, k2)
        assert (o1 + o2 * LAM - k) % N == 0
        assert o1 & 1 and o2 & 1
        worst = max(worst, abs(o1).bit_length(), abs(o2).bit_length())
    assert worst <= 129


def test_worst_case_bound_is_structural():
    # |k1|, |k2| < 2^128 in; every branch adds the long component against the sign of the half it
    # could overflow, so the bound does not depend on the sampled scalars
    lim = 1 << 128
    assert lim + A1 < (1 << 129) and lim + abs(B1) < (1 << 129)                  # (even, even): any signs
    assert max(lim, A2) < (1 << 129) and lim + B2 < (1 << 129)                  # (odd, even): A2 against k1
    assert lim + (A2 - A1) < (1 << 129) and max(lim, B2 - B1) < (1 << 129)      # (even, odd): B2-B1 against k2


def split_lattice_form(k):
    """sc_split_glv as the device computes it since round 3: k1 = k - c1 a1 - c2 a2, k2 = -c1 b1 - c2 b2 as integers modulo
    2^256 (four 128-bit products, no reduction mod n), the sign taken from bit 255"""
    M = 1 << 256
    c1 = (k * G1 + (1 << 383)) >> 384
    c2 = (k * G2 + (1 << 383)) >> 384
    d1 = (k - c1 * A1 - c2 * A2) % M
    d2 = (-c1 * B1 - c2 * B2) % M
    return (d1 - M if d1 >> 255 else d1), (d2 - M if d2 >> 255 else d2)


def test_lattice_form_split_is_the_reference_split():
    """same (k1, k2), signs included, as the reference's three products mod n - for the boundary values, values on the
    lattice's short vectors and 10^5 random scalars; c1, c2 below 2^128 (the device multiplies 128-bit operands)"""
    rnd = random.Random(2025)
    ks = [0, 1, 2, 3, N - 1, N - 2, N // 2, N // 2 + 1, LAM, N - LAM, (1 << 128) - 1, 1 << 128, (1 << 255) + 5]
    ks += [rnd.randrange(N) for _ in range(100000)]
    ks += [(rnd.randrange(1 << 16) * A1 + rnd.randrange(1 << 16) * A2) % N for _ in range(2000)]
    for k in ks:
        assert split_lattice_form(k) == split_reference(k), hex(k)
        assert ((k * G1 + (1 << 383)) >> 384) < (1 << 128) and ((k * G2 + (1 << 383)) >> 384) < (1 << 128)


def ladder_events(u2, keyed, avoid_last_addition=True):
    """Integer simulation of the two ladders' partial sums a Q + b lambda Q over the odd split of u2: the positions at which
    an addition would meet an operand at infinity, P + P or P - P (what makes Z = 0 in the incomplete formulas)."""
    k1, k2 = make_odd(*split_reference(u2), avoid_last_addition=avoid_last_addition)
    assert (k1 + k2 * LAM - u2) % N == 0 and k1 & 1 and k2 & 1 and abs(k1) < (1 << 129) and abs(k2) < (1 << 129)
    s1, s2 = (1 if k1 > 0 else -1), (1 if k2 > 0 else -1)
    d1, d2 = [signed_digit(k1, i) for i in range(32)], [signed_digit(k2, i) for i in range(32)]
    ev = []

    def add(a, b, da, db, where):
        v, w = (a + b * LAM) % N, (da + db * LAM) % N
        if v == 0 or w == 0 or v == w or v == (N - w) % N:
            ev.append(where)
        return a + da, b + db
    if keyed:       # k_verify_fast<ECDSA_KEYED>: start +-2^116 Q +- 2^116 lambda Q, four rounds of eight chunks, four doublings between rounds
        a, b = s1 << 116, s2 << 116
        for rnd in range(4):
            j = 3 - rnd
            if rnd:
                a, b = 16 * a, 16 * b
            for c in range(8):
                i = 4 * c + j
                a, b = add(a, b, s1 * d1[i] * 16 ** (4 * c), 0, ("k1", i))
                a, b = add(a, b, 0, s2 * d2[i] * 16 ** (4 * c), ("k2", i))
    else:           # k_verify_fast<ECDSA>: start +-Q +- lambda Q, 32 windows of (four doublings, two additions)
        a, b = s1, 0
        a, b = add(a, b, 0, s2, "start")
        for i in range(31, -1, -1):
            a, b = 16 * a, 16 * b
            a, b = add(a, b, s1 * d1[i], 0, ("k1", i))
            a, b = add(a, b, 0, s2 * d2[i], ("k2", i))
    assert (a + b * LAM - u2) % N == 0
    return ev


def test_no_last_addition_collisions():
    """Scalars with SMALL natural halves at the weights where the ladders' additions sit (u2 = a w + b w lambda, |a|, |b| <= 40,
    w in {1, 16^4, 16^27, 16^28, 16^31}: 32 800 of them) through both ladders.  With the plain rule exactly two of them make
    the LAST table addition of a ladder add a point to itself (u2 = -26 lambda: general ladder; -26 * 16^28 lambda: the
    ladder over per-key tables); with the rule the device uses, none meets any exceptional addition."""
    plain, fixed = [], []
    for w in (1, 16 ** 4, 16 ** 27, 16 ** 28, 16 ** 31):
        for a in range(-40, 41):
            for b in range(-40, 41):
                u2 = (a * w + b * w * LAM) % N
                if u2 == 0:
                    continue
                for keyed in (False, True):
                    if ladder_events(u2, keyed, avoid_last_addition=False):
                        plain.append((a, b, w.bit_length(), keyed))
                    if ladder_events(u2, keyed):
                        fixed.append((a, b, w.bit_length(), keyed))
    assert sorted(plain) == [(0, -26, 1, False), (0, -26, 113, True)]
    assert fixed == []
    # the two values by name (secp256k1_voi_amd/synth.py quotes them)
    assert (-26 * LAM) % N == 0x87e0663476a3092f3a2127be2e21ceceb7763854dc6939d318220a5890470db5
    assert (-26 * 16 ** 28 * LAM) % N == 0xcecf64212ab6eb5a5997b96cac25ca2bb234b7d1e5bd755372f392d91409e89f


def test_random_scalars_meet_no_exceptional_addition():
    rnd = random.Random(77)
    for _ in range(3000):
        u2 = rnd.randrange(1, N)
        assert not ladder_events(u2, False) and not ladder_events(u2, True)

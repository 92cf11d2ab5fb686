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


def make_odd(k1, k2):
    """the device rule: magnitudes and signs in, magnitudes and signs out"""
    p = (k1 & 1, k2 & 1)
    if p == (1, 1):
        return k1, k2
    if p == (0, 0):
        return k1 + A1, k2 + B1
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

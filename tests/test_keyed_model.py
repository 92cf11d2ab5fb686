"""Integer model of the recoding behind the per-key tables (engine.hip: ds4_init_chunked and the keyed branch
of k_verify_fast; keyed.hip: table contents).  No GPU: it pins down the identities the device code relies on.

  * an odd k < 2^129 is  16^32 + sum_{i<32} d_i 16^i  with d_i = 2 nib_i - 15, nib_i = bits 4i+1 .. 4i+4 of k
    (the general kernel's signed odd digits, complete_path.h: ds_init / ds_next);
  * digit i = 4c + j belongs to chunk c (base 2^(16c) Q) and round j; running the rounds 3, 2, 1, 0 with four
    doublings in between and starting from 2^116 Q gives k Q:  16^3 * 2^116 = 16^32;
  * the nibble stream is consumed round 3 first, chunks upwards: nibble 4c + j sits at place 8 (3 - j) + c
    from the top of a 128-bit word, i.e. at bit 124 - 4 (8 (3 - j) + c).
"""
import random

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def nibbles(k):
    return [(k >> (4 * i + 1)) & 15 for i in range(32)]


def chunked_stream(k):
    """What ds4_init_chunked builds: a 128-bit integer, first-consumed nibble on top."""
    s = 0
    for i, nib in enumerate(nibbles(k)):
        c, j = i >> 2, i & 3
        s |= nib << (124 - 4 * ((3 - j) * 8 + c))
    return s


def ladder_multiple(k):
    """The multiple of Q the keyed ladder computes, as an integer: table entry e of chunk c is (2e + 1) 2^(16c) Q."""
    s = chunked_stream(k)
    acc = 1 << 116                                  # the lead point
    for rnd in range(4):
        if rnd:
            acc *= 16                               # four doublings
        for c in range(8):
            w = (s >> 124) & 15                     # ds4_next
            s = (s << 4) & ((1 << 128) - 1)
            neg = w < 8
            entry = 7 - w if w < 8 else w - 8
            term = (2 * entry + 1) << (16 * c)
            acc += -term if neg else term
    assert s == 0
    return acc


def test_signed_odd_digits_reproduce_k():
    rnd = random.Random(5)
    ks = [1, 3, (1 << 129) - 1, (1 << 128) + 1, (1 << 128) - 1, 0x1_0001_0001_0001_0001_0001_0001_0001_0001 | 1]
    ks += [rnd.getrandbits(129) | 1 for _ in range(2000)]
    for k in ks:
        d = [2 * nib - 15 for nib in nibbles(k)]
        assert all(x % 2 for x in d) and all(-15 <= x <= 15 for x in d)
        assert 16 ** 32 + sum(x * 16 ** i for i, x in enumerate(d)) == k


def test_chunked_rounds_reproduce_k():
    rnd = random.Random(6)
    ks = [1, 3, 15, 17, (1 << 16) - 1, (1 << 16) + 1, (1 << 129) - 1, (1 << 128) + 1, (1 << 128) - 1]
    ks += [rnd.getrandbits(129) | 1 for _ in range(2000)]
    ks += [(rnd.getrandbits(129) | 1) & ~(0xFFFF << (16 * rnd.randrange(8))) | 1 for _ in range(500)]   # an empty chunk
    for k in ks:
        assert ladder_multiple(k) == k, hex(k)


def test_table_points_are_never_the_identity():
    """Every table entry (2e + 1) 2^(16c) Q and the lead point 2^116 Q is a non-zero multiple of Q below the group
    order, so the table build (doublings, additions of 2B to an odd multiple of B) meets no exceptional case for a
    key of prime order (every valid key)."""
    for c in range(8):
        for e in range(8):
            assert 0 < ((2 * e + 1) << (16 * c)) < N_ORDER
            assert ((2 * e + 1) << (16 * c)) % N_ORDER != (2 << (16 * c)) % N_ORDER      # (2e+1) B != 2 B
    assert 0 < (1 << 116) < N_ORDER

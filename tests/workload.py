"""Synthetic inputs for the parity tests and the bench.

Valid signatures are produced with plain Python integers mod n plus the oracle's base
multiplication (test infrastructure); nothing here is on the product path.
"""
import random

import numpy as np

import pyref as R


def _b32(x):
    return int(x).to_bytes(32, "big")


def make_ecdsa_batch(oracle, n, seed=1, n_keys=None, corrupt_every=0, low_s=True):
    """Returns dict of uint8 arrays pub (n,64), digest (n,32), r (n,32), s (n,32) and the
    list `kinds` describing the corruption applied to item i (None = valid)."""
    rnd = random.Random(seed)
    n_keys = n_keys or max(1, min(n, 64))
    keys = []
    for _ in range(n_keys):
        d = rnd.randrange(1, R.N)
        q = oracle.scalar_base_mult_vartime(_b32(d))
        keys.append((d, q[1:]))
    pub = np.zeros((n, 64), np.uint8)
    dig = np.zeros((n, 32), np.uint8)
    rr = np.zeros((n, 32), np.uint8)
    ss = np.zeros((n, 32), np.uint8)
    kinds = []
    for i in range(n):
        ki = rnd.randrange(n_keys)
        d, q = keys[ki]
        digest = rnd.randbytes(32)
        k = rnd.randrange(1, R.N)
        Rp = oracle.scalar_base_mult_vartime(_b32(k))
        r = int.from_bytes(Rp[1:33], "big") % R.N
        e = int.from_bytes(digest, "big") % R.N
        s = pow(k, -1, R.N) * (e + r * d) % R.N
        if low_s and s > R.N // 2:
            s = R.N - s
        kind = None
        qb = bytearray(q)
        db = bytearray(digest)
        if corrupt_every and i % corrupt_every == 0:
            kind = rnd.choice(["r", "s", "digest", "qx", "r0", "s0", "rn", "high_s", "offcurve"])
            if kind == "r":
                r ^= 1 << rnd.randrange(255)
                r %= R.N
                r = r or 1
            elif kind == "s":
                s = (s + 1 + rnd.randrange(1000)) % R.N or 1
            elif kind == "digest":
                db[rnd.randrange(32)] ^= 1 << rnd.randrange(8)
            elif kind == "qx":       # another valid key
                qb = bytearray(keys[(ki + 1) % n_keys][1])
            elif kind == "r0":
                r = 0
            elif kind == "s0":
                s = 0
            elif kind == "rn":
                r = R.N + (r % 1000)          # non-canonical r
            elif kind == "high_s":
                s = R.N - s                   # still valid unless reject_malleable
            elif kind == "offcurve":
                qb[63] ^= 1
        pub[i] = np.frombuffer(bytes(qb), np.uint8)
        dig[i] = np.frombuffer(bytes(db), np.uint8)
        rr[i] = np.frombuffer(_b32(r), np.uint8)
        ss[i] = np.frombuffer(_b32(s), np.uint8)
        kinds.append(kind)
    return {"pub": pub, "digest": dig, "r": rr, "s": ss, "kinds": kinds}

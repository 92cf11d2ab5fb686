#!/usr/bin/env python3
"""Randomised differential run of the multi-scalar multiplication (round 3: signed digits, equal-range bucket pass with
stitching, quad-spread tail, one- and two-word sort pairs, optional two-part flow) against big-integer arithmetic on
points with known discrete logarithms: sum k_i (d_i G) == (sum k_i d_i) G.  Sizes around every geometry switch (8 / 12 /
16-bit windows), scalar patterns that stress the recoding (carries through all windows, top-window overflow, zero
digits, few distinct values -> buckets spread over many ranges), repeated and negated points, identity inputs.

    python3 tools/stress_msm.py [iterations] [seed]
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import secp256k1_voi_amd as S

N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
P = 2**256 - 2**32 - 977
b32 = lambda v: int(v).to_bytes(32, "big")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
eng = S.Engine(0)

POOL = 1 << 15
dpool = [rnd.randrange(1, N) for _ in range(POOL)]
ppool = eng.scalar_base_mult_batch(np.frombuffer(b"".join(b32(d) for d in dpool), np.uint8).reshape(POOL, 32))


def scalar(kind):
    if kind == 0:
        return rnd.randrange(N)
    if kind == 1:   # carries through every window: ...8000 8000 / ...FFFF FFFF patterns
        w = rnd.choice([0x8000, 0x8001, 0xFFFF, 0x7FFF, 0x0000, 0x0001])
        return sum(w << (16 * i) for i in range(16)) % N
    if kind == 2:   # few distinct values
        return [3, N - 1, 1 << 128, (1 << 128) - 1, 0xA2A8918CA85BAFE22016D0B917E4DD77, 0][rnd.randrange(6)]
    if kind == 3:   # sparse
        return (1 << rnd.randrange(256)) % N
    return rnd.randrange(1 << 64)


t0 = time.time()
for it in range(iters):
    n = rnd.choice([1, 2, 7, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 20000, 65536, 100000,
                    rnd.randrange(1, 3000), rnd.randrange(3000, 40000)])
    if os.environ.get("S2K_MSM_SPLIT_WINDOW") and rnd.random() < 0.5:     # the two-part flow starts at 2^17 terms = 2^16 inputs
        n = rnd.choice([65536, 65537, 70000, 100000, 131072, 150001])
    kind = rnd.randrange(5)
    mix = rnd.random() < 0.3
    idx = [rnd.randrange(POOL if rnd.random() < 0.8 else 4) for _ in range(n)]      # sometimes only 4 distinct points
    ks = [scalar(rnd.randrange(5) if mix else kind) for _ in range(n)]
    pts = ppool[idx].copy()
    sign = [1] * n
    for i in range(n):
        r = rnd.random()
        if r < 0.05:          # negated point: (x, p - y)
            y = P - int.from_bytes(bytes(pts[i, 33:65]), "big")
            pts[i, 33:65] = np.frombuffer(b32(y), np.uint8)
            sign[i] = -1
        elif r < 0.07:        # identity input
            pts[i, :] = 0
            sign[i] = 0
    total = sum(k * s * dpool[j] for k, s, j in zip(ks, sign, idx)) % N
    want = bytes(65) if total == 0 else eng.scalar_base_mult_batch([b32(total)])[0].tobytes()
    got = eng.multi_scalar_mult(np.frombuffer(b"".join(b32(k) for k in ks), np.uint8).reshape(n, 32), pts)
    if got != want:
        print("MISMATCH iteration %d n=%d kind=%d mix=%s seed=%d" % (it, n, kind, mix, seed), flush=True)
        sys.exit(1)
print("ok: %d iterations, seed %d, %.1f s" % (iters, seed, time.time() - t0))

#!/usr/bin/env python3
"""Randomised run of the BIP-340 whole-batch check (two-stream flow of round 3) and of the bisection: random batch sizes and
key multiplicities; the valid batch must be accepted, the batch with damaged signatures rejected, and the bisection's
per-signature verdicts must equal the per-signature verifier's.

    python3 tools/stress_rlc.py [iterations] [seed]
"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_schnorr_batch

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rnd = random.Random(seed)
eng = S.Engine(0)
t0 = time.time()
for it in range(iters):
    n = rnd.choice([1, 2, 3, 63, 64, 65, 255, 256, 257, 1000, 5000, 8192, 20000, 70000, 300000, rnd.randrange(1, 4000)])
    nk = max(1, rnd.choice([1, 2, n // 16 or 1, n // 3 or 1, n]))
    pk, msgs, sig = synth_schnorr_batch(eng, n, min(nk, n), seed=seed * 1000 + it)
    assert eng.schnorr_batch_verify_rlc(pk, msgs, sig, os.urandom(32)), ("valid batch rejected", it, n, nk)
    bad = sig.copy()
    nbad = rnd.choice([1, 1, 2, 5])
    pos = sorted(set(rnd.randrange(n) for _ in range(nbad)))
    for p in pos:
        what = rnd.randrange(3)
        if what == 0:
            bad[p, 32 + rnd.randrange(32)] ^= 1 << rnd.randrange(8)        # s
        elif what == 1:
            bad[p, rnd.randrange(32)] ^= 1 << rnd.randrange(8)             # r (may stop being an x-coordinate)
        else:
            bad[p, 32:] = 0xFF                                             # s >= n
    assert not eng.schnorr_batch_verify_rlc(pk, msgs, bad, os.urandom(32)), ("damaged batch accepted", it, n, nk, pos)
    single = eng.schnorr_verify_batch(pk, msgs, bad)
    assert not single[pos].any() and int(single.sum()) == n - len(pos)
    got = eng.schnorr_verify_batch_auto(pk, msgs, bad, os.urandom(32))
    assert np.array_equal(np.asarray(got), single), ("bisection differs", it, n, nk, pos)
print("ok: %d iterations, seed %d, %.1f s" % (iters, seed, time.time() - t0))

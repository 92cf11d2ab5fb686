"""Randomised differential run of the wave-per-signature ladders (small calls) against the CPU oracle and against
the lane-per-signature kernels.

    python tools/stress_small.py [iterations] [seed] [max_n]

Every iteration draws a call size (1 .. 6000, mostly small), a key population, damage of every kind (bit flips in r / s /
digest / key, zero and out-of-range values, foreign keys, keys that are not on the curve, BIP-340 keys and r that are no x
coordinate, recovery ids 0 .. 4) and runs ECDSA verification, BIP-340 verification and public-key recovery three ways: the
wave-per-signature ladder (threshold above the size), four lanes per signature (ECDSA; the other two entry points take the lane
kernels then), the lane kernels (both thresholds 0), and the oracle; synchronously and, every
fourth iteration, as two tickets in flight.  Prints one line per iteration; exits non-zero on the first mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as O
import pyref as R
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import N_ORDER, synth_batch, synth_schnorr_batch

P_FIELD = 2**256 - 2**32 - 977


def b32(v):
    return np.frombuffer(int(v).to_bytes(32, "big"), np.uint8)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    max_n = int(sys.argv[3]) if len(sys.argv) > 3 else 6000        # (above 6000: the mid-size range of the four-lanes-per-signature ladders)
    O.build()
    eng = S.Engine(0)
    threads = os.cpu_count() or 1
    not_x = next(x for x in range(2, 100) if R.lift_x(x, 0) is None)
    edge = [0, 1, N_ORDER - 1, N_ORDER, N_ORDER + 1, P_FIELD - N_ORDER - 1, P_FIELD - N_ORDER, P_FIELD - 1, P_FIELD, 2**256 - 1]
    for it in range(iters):
        rng = np.random.default_rng(seed0 * 100019 + it)
        n = int(rng.choice([rng.integers(1, 9), rng.integers(1, 300), rng.integers(300, 2000), rng.integers(2000, 6000)]))
        if max_n > 6000:
            n = int(rng.integers(3000, max_n))
        nk = int(rng.choice([1, max(1, n // 7), n]))
        pub, dig, r, s = synth_batch(eng, n, nk, seed=int(rng.integers(1 << 30)))
        dmg = rng.integers(0, int(rng.choice([8, 30, 300])), size=n)
        idx = lambda k: np.nonzero(dmg == k)[0]
        i = idx(0); r[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(1); s[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(2); dig[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(3); pub[i, rng.integers(0, 64, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        for j in idx(4): r[j] = b32(edge[int(rng.integers(len(edge)))])
        for j in idx(5): s[j] = b32(edge[int(rng.integers(len(edge)))])
        i = idx(6); pub[i] = pub[(i + 1) % n]
        for j in idx(7): pub[j, :32] = b32(edge[int(rng.integers(len(edge)))])
        rm = bool(rng.integers(0, 2))
        rid = rng.integers(0, 5 if rng.integers(0, 4) == 0 else 4, size=n).astype(np.uint8)
        m = n
        pk, msgs, sig = synth_schnorr_batch(eng, m, max(1, min(nk, m)), seed=int(rng.integers(1 << 30)), msg_len=int(rng.choice([0, 1, 32, 55, 56, 64, 119])))
        d2 = rng.integers(0, int(rng.choice([6, 40])), size=m)
        i = np.nonzero(d2 == 0)[0]; sig[i, rng.integers(0, 64, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        if msgs.shape[1]:
            i = np.nonzero(d2 == 1)[0]; msgs[i, 0] ^= 1
        for j in np.nonzero(d2 == 2)[0]: pk[j] = b32(not_x if rng.integers(0, 2) else edge[int(rng.integers(len(edge)))])
        for j in np.nonzero(d2 == 3)[0]: sig[j, :32] = b32(not_x if rng.integers(0, 2) else edge[int(rng.integers(len(edge)))])
        for j in np.nonzero(d2 == 4)[0]: sig[j, 32:] = b32(edge[int(rng.integers(len(edge)))])
        exp_v = O.ecdsa_verify_batch(pub, dig, r, s, reject_malleable=rm, nthreads=threads)
        exp_s = np.array([1 if O.schnorr_verify(bytes(pk[j]), bytes(msgs[j]), bytes(sig[j])) == 1 else 0 for j in range(m)], dtype=np.uint8)
        exp_r = [O.ecdsa_recover(bytes(dig[j]), bytes(r[j]), bytes(s[j]), int(rid[j])) for j in range(n)]
        bad = []
        for name, row_max, quad_max in (("wave", 1 << 20, 0), ("quad", 0, 1 << 20), ("lane", 0, 0)):
            eng.set_small_batch_max(row_max)
            eng.set_mid_batch_max(quad_max)
            if it % 4 == 3:
                t = [eng.ecdsa_verify_batch_submit(pub, dig, r, s, reject_malleable=rm) for _ in range(2)]
                got_v = t[1].wait()
                if not np.array_equal(t[0].wait(), got_v):
                    bad.append(name + " tickets differ")
            else:
                got_v = eng.ecdsa_verify_batch(pub, dig, r, s, reject_malleable=rm)
            got_s = eng.schnorr_verify_batch(pk, msgs, sig)
            q, ok = eng.ecdsa_recover_batch(dig, r, s, rid)
            got_r = [bytes(a) if k else None for a, k in zip(q, ok)]
            if not np.array_equal(got_v, exp_v):
                bad.append("%s ecdsa %s" % (name, np.nonzero(got_v != exp_v)[0][:8]))
            if not np.array_equal(got_s, exp_s):
                bad.append("%s schnorr %s" % (name, np.nonzero(got_s != exp_s)[0][:8]))
            if got_r != exp_r:
                bad.append("%s recover %s" % (name, [j for j in range(n) if got_r[j] != exp_r[j]][:8]))
        line = (f"it {it:4d} n {n:5d} keys {nk:5d} ecdsa valid {int(exp_v.sum()):5d} low-s {int(rm)} | schnorr valid {int(exp_s.sum()):5d} msg {msgs.shape[1]:3d} | "
                f"recovered {sum(e is not None for e in exp_r):5d} {'ok' if not bad else 'MISMATCH ' + '; '.join(bad)}")
        print(line, flush=True)
        if bad:
            sys.exit(1)
    print("stress ok")


if __name__ == "__main__":
    main()

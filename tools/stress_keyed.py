"""Randomised differential run of the grouped verification paths against the CPU oracle.

    python tools/stress_keyed.py [iterations] [seed]

Every iteration draws a batch size, a key population with a skewed reuse pattern, damage of every kind
(bit flips in r / s / digest / key, zero and out-of-range scalars, foreign keys, invalid keys shared by whole
groups), and a grouping configuration (off / auto / always, threshold, hash-table size, table cap), runs ECDSA
verification (and every third iteration BIP-340 per-signature verification and the whole-batch check) and
compares every verdict with the oracle's.  Prints one line per iteration; exits non-zero on the first mismatch."""
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


def key_indices(rng, n):
    kind = rng.integers(0, 4)
    if kind == 0:
        k = int(rng.integers(1, max(2, n // 3)))
        return rng.integers(0, k, size=n), k
    if kind == 1:                      # skewed: a few heavy keys, a long tail
        k = int(rng.integers(2, max(3, n // 2)))
        w = 1.0 / np.arange(1, k + 1) ** rng.uniform(0.5, 1.5)
        return rng.choice(k, size=n, p=w / w.sum()), k
    if kind == 2:                      # all distinct
        return np.arange(n), n
    k = int(rng.integers(1, 5))        # one to four keys
    return rng.integers(0, k, size=n), k


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    O.build()
    eng = S.Engine(0)
    threads = os.cpu_count() or 1
    not_x = next(x for x in range(2, 100) if R.lift_x(x, 0) is None)
    for it in range(iters):
        rng = np.random.default_rng(seed0 * 100003 + it)
        n = int(rng.choice([rng.integers(256, 700), rng.integers(700, 5000), rng.integers(5000, 30000)]))
        kidx, nk = key_indices(rng, n)
        mode = int(rng.choice([S.KEYS_OFF, S.KEYS_AUTO, S.KEYS_AUTO, S.KEYS_ALWAYS]))
        cfg = dict(min_group=int(rng.choice([0, 0, 2, 3, 9])), hash_bits=int(rng.choice([0, 0, 0, 5, 9])),
                   max_tables=int(rng.choice([0, 0, 0, 7, 100])))
        eng.set_key_grouping(mode, **cfg)
        pub, dig, r, s = synth_batch(eng, n, nk, seed=int(rng.integers(1 << 30)), key_idx=kidx)
        dmg = rng.integers(0, int(rng.choice([6, 20, 200])), size=n)
        idx = lambda k: np.nonzero(dmg == k)[0]
        i = idx(0); r[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(1); s[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(2); dig[i, rng.integers(0, 32, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(3); pub[i, rng.integers(0, 64, size=i.size)] ^= (1 << rng.integers(0, 8, size=i.size)).astype(np.uint8)
        i = idx(4); r[i] = 0
        i = idx(5); pub[i] = pub[(i + 1) % n]
        keys = np.unique(pub, axis=0)
        for j in range(min(3, len(keys))):        # invalid keys shared by whole groups
            which = rng.integers(0, 4)
            bad = [keys[j].copy() for _ in range(1)][0]
            if which == 0: bad[63] ^= 1
            elif which == 1: bad[:32] = np.frombuffer((P_FIELD + 3).to_bytes(32, "big"), np.uint8)
            elif which == 2: bad[:] = 0
            else: continue
            pub[(pub == keys[j]).all(axis=1)] = bad
        rm = bool(rng.integers(0, 2))
        exp = O.ecdsa_verify_batch(pub, dig, r, s, reject_malleable=rm, nthreads=threads)
        got = eng.ecdsa_verify_batch(pub, dig, r, s, reject_malleable=rm)
        st = eng.key_grouping_stats()
        ok = np.array_equal(got, exp)
        line = f"it {it:4d} n {n:6d} keys {nk:6d} mode {mode} {cfg} ecdsa valid {int(exp.sum()):6d} keyed {st['keyed']:6d} tables {st['tables']:5d} general {st['general']:6d} complete {st['complete']:4d} {'ok' if ok else 'MISMATCH'}"
        if not ok:
            print(line, np.nonzero(got != exp)[0][:10], flush=True)
            sys.exit(1)
        if it % 3 == 0:
            m = min(n, 6000)
            pk, msgs, sig = synth_schnorr_batch(eng, m, max(1, min(nk, m)), seed=int(rng.integers(1 << 30)))
            d2 = rng.integers(0, 30, size=m)
            i = np.nonzero(d2 == 0)[0]; sig[i, rng.integers(0, 64, size=i.size)] ^= 1
            i = np.nonzero(d2 == 1)[0]; msgs[i, 0] ^= 1
            k2 = np.unique(pk, axis=0)
            if rng.integers(0, 2):
                pk[(pk == k2[0]).all(axis=1)] = np.frombuffer(R.b32(not_x), np.uint8)
            e2 = np.array([1 if O.schnorr_verify(bytes(pk[j]), bytes(msgs[j]), bytes(sig[j])) == 1 else 0 for j in range(m)], dtype=np.uint8)
            g2 = eng.schnorr_verify_batch(pk, msgs, sig)
            ok2 = np.array_equal(g2, e2)
            allv = eng.schnorr_batch_verify_rlc(pk, msgs, sig, bytes(range(32)))
            ok3 = bool(allv) == bool(e2.all())
            line += f" | schnorr {m} valid {int(e2.sum())} {'ok' if ok2 else 'MISMATCH'} batch {'ok' if ok3 else 'MISMATCH'}"
            if not (ok2 and ok3):
                print(line, np.nonzero(g2 != e2)[0][:10], flush=True)
                sys.exit(1)
        print(line, flush=True)
    eng.set_key_grouping(S.KEYS_AUTO)
    print("stress ok")


if __name__ == "__main__":
    main()

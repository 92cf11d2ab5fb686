"""Randomised differential run of the streaming boundary against the synchronous entry point and the CPU oracle.

    python tools/stress_pipeline.py [iterations] [seed]

Every iteration draws a handful of batches (sizes from empty to tens of thousands, skewed key reuse, every kind of damage),
a grouping mode and flags, pinned or pageable inputs and verdict arrays, a depth (how many tickets stay in flight), an order of
waiting (in order, reversed, or polling), and sends them through s2k_ecdsa_verify_batch_submit / s2k_wait on one context AND
through a two-member group on the same device, with synchronous calls on the same context sprinkled in between.  In half of the
iterations the distinct keys of all batches also form a key set (chunk or joint tables, on the context and on the group) and a
random share of the tickets names its keys by index (s2k_ecdsa_verify_batch_keyset_submit, s2k_group_ecdsa_verify_batch_keyset_submit),
mixed with the plain ones.  The grouping modes include the adaptive default, and some iterations carry batches of 2^16 and more
signatures, so that its skipping comes and goes underneath.  Every ticket's verdicts must equal the synchronous call's, and a sample
of them the oracle's.  Exits non-zero on the first mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import oracle as O
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def make_batch(eng, rng, n, lone=False):
    if n == 0:
        return [np.zeros((0, w), np.uint8) for w in (64, 32, 32, 32)]
    nk = n if lone else int(rng.integers(1, max(2, n // int(rng.integers(1, 40)) + 1)))
    arrs = [np.array(a) for a in synth_batch(eng, n, nk, seed=int(rng.integers(1 << 30)))]
    for i in range(0, n, int(rng.integers(2, 12))):
        a = arrs[int(rng.integers(0, 4))]
        a[i, int(rng.integers(0, a.shape[1]))] ^= 1 << int(rng.integers(0, 8))
    if n > 4 and rng.integers(0, 3) == 0:        # zero / out-of-range scalars, high s
        arrs[2][1] = 0
        arrs[3][2] = 0xFF
    return arrs


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    O.build()
    eng = S.Engine(0)
    grp = S.Group([0, 0])
    total = 0
    for it in range(iters):
        nb = int(rng.integers(1, 7))
        sizes = [int(rng.choice([0, 1, 63, 255, 256, 257, 1000, 4097, int(rng.integers(1, 60000))])) for _ in range(nb)]
        lone = False
        if rng.integers(0, 3) == 0:                  # large batches: what the adaptive grouping learns from; half of them without a repeated key
            sizes = [int(rng.integers(1 << 16, 90000)) for _ in range(int(rng.integers(2, 5)))]
            nb = len(sizes)
            lone = bool(rng.integers(0, 2))
        mode = [S.KEYS_AUTO, S.KEYS_ADAPTIVE, S.KEYS_ADAPTIVE, S.KEYS_OFF, S.KEYS_ALWAYS][int(rng.integers(0, 5))]
        low_s = bool(rng.integers(0, 2))
        pinned = bool(rng.integers(0, 2))
        eng.set_key_grouping(mode)
        grp.set_key_grouping(mode)
        batches = [make_batch(eng, rng, n, lone) for n in sizes]
        refs = [eng.ecdsa_verify_batch(*b, reject_malleable=low_s) if b[0].shape[0] else np.zeros(0, np.uint8) for b in batches]
        for b, ref in zip(batches, refs):
            m = min(b[0].shape[0], 512)
            if m:
                exp = O.ecdsa_verify_batch(*(a[:m] for a in b), reject_malleable=low_s, nthreads=os.cpu_count() or 1)
                assert np.array_equal(ref[:m], exp), ("synchronous call vs oracle", it)
        # a key set of all the batches' distinct keys (damaged key bytes are keys of their own, and no public keys)
        use_ks = bool(rng.integers(0, 2)) and sum(sizes) > 0
        kidx, ks_ctx, ks_grp = None, None, None
        if use_ks:
            keys, inv = np.unique(np.concatenate([b[0] for b in batches]), axis=0, return_inverse=True)
            inv = inv.reshape(-1).astype(np.uint32)
            cuts = np.cumsum([0] + sizes)
            kidx = [np.ascontiguousarray(inv[cuts[k]:cuts[k + 1]]) for k in range(nb)]
            # (joint tables: 356 KiB per key at 4-bit digits, 1.04 MiB at 5, 3.6 MiB at 6; three copies of the set here)
            layout = int(rng.integers(1, 5)) if len(keys) <= 4000 else int(rng.integers(1, 4)) if len(keys) <= 20000 else 1
            ks_ctx, ks_grp = eng.keyset_create(keys, layout), grp.keyset_create(keys, layout)
        if pinned:
            src = []
            for b in batches:
                pb = [S.pinned_array(a.shape) for a in b]
                for d, a in zip(pb, b):
                    d[...] = a
                src.append(pb)
        else:
            src = batches
        for owner, name in ((eng, "context"), (grp, "group")):
            depth = int(rng.integers(1, 6))
            order = int(rng.integers(0, 3))
            tickets, pending = [], []
            for k, b in enumerate(src):
                out = S.pinned_array((b[0].shape[0],)) if (pinned and rng.integers(0, 2)) else None
                if out is not None:
                    out[...] = 9
                if use_ks and rng.integers(0, 2):
                    tickets.append(owner.ecdsa_verify_batch_keyset_submit(ks_ctx if name == "context" else ks_grp, kidx[k], b[1], b[2], b[3],
                                                                          out=out, reject_malleable=low_s))
                else:
                    tickets.append(owner.ecdsa_verify_batch_submit(*b, out=out, reject_malleable=low_s))
                pending.append(k)
                if name == "context" and rng.integers(0, 4) == 0:          # a synchronous call in between
                    j = int(rng.integers(0, nb))
                    if batches[j][0].shape[0]:
                        assert np.array_equal(eng.ecdsa_verify_batch(*batches[j], reject_malleable=low_s), refs[j]), ("interleaved synchronous call", it)
                while len(pending) >= depth:
                    j = pending.pop(0)
                    assert np.array_equal(tickets[j].wait(), refs[j]), (name, "ticket", it, j, sizes[j])
            if order == 1:
                pending.reverse()
            for j in pending:
                if order == 2 and name == "context":
                    while not tickets[j].done():
                        pass
                assert np.array_equal(tickets[j].wait(), refs[j]), (name, "ticket at the end", it, j, sizes[j])
        if use_ks:
            ks_ctx.close()
            ks_grp.close()
        total += sum(sizes)
        print("iteration %d: %d batches %s%s, mode %d, low_s %d, pinned %d, key set %s, adaptive %s" %
              (it, nb, sizes, " (no repeated keys)" if lone else "", mode, low_s, pinned, ("layout %d of %d keys" % (layout, len(keys))) if use_ks else "-",
               eng.key_grouping_adaptive()), flush=True)
    eng.set_key_grouping(S.KEYS_ADAPTIVE)
    grp.close()
    print("ok: %d iterations, %d signatures through submit / wait and the group" % (iters, total))


if __name__ == "__main__":
    main()

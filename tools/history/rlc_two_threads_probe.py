#!/usr/bin/env python3
"""BASELINE configs 3 and 4 from host memory: one verifier calling synchronously against TWO verifiers (two contexts, two host
threads) taking whole batches alternately, so that one batch's transfer runs beside the other's kernels.  Page-locked inputs.
    python tools/rlc_two_threads_probe.py [log2 n] [calls per thread]"""
import os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_schnorr_batch

n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 20)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
engs = [S.Engine(0), S.Engine(0)]
pk, msgs, sig = synth_schnorr_batch(engs[0], n, min(n, 1 << 16), seed=340)
rng = np.random.default_rng(3)
k = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); k[:, 0] &= 0x7F
d = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); d[:, 0] &= 0x7F
pts = np.array(engs[0].scalar_base_mult_batch(d))


def pinned(a):
    p = S.pinned_array(a.shape, a.dtype)
    p[...] = a
    return p


from secp256k1_voi_amd.synth import synth_batch
epub, edig, er, es = (np.array(a) for a in synth_batch(engs[0], n, min(n, 1 << 16), seed=5))
erid = np.zeros(n, np.uint8)
sets = [[pinned(x) for x in (pk, msgs, sig, k, pts, edig, er, es, erid)] for _ in range(2)]


def run(kind, nthreads, stagger_ms=0.0):
    def work(j):
        e, (ppk, pm, ps, pkk, pp, pdig, pr, pss, prid) = engs[j], sets[j]
        if j and stagger_ms:
            time.sleep(stagger_ms * 1e-3)       # the second verifier starts half a period later: its transfer beside the first one's kernels
        for _ in range(reps):
            if kind == "rlc":
                assert e.schnorr_batch_verify_rlc(ppk, pm, ps)
            elif kind == "schnorr":
                e.schnorr_verify_batch(ppk, pm, ps)
            elif kind == "recover":
                e.ecdsa_recover_batch(pdig, pr, pss, prid)
            else:
                e.multi_scalar_mult(pkk, pp)
    for j in range(nthreads):
        work_ = (lambda jj: (lambda: work(jj)))(j)
    work(0) if False else None
    ths = [threading.Thread(target=work, args=(j,)) for j in range(nthreads)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    return (time.perf_counter() - t0) * 1e3 / (reps * nthreads)


if os.environ.get("PROBE_TRACE"):                  # for tools/gpu_rlc_overlap_trace.sh: only the staggered phase of one kind
    kind = os.environ["PROBE_TRACE"]
    run(kind, 2)
    one = run(kind, 1)
    print("TRACE_PHASE_START", flush=True)
    print(kind, "staggered", run(kind, 2, stagger_ms=one / 2), "ms per batch; one verifier", one)
    sys.exit(0)
for kind in ("rlc", "msm", "schnorr", "recover"):
    run(kind, 2)                                   # buffers, streams
    one, two = run(kind, 1), run(kind, 2)
    stag = run(kind, 2, stagger_ms=one / 2)
    print("%s: one verifier %.2f ms per batch of 2^%d; two verifiers on two threads %.2f ms per batch (%.2fx), started half a period apart %.2f ms (%.2fx)" %
          (kind, one, n.bit_length() - 1, two, one / two, stag, one / stag), flush=True)

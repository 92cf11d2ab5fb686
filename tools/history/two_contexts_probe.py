#!/usr/bin/env python3
"""Does it pay to keep two batches in flight?  2^20 signatures of 2^16 keys per batch, K batches: one context on one
stream, batch after batch, against two contexts on two streams taking the batches alternately (the device then has the
table phase of one batch to run beside the ladder of the other).  Prints ms per batch for both."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

n, K = 1 << 20, 20
dev = torch.device("cuda", 0)
engs = [S.Engine(0), S.Engine(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
inp = [[torch.from_numpy(x).to(dev) for x in synth_batch(engs[0], n, 1 << 16, seed=40 + j)] for j in range(2)]
out = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]


def run(two):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            j = k & 1 if two else 0
            with torch.cuda.stream(streams[j]):
                out[j].zero_()
                engs[j].ecdsa_verify_batch_device(n, *(x.data_ptr() for x in inp[j]), out[j].data_ptr(), 0, streams[j].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / K
    assert all(int(o.sum()) == n for o in (out if two else out[:1]))
    return dt


for rep in range(2):
    print("one context, one stream:   %.3f ms per batch" % run(False))
    print("two contexts, two streams: %.3f ms per batch" % run(True))

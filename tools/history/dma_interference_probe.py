#!/usr/bin/env python3
"""Do host-to-device transfers slow the kernels down?  Two contexts on two streams verify resident batches alternately
(tools/two_contexts_probe.py) while a third stream copies 160 MiB of pinned host memory to the device once per batch
(what submit / wait does), or does not."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch

n, K = 1 << 20, 24
dev = torch.device("cuda", 0)
engs = [S.Engine(0), S.Engine(0)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
cstream = torch.cuda.Stream()
inp = [[torch.from_numpy(x).to(dev) for x in synth_batch(engs[0], n, 1 << 16, seed=40 + j)] for j in range(2)]
out = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
hsrc = torch.empty(160 << 20, dtype=torch.uint8).pin_memory()
ddst = torch.empty(160 << 20, dtype=torch.uint8, device=dev)
hdst = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()


def run(copies, d2h=False):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            j = k & 1
            if copies:
                with torch.cuda.stream(cstream):
                    ddst.copy_(hsrc, non_blocking=True)
            with torch.cuda.stream(streams[j]):
                out[j].zero_()
                engs[j].ecdsa_verify_batch_device(n, *(x.data_ptr() for x in inp[j]), out[j].data_ptr(), 0, streams[j].cuda_stream)
                if d2h:
                    hdst.copy_(out[j], non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3 / K
    return dt


for rep in range(2):
    print("two contexts, no copies:            %.3f ms per batch" % run(False))
    print("two contexts + 160 MiB H2D per batch: %.3f ms per batch" % run(True))
    print("two contexts + H2D + 1 MiB D2H:       %.3f ms per batch" % run(True, True))

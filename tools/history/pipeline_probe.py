#!/usr/bin/env python3
"""Does double-buffering consecutive batches on two HIP streams (two contexts, two workspaces)
hide the tail of k_verify_fast and the one-wave-per-SIMD scalar preparation?  Prints ms per 2^20
batch for 1 and 2 streams."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    n = 1 << 20
    engs = [S.Engine(0), S.Engine(0)]
    pub, digest, r, s = synth_batch(engs[0], n, 1 << 16, seed=1)
    d = [torch.from_numpy(x).to(dev) for x in (pub, digest, r, s)]
    valid = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    for nstreams in (1, 2, 1, 2):
        for k in range(2):
            engs[k].ecdsa_verify_batch_device(n, *[x.data_ptr() for x in d], valid[k].data_ptr(), 0, streams[k].cuda_stream)
        torch.cuda.synchronize()
        steps = 20
        t0 = time.perf_counter()
        for i in range(steps):
            k = i % nstreams
            engs[k].ecdsa_verify_batch_device(n, *[x.data_ptr() for x in d], valid[k].data_ptr(), 0, streams[k].cuda_stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        assert int(valid[0].sum().item()) == n and int(valid[1].sum().item()) == n
        print(f"streams={nstreams}: {dt * 1e3:.3f} ms per batch")


if __name__ == "__main__":
    main()

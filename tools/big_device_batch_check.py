"""Batches far beyond the bench sizes on the device entry point (2^24 and 2^25 signatures; 16 and 1 and 128 signatures per
key), grouping auto and off: verdicts against a seeded damage mask, time of the second call, memory the context holds."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch
eng = S.Engine(0)
dev = torch.device("cuda", 0)
for lg, nk in ((24, 1 << 20), (24, 1 << 24), (25, 1 << 18)):
    n = 1 << lg
    t0 = time.perf_counter()
    pub, dig, r, s = synth_batch(eng, n, nk, seed=lg)
    mask = np.random.default_rng(lg).random(n) < 1 / 64
    s[mask, 31] ^= 1
    d = [torch.from_numpy(a).to(dev) for a in (pub, dig, r, s)]
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for mode in (S.KEYS_AUTO, S.KEYS_OFF):
        eng.set_key_grouping(mode)
        for rep in range(2):                                  # the first call grows the context's buffers
            out.zero_()
            torch.cuda.synchronize(); t1 = time.perf_counter()
            eng.ecdsa_verify_batch_device(n, *(x.data_ptr() for x in d), out.data_ptr(), 0, st)
            torch.cuda.synchronize(); dt = time.perf_counter() - t1
        v = out.cpu().numpy()
        ok = bool((v == (~mask).astype(np.uint8)).all())
        print(f"n=2^{lg} keys={nk} mode={mode}: ok={ok} {dt*1e3:.1f} ms  stats={eng.key_grouping_stats()} device_bytes={eng.device_bytes(n)/2**30:.1f} GiB", flush=True)
        assert ok
    del d, out
    torch.cuda.empty_cache()

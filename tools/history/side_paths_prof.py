#!/usr/bin/env python3
"""Per-kernel times of the entry points beside the headline at 2^20 items (run under rocprofv3 --kernel-trace --stats):
recovery, BIP-340 per-signature verification, BIP-340 whole batch, multiscalar multiplication, the encoded boundary."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch, synth_schnorr_batch

eng = S.Engine(0, wait_tables=True)      # (the wide generator tables are built in the background: a measurement waits for them)
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
pub, dig, r, s = synth_batch(eng, n, 1 << 16, seed=5)
rid = np.zeros(n, np.uint8)
for rep in range(3):
    t0 = time.perf_counter(); v = eng.ecdsa_verify_batch(pub, dig, r, s); dt = time.perf_counter() - t0
print("ecdsa (host buffers): %.2f ms, valid %d" % (dt * 1e3, int(v.sum())))
for rep in range(3):
    t0 = time.perf_counter(); q, ok = eng.ecdsa_recover_batch(dig, r, s, rid); dt = time.perf_counter() - t0
print("recover (host buffers): %.2f ms, ok %d" % (dt * 1e3, int(ok.sum())))
pk, msgs, sig = synth_schnorr_batch(eng, n, 1 << 16, seed=9)
for rep in range(3):
    t0 = time.perf_counter(); v = eng.schnorr_verify_batch(pk, msgs, sig); dt = time.perf_counter() - t0
print("schnorr per signature (host buffers): %.2f ms, valid %d" % (dt * 1e3, int(v.sum())))
for rep in range(3):
    t0 = time.perf_counter(); okb = eng.schnorr_batch_verify_rlc(pk, msgs, sig); dt = time.perf_counter() - t0
print("schnorr whole batch (host buffers): %.2f ms, %s" % (dt * 1e3, okb))

#!/usr/bin/env python3
"""All GPUs of a node from ONE process through the C-ABI's device group (s2k_group_*): what a cgo host gets.

    python tools/group_bench.py [--devices 0,1,...] [--batch-log2 20] [--keys-log2 16] [--batches 16] [--keyset] [--whole-batch]

Every listed device gets 2^batch_log2 signatures per group batch (weak scaling, like bench.py --gpus N: 2^21 per GPU is BASELINE
config 5's shape); the packed arrays live in page-locked host memory, four group batches are kept in flight, and the clock runs
from the completion of the fourth batch to the completion of the last (steady state).  Prints ONE JSON line: whole-node
verifications/s host to host, per-member times of the last shards, and the verdict check (every batch carries a seeded
pattern of damaged signatures that must come back as exactly that pattern).  --keyset: the keys' tables are built once on
every device (s2k_group_keyset_create, joint tables; not timed) and the batches name their keys by index
(s2k_group_ecdsa_verify_batch_keyset_submit).  --whole-batch: BASELINE configs 3 and 4 instead - 2^batch_log2 terms / BIP-340
signatures per device and call through s2k_group_multi_scalar_mult (result checked against the known answer) and
s2k_group_schnorr_batch_verify_rlc (accepts; rejects with one bad signature), synchronous calls from pageable memory.  No torch,
no RCCL.  On this pool only one device
per box exists; `--devices 0` is what has been run."""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def member_lines(g, bytes_per_signature):
    """per member: its last shard, where its device hangs (NUMA node, CPUs its thread is bound to), and the transfer / compute
    split of the shard on the device's clock with the host-to-device rate that follows"""
    out = []
    for s in g.member_stats_ex():
        s = dict(s)
        s["h2d_GBps"] = (s["n"] * bytes_per_signature / (s["h2d_ms"] * 1e-3) / 1e9) if s["h2d_ms"] > 0 else None
        s["after_copies_ms"] = s["device_ms"] - s["h2d_ms"] if s["device_ms"] else None
        out.append(s)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--devices", default=None)
    ap.add_argument("--batch-log2", type=int, default=20)
    ap.add_argument("--keys-log2", type=int, default=16)
    ap.add_argument("--batches", type=int, default=16)
    ap.add_argument("--keyset", action="store_true")
    ap.add_argument("--whole-batch", action="store_true")
    a = ap.parse_args()
    ndev = S.device_count()
    devices = [int(x) for x in a.devices.split(",")] if a.devices else list(range(ndev))
    if not devices:
        print("group_bench.py: no GPU", file=sys.stderr)
        return 2
    per = 1 << a.batch_log2
    n = per * len(devices)
    if a.whole_batch:
        return whole_batch(devices, per, n, a)
    # one shard's worth of signatures is synthesised (on device 0, by a temporary engine) and tiled: every member gets the same
    # keys, which is what per-shard key grouping sees anyway
    eng = S.Engine(devices[0])
    base = [np.array(x) for x in synth_batch(eng, per, min(per, 1 << a.keys_log2), seed=0x5EC9)]
    eng.close()
    del eng
    lead, depth = 4, 4
    g = S.Group(devices)
    gt_bits = g.gt_wait()          # the wide generator tables are built in the background: a benchmark waits for them
    g.member_stats_ex()            # placement and per-ticket timing on from the first shard
    bufs, outs, masks = [], [], []
    for k in range(depth):
        # the group's own page-locked blocks: every member's shard range lies on the NUMA node of its device (s2k_group_host_alloc)
        pb = [g.host_alloc(n, x.shape[1]) for x in base]
        for d, x in zip(pb, base):
            for m in range(len(devices)):
                d[m * per:(m + 1) * per] = x
        i = np.arange(n, dtype=np.uint64)
        bad = ((i * np.uint64(2654435761) + np.uint64(k * 7919 + 5)) % np.uint64(61)) == 0
        pb[3][bad, 31] ^= 1
        bufs.append(pb)
        masks.append((~bad).astype(np.uint8))
        outs.append(g.host_alloc(n, 1).reshape(-1))
    gks = None
    if a.keyset:
        keys, inv = np.unique(base[0], axis=0, return_inverse=True)
        gks = g.keyset_create(keys, S.KEYSET_JOINT)
        kx = S.pinned_array((n,), np.uint32)
        for m in range(len(devices)):
            kx[m * per:(m + 1) * per] = inv.reshape(-1).astype(np.uint32)

    def submit(k):
        if gks is not None:
            return g.ecdsa_verify_batch_keyset_submit(gks, kx, bufs[k % depth][1], bufs[k % depth][2], bufs[k % depth][3], out=outs[k % depth])
        return g.ecdsa_verify_batch_submit(*bufs[k % depth], out=outs[k % depth])
    tickets, done, t0 = [], 0, None
    total = a.batches + lead
    for k in range(total):
        outs[k % depth][...] = 9
        tickets.append((k, submit(k)))
        if len(tickets) >= depth:
            j, t = tickets.pop(0)
            assert np.array_equal(t.wait(), masks[j % depth]), "group verdicts differ from the damage pattern (batch %d)" % j
            done += 1
            if done == lead:
                t0 = time.perf_counter()
    for j, t in tickets:
        assert np.array_equal(t.wait(), masks[j % depth]), "group verdicts differ from the damage pattern (batch %d)" % j
        done += 1
        if done == lead:
            t0 = time.perf_counter()
    dt = time.perf_counter() - t0
    ms = dt * 1e3 / (done - lead)
    line = {"metric": "secp256k1 ECDSA verifications/sec, host to host, one process, s2k_group over %d device(s)" % len(devices),
            "value": n / (ms * 1e-3), "unit": "verifications/s", "n_gpus": len(devices), "devices": devices,
            "ms_per_group_batch": ms, "signatures_per_group_batch": n, "per_gpu_value": per / (ms * 1e-3),
            "batches_timed": done - lead, "in_flight": depth, "scaling": "weak", "data": "synthetic, page-locked host memory",
            "gt_bits": gt_bits, "member_stats_last_shard": member_lines(g, 100 if gks is not None else 160),
            "keyset": None if gks is None else {"keys": len(gks), "device_bytes_per_member": gks.device_bytes(), "layout": "joint tables"},
            "check": "every batch's verdicts equal its seeded damage pattern (one bit of s flipped in every 61st signature)"}
    if gks is not None:
        gks.close()
    g.close()
    print(json.dumps(line), flush=True)
    return 0


def whole_batch(devices, per, n, a):
    """configs 3 and 4 through the group: one shard's worth of inputs synthesised on device 0 and tiled over the members"""
    from secp256k1_voi_amd.synth import synth_schnorr_batch
    eng = S.Engine(devices[0])
    rng = np.random.default_rng(7)
    d = rng.integers(0, 256, size=(per, 32), dtype=np.uint8)
    d[:, 0] &= 0x7F
    k = rng.integers(0, 256, size=(per, 32), dtype=np.uint8)
    k[:, 0] &= 0x7F
    pts = np.array(eng.scalar_base_mult_batch(d))
    kd, _ = eng.fn_op_batch(S.OP_MUL, k, d)                  # k_i d_i mod n
    acc = np.zeros((1, 32), np.uint8)
    tot = 0
    N_ = int("FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141", 16)
    for row in kd:
        tot = (tot + int.from_bytes(bytes(row), "big")) % N_
    tot = tot * len(devices) % N_
    expect = bytes(np.array(eng.scalar_base_mult_batch(np.frombuffer(tot.to_bytes(32, "big"), np.uint8).reshape(1, 32)))[0])
    pk, msgs, sig = (np.array(x) for x in synth_schnorr_batch(eng, per, min(per, 1 << a.keys_log2), seed=340))
    eng.close()
    del eng
    K, P = np.tile(k, (len(devices), 1)), np.tile(pts, (len(devices), 1))
    PK, M, SG = np.tile(pk, (len(devices), 1)), np.tile(msgs, (len(devices), 1)), np.tile(sig, (len(devices), 1))
    g = S.Group(devices)
    assert g.multi_scalar_mult(K, P) == expect, "group multiscalar multiplication: wrong sum"
    reps = max(a.batches // 4, 3)
    t0 = time.perf_counter()
    for _ in range(reps):
        g.multi_scalar_mult(K, P)
    msm_ms = (time.perf_counter() - t0) * 1e3 / reps
    assert g.schnorr_batch_verify_rlc(PK, M, SG), "group BIP-340 batch check rejected a valid batch"
    t0 = time.perf_counter()
    for _ in range(reps):
        g.schnorr_batch_verify_rlc(PK, M, SG)
    rlc_ms = (time.perf_counter() - t0) * 1e3 / reps
    SG[n - 1, 40] ^= 1
    assert not g.schnorr_batch_verify_rlc(PK, M, SG), "group BIP-340 batch check accepted a bad signature"
    g.close()
    print(json.dumps({"metric": "BASELINE configs 3 and 4 through s2k_group over %d device(s), host (pageable) buffers, synchronous calls" % len(devices),
                      "n_gpus": len(devices), "devices": devices, "per_gpu": per,
                      "msm": {"terms": n, "ms": msm_ms, "terms_per_s": n / (msm_ms * 1e-3), "check": "sum equals (sum k_i d_i) G"},
                      "schnorr_rlc": {"sigs": n, "ms": rlc_ms, "sigs_per_s": n / (rlc_ms * 1e-3),
                                      "check": "accepts the valid batch, rejects it with the last signature damaged"},
                      "data": "synthetic"}), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""Host-boundary throughput: the synchronous host-pointer call against submit / wait with two batches in flight, from
pinned and from pageable memory; the encoded entry point the same way; a group of one and of two members on device 0.
python tools/boundary_probe.py [batch_log2] [keys_log2] [batches] -> one JSON line."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import secp256k1_voi_amd as S
from secp256k1_voi_amd.synth import synth_batch


def pipelined(submit, nb, depth=2, lead=3):
    """ms per batch in steady state: `lead` batches fill the pipeline (the first transfer has nothing to hide behind), then
    the clock runs from the completion of batch `lead` to the completion of batch `lead + nb`; `depth` in flight."""
    tickets, done, t0 = [], 0, None
    for k in range(nb + lead):
        tickets.append(submit(k))
        if len(tickets) >= depth:
            tickets.pop(0).wait()
            done += 1
            if done == lead:
                t0 = time.perf_counter()
    for t in tickets:
        t.wait()
        done += 1
        if done == lead:
            t0 = time.perf_counter()
    return (time.perf_counter() - t0) * 1e3 / (done - lead)


def main():
    lg = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    kl = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    nb = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    n = 1 << lg
    eng = S.Engine(0)
    base = [np.array(x) for x in synth_batch(eng, n, min(n, 1 << kl), seed=0x5EC9)]
    pin = [[S.pinned_array(a.shape) for a in base] for _ in range(4)]
    for pb in pin:
        for d, a in zip(pb, base):
            d[...] = a
    outs = [S.pinned_array((n,)) for _ in range(4)]
    res = {"n_log2": lg, "keys_log2": kl, "batches": nb, "queues": os.environ.get("GPU_MAX_HW_QUEUES")}

    def sync_ms(arrs, reps=4):
        eng.ecdsa_verify_batch(*arrs)
        ms = []
        for _ in range(reps):
            t0 = time.perf_counter()
            v = eng.ecdsa_verify_batch(*arrs)
            ms.append((time.perf_counter() - t0) * 1e3)
            assert int(v.sum()) == n
        return sorted(ms)[len(ms) // 2]

    res["sync_pinned_ms"] = sync_ms(pin[0])
    res["sync_pageable_ms"] = sync_ms(base)
    # submit / wait
    for name, bufs, o in (("pinned", pin, outs), ("pageable", [base] + [[a.copy() for a in base] for _ in range(3)], [None] * 4)):
        pipelined(lambda k: eng.ecdsa_verify_batch_submit(*bufs[k % 4], out=o[k % 4]), 8, 4)      # creates the slots
        for depth in (2, 3, 4):
            ms = [pipelined(lambda k: eng.ecdsa_verify_batch_submit(*bufs[k % 4], out=o[k % 4]), nb, depth) for _ in range(3)]
            res["pipelined_%s_depth%d_ms" % (name, depth)] = sorted(ms)[1]
            res["pipelined_%s_depth%d_ms_all" % (name, depth)] = ms
    assert int(outs[0].sum()) == n and int(outs[1].sum()) == n
    # depth 1 = submit immediately followed by wait (the synchronous call on a child context)
    res["submit_wait_depth1_pinned_ms"] = pipelined(lambda k: eng.ecdsa_verify_batch_submit(*pin[k % 4], out=outs[k % 4]), nb, depth=1)
    # encoded
    if len(sys.argv) <= 4:
        def der_int(b):
            b = bytes(b).lstrip(b"\0") or b"\0"
            if b[0] & 0x80:
                b = b"\0" + b
            return b"\x02" + bytes([len(b)]) + b
        sigs, pubs = [], []
        for i in range(n):
            body = der_int(base[2][i]) + der_int(base[3][i])
            sigs.append(b"\x30" + bytes([len(body)]) + body)
            pubs.append(b"\x04" + bytes(base[0][i]))
        cat = [S._concat(pubs), S._concat([bytes(d) for d in base[1]]), S._concat(sigs)]
        pcat = []
        for blob, offs in cat:
            pb, po = S.pinned_array(blob.shape), S.pinned_array(offs.shape, np.uint64)
            pb[...] = blob
            po[...] = offs
            pcat.append((pb, po))
        for name, c in (("pageable", cat), ("pinned", pcat)):
            t0 = time.perf_counter()
            v = eng.ecdsa_verify_encoded_batch_submit(*c, digest_len=32).wait()
            assert int(v.sum()) == n
            for depth in (2, 4):
                ms = [pipelined(lambda k: eng.ecdsa_verify_encoded_batch_submit(*c, digest_len=32), nb, depth) for _ in range(3)]
                res["encoded_pipelined_%s_depth%d_ms" % (name, depth)] = sorted(ms)[1]
        out = np.zeros(n, np.uint8)
        ms = []
        for _ in range(3):
            t0 = time.perf_counter()
            eng._check(eng._lib.s2k_ecdsa_verify_encoded_batch(eng._h, n, cat[0][0].ctypes.data, cat[0][1].ctypes.data, cat[1][0].ctypes.data,
                                                               cat[1][1].ctypes.data, cat[2][0].ctypes.data, cat[2][1].ctypes.data, 0, 32, 0,
                                                               out.ctypes.data))
            ms.append((time.perf_counter() - t0) * 1e3)
        res["encoded_sync_pageable_ms"] = sorted(ms)[1]
    # groups on device 0
    for members in (1, 2):
        g = S.Group([0] * members)
        big = [np.concatenate([a] * members) for a in base] if members > 1 else base
        pb = [[S.pinned_array(a.shape) for a in big] for _ in range(4)]
        for q in pb:
            for d, a in zip(q, big):
                d[...] = a
        po = [S.pinned_array((n * members,)) for _ in range(4)]
        pipelined(lambda k: g.ecdsa_verify_batch_submit(*pb[k % 4], out=po[k % 4]), 8, 4)
        for depth in (2, 4):
            ms = [pipelined(lambda k: g.ecdsa_verify_batch_submit(*pb[k % 4], out=po[k % 4]), nb, depth) for _ in range(3)]
            res["group_%d_member_pipelined_depth%d_ms_per_2p%d" % (members, depth, lg)] = sorted(ms)[1] / members
        assert int(po[0].sum()) == n * members
        res["group_%d_member_stats" % members] = g.member_stats()
        g.close()
        del pb, po
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

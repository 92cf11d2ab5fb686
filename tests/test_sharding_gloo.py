"""world_size-2 test of the N > 1 path on CPU (gloo): shard, verify each shard, gather the
bitmap.  The per-shard verifier here is the oracle standing in for the GPU engine (there is
no GPU in this container); the sharding / collective code is the product's.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from secp256k1_voi_amd.sharding import pack_bitmap, shard_range, unpack_bitmap


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 9, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            got = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                got += list(range(lo, hi)) if n < 2000 else []
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
            if n < 2000:
                assert got == list(range(n))
            assert shard_range(n, world - 1, world)[1] == n


def test_bitmap_roundtrip():
    rng = np.random.default_rng(1)
    for n in (0, 1, 8, 13, 1000):
        v = rng.integers(0, 2, n).astype(np.uint8)
        assert (unpack_bitmap(pack_bitmap(v), n) == v).all()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, seed, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    import oracle
    from secp256k1_voi_amd.sharding import gather_valid, shard_range
    from workload import make_ecdsa_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w = make_ecdsa_batch(oracle, n, seed=seed, corrupt_every=4)      # same batch on every rank
    lo, hi = shard_range(n, rank, world)
    mine = oracle.ecdsa_verify_batch(w["pub"][lo:hi], w["digest"][lo:hi], w["r"][lo:hi], w["s"][lo:hi])
    full, count = gather_valid(mine, n, dist)
    q.put((rank, full.tobytes(), count))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n", [101, 256])
def test_two_rank_gather(n, oracle):
    from workload import make_ecdsa_batch
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, 5, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    w = make_ecdsa_batch(oracle, n, seed=5, corrupt_every=4)
    exp = oracle.ecdsa_verify_batch(w["pub"], w["digest"], w["r"], w["s"])
    for rank, full, count in res:
        assert np.frombuffer(full, dtype=np.uint8).tolist() == exp.tolist()
        assert count == int(exp.sum())


class _OracleEngine:
    """Stands in for the GPU Engine in the CPU collective tests (same method contracts)."""

    def __init__(self, oracle):
        self.o = oracle

    def multi_scalar_mult(self, scalars, points):
        return self.o.multi_scalar_mult_vartime(list(scalars), list(points))

    def schnorr_batch_verify_rlc(self, pks, msgs, sigs, seed32=None):
        return all(self.o.schnorr_verify(pk, m, s) == 1 for pk, m, s in zip(pks, msgs, sigs))


def _worker_msm(rank, world, port, n, q):
    import random
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    import torch.distributed as dist
    import oracle
    import pyref
    from secp256k1_voi_amd.sharding import msm_sharded, schnorr_batch_verify_sharded, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rnd = random.Random(77)
    ks = [rnd.randrange(pyref.N).to_bytes(32, "big") for _ in range(n)]
    ps = [oracle.scalar_base_mult_vartime(rnd.randrange(1, pyref.N).to_bytes(32, "big")) for _ in range(n)]
    lo, hi = shard_range(n, rank, world)
    eng = _OracleEngine(oracle)
    total = msm_sharded(eng, ks[lo:hi], ps[lo:hi], dist)
    # BIP-340 shards: rank 1's shard carries one bad signature in the second run
    sk = [rnd.randrange(1, pyref.N) for _ in range(n)]
    msgs = [rnd.randbytes(32) for _ in range(n)]
    sigs, pks = [], []
    for d, m in zip(sk, msgs):
        pks.append(pyref.b32(pyref.mul(d, pyref.G)[0]))
        sigs.append(pyref.schnorr_sign(d, m, bytes(32)))
    ok_all = schnorr_batch_verify_sharded(eng, pks[lo:hi], msgs[lo:hi], sigs[lo:hi], b"seed", dist)
    bad = list(sigs)
    bad[n - 1] = bad[n - 1][:63] + bytes([bad[n - 1][63] ^ 1])
    ok_bad = schnorr_batch_verify_sharded(eng, pks[lo:hi], msgs[lo:hi], bad[lo:hi], b"seed", dist)
    q.put((rank, total, ok_all, ok_bad))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_msm_and_schnorr_batch(oracle):
    import random
    import pyref
    n, world, port = 24, 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_msm, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rnd = random.Random(77)
    ks = [rnd.randrange(pyref.N).to_bytes(32, "big") for _ in range(n)]
    ps = [oracle.scalar_base_mult_vartime(rnd.randrange(1, pyref.N).to_bytes(32, "big")) for _ in range(n)]
    exp = oracle.multi_scalar_mult_vartime(ks, ps)
    for rank, total, ok_all, ok_bad in res:
        assert total == exp
        assert ok_all is True and ok_bad is False


def _worker_device_form(rank, world, port, n, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch
    import torch.distributed as dist
    from secp256k1_voi_amd.sharding import gather_valid_device
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scratch = {}
    out = []
    for rep in range(2):                                   # second call: the kept buffers are reused
        v = torch.from_numpy(np.random.default_rng(100 * rep + rank).integers(0, 2, n).astype(np.uint8))
        bitmap, count = gather_valid_device(v, n * world, dist, scratch=scratch)
        out.append((bitmap.numpy().tobytes(), int(count.item())))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_device_form_one_collective():
    """gather_valid_device (the hot-path form: equal shards, ONE all_gather_into_tensor carrying every rank's bitmap shard
    with its valid count behind it) in two gloo ranks on CPU tensors with the torch packing: every rank ends with the
    concatenated bitmaps and the sum of the counts.  The nccl backend runs the same statements on device buffers."""
    world, port, n = 2, _free_port(), 1024
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_device_form, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rep in range(2):
        shards = [np.random.default_rng(100 * rep + r).integers(0, 2, n).astype(np.uint8) for r in range(world)]
        exp_bits = b"".join(pack_bitmap(v).tobytes() for v in shards)
        exp_count = int(sum(int(v.sum()) for v in shards))
        for r in range(world):
            assert res[r][rep] == (exp_bits, exp_count)

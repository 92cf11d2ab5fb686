"""Multi-GPU sharding of the batch path.

Signatures are independent, so a batch is cut into contiguous index ranges, one per rank
(one process per GPU); there is no exchange during compute.  The only collectives are at
the end: one all-gather of the per-rank valid-bitmap shards with the rank's valid count behind
each (the host-side helper gather_valid keeps the two-collective form).  Works on any torch.distributed backend (nccl = RCCL over xGMI on the GPU
box, gloo in the CPU tests).
"""
from __future__ import annotations

import os

import numpy as np


def _alone(dist) -> bool:
    """No process group, or a group of one: nothing to exchange.  S2K_FORCE_COLLECTIVES=1 (test hook) sends a
    group of one through the collectives anyway, which is how the single-GPU box exercises the RCCL calls."""
    if dist is None or not dist.is_initialized():
        return True
    return dist.get_world_size() == 1 and os.environ.get("S2K_FORCE_COLLECTIVES") != "1"


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) of rank `rank` among `world` ranks; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_bitmap(valid: np.ndarray) -> np.ndarray:
    """0/1 bytes -> bitmap, bit i of byte i//8 (LSB first), zero padded."""
    return np.packbits(np.asarray(valid, dtype=np.uint8), bitorder="little")


def unpack_bitmap(bitmap: np.ndarray, n: int) -> np.ndarray:
    return np.unpackbits(np.asarray(bitmap, dtype=np.uint8), bitorder="little")[:n]


def gather_valid(valid_shard, n_total: int, dist=None, device=None):
    """All ranks get the full 0/1 vector (length n_total) and the global valid count.

    `valid_shard` is this rank's 0/1 uint8 vector (numpy array or torch tensor) for
    shard_range(n_total, rank, world).  Collectives: one all_gather of equal-sized, zero
    padded bitmap shards and one all_reduce of the count.
    """
    import torch

    if _alone(dist):
        v = valid_shard.cpu().numpy() if hasattr(valid_shard, "cpu") else np.asarray(valid_shard)
        return v.astype(np.uint8), int(v.sum())
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_total, rank, world)
    v = valid_shard.cpu().numpy() if hasattr(valid_shard, "cpu") else np.asarray(valid_shard)
    assert v.shape[0] == hi - lo, "shard length does not match shard_range"
    per = (n_total + world - 1) // world           # max shard length
    nbytes = (per + 7) // 8
    buf = np.zeros(nbytes, dtype=np.uint8)
    pb = pack_bitmap(v)
    buf[: pb.size] = pb
    dev = device if device is not None else torch.device("cpu")
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    cnt = torch.tensor([int(v.sum())], dtype=torch.int64, device=dev)
    dist.all_reduce(cnt)
    out = np.zeros(n_total, dtype=np.uint8)
    for r, p in enumerate(parts):
        a, b = shard_range(n_total, r, world)
        out[a:b] = unpack_bitmap(p.cpu().numpy(), b - a)
    return out, int(cnt.item())


def gather_valid_device(valid, n_total: int, dist=None, engine=None, bitmap=None, count=None, scratch=None, events=None):
    """Device-resident variant used on the hot path: `valid` is this rank's uint8 0/1 torch
    tensor (equal shard length on every rank; a multiple of 8, in a group of 64).  Packs the bitmap on the
    device (engine.pack_valid_device when an Engine is given, torch ops otherwise) and exchanges it
    in ONE collective: the rank's valid count rides behind its bitmap shard in the same buffer
    (all_gather_into_tensor of n/8 + 8 bytes per rank), and every rank adds up the gathered counts
    itself - a second collective (an all_reduce of one int64) cost as much as the first: both are
    launch latency, about 0.13 ms each per step through RCCL on MI355X.  Returns (bitmap tensor of
    n_total/8 bytes, int64 count tensor of one element).  No host round trip on the nccl backend.
    `bitmap` / `count` (alone) or `scratch` (a dict the caller keeps between calls; in a group) hold
    preallocated buffers to keep the step allocation-free.  The returned bitmap ALIASES `bitmap` / the scratch buffer: it is
    overwritten by the next call with the same buffers (copy it if it has to outlive the step); without `scratch` a call in
    a group allocates its three buffers afresh.  `events`: a pair of torch.cuda.Event recorded on the current stream right
    before and right after the collective (bench.py's multi-rank diagnostics)."""
    import torch

    n = valid.numel()
    if n % 8:
        raise ValueError("gather_valid_device: the shard length must be a multiple of 8")
    nb = n // 8
    if _alone(dist):
        if bitmap is None:
            bitmap = torch.empty(nb, dtype=torch.uint8, device=valid.device)
        if count is None:
            count = torch.zeros(1, dtype=torch.int64, device=valid.device)
        _pack(valid, bitmap, count, engine)
        return bitmap, count
    world = dist.get_world_size()
    if n * world != n_total:
        raise ValueError("gather_valid_device needs equal shards")
    if n % 64:
        raise ValueError("in a group the shard length must be a multiple of 64 (the count sits 8-byte aligned behind the bitmap)")
    scratch = scratch if scratch is not None else {}
    key = ("packed", n, world, str(valid.device))
    if scratch.get("key") != key:
        scratch["key"] = key
        scratch["packed"] = torch.zeros(nb + 8, dtype=torch.uint8, device=valid.device)
        scratch["full"] = torch.empty((nb + 8) * world, dtype=torch.uint8, device=valid.device)
        scratch["bitmap"] = torch.empty(nb * world, dtype=torch.uint8, device=valid.device)
    packed = scratch["packed"]
    _pack(valid, packed[:nb], packed[nb:].view(torch.int64), engine)
    # One code path for both backends: the same collective on the same shape; the only difference is where the
    # tensors live (gloo moves bytes between host buffers, nccl = RCCL between device buffers over xGMI).  What the
    # 2-rank tests exercise on one GPU with gloo is therefore the logic an 8-GPU RCCL run executes.
    on_host = dist.get_backend() == "gloo"
    mine = packed.cpu() if on_host else packed
    full = torch.empty(mine.numel() * world, dtype=torch.uint8) if on_host else scratch["full"]
    if events is not None:
        events[0].record()
    dist.all_gather_into_tensor(full, mine)
    if on_host:
        full = full.to(valid.device)
    if events is not None:
        events[1].record()
    rows = full.view(world, nb + 8)
    out_bitmap = bitmap if (bitmap is not None and bitmap.numel() == nb * world) else scratch["bitmap"]
    out_bitmap.view(world, nb).copy_(rows[:, :nb])                           # (the shards are nb + 8 bytes apart in `full`)
    out_count = rows[:, nb:].contiguous().view(torch.int64).sum().reshape(1)
    if count is not None:
        count.copy_(out_count)
        out_count = count
    return out_bitmap, out_count


def _pack(valid, bitmap, count, engine):
    import torch
    n = valid.numel()
    if engine is not None:
        engine.pack_valid_device(n, valid.data_ptr(), bitmap.data_ptr(), count.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    else:
        w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=valid.device)
        bitmap.copy_((valid.view(-1, 8) * w).sum(dim=1, dtype=torch.uint8))
        count.copy_(valid.sum(dtype=torch.int64).reshape(1))


def msm_sharded(engine, scalars, points, dist=None):
    """Sum of s_i * P_i with the terms sharded over the ranks: every rank passes
    ITS shard (lists of 32-byte scalars and 65-byte point records), computes one partial sum
    with `engine.multi_scalar_mult`, the 65-byte partial sums are all-gathered (point addition is
    not a reduction RCCL knows) and folded by one more `multi_scalar_mult` with unit scalars.
    Every rank returns the same 65-byte record (all zero for the identity)."""
    import torch

    part = engine.multi_scalar_mult(scalars, points)
    if _alone(dist):
        return part
    world = dist.get_world_size()
    mine = torch.tensor(list(part), dtype=torch.uint8)
    if dist.get_backend() != "gloo":
        mine = mine.to(torch.device("cuda", engine.device))   # the engine's GPU, not torch's current device
    full = torch.empty(65 * world, dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(full, mine)
    raw = full.cpu().numpy().tobytes()
    recs = [raw[65 * r:65 * r + 65] for r in range(world)]
    one = (1).to_bytes(32, "big")
    return engine.multi_scalar_mult([one] * world, recs)


def schnorr_batch_verify_sharded(engine, pks, msgs, sigs, seed: bytes, dist=None) -> bool:
    """Whole-batch BIP-340 verification with the signatures sharded over the ranks: each rank
    checks its shard as one random-linear-combination multiscalar multiplication
    (`engine.schnorr_batch_verify_rlc`, independent coefficients per rank: the rank is mixed
    into the seed) and the verdicts are combined with an all-reduce (min).  Accepts iff every
    shard accepts, i.e. iff every signature verifies, up to 2^-128 per shard."""
    import hashlib

    import torch

    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    ok = bool(engine.schnorr_batch_verify_rlc(pks, msgs, sigs, hashlib.sha256(seed + rank.to_bytes(4, "big")).digest()))
    if _alone(dist):
        return ok
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if dist.get_backend() != "gloo":
        flag = flag.to(torch.device("cuda", engine.device))
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))

"""One rank of the multi-rank GPU tests (tests/test_gpu_round3.py): a fresh process per rank, all ranks on GPU 0, gloo
collectives - the shape of `bench.py --oversubscribe`.  Runs the sharded entry points of secp256k1_voi_amd.sharding with
the REAL engine and prints one JSON line.

    python multirank_worker.py <rank> <world> <port> <log2 n> [backend]

backend nccl (= RCCL) needs one GPU per rank; on the one-GPU box it runs as a group of ONE with S2K_FORCE_COLLECTIVES=1,
so that the same calls (all_gather_into_tensor of uint8, all_reduce SUM of int64 and MIN of int32, barrier) go through
RCCL on device buffers.
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, log2n = (int(x) for x in sys.argv[1:5])
backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)

import numpy as np
import torch
import torch.distributed as dist

import secp256k1_voi_amd as S
from secp256k1_voi_amd.sharding import gather_valid_device, msm_sharded, schnorr_batch_verify_sharded, shard_range
from secp256k1_voi_amd.synth import synth_batch, synth_msm_terms, synth_schnorr_batch

dev = torch.device("cuda", 0)
if backend == "nccl":
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group(backend, rank=rank, world_size=world)
eng = S.Engine(0)
eng.set_mid_batch_max(0)     # (2^14 signatures per rank would take the four-lanes-per-signature ladder; this worker reports the grouped path's statistics)
n = 1 << log2n
out = {"rank": rank, "backend": dist.get_backend()}

# ---- multi-scalar multiplication: every rank builds the same terms (seeded), takes its shard ----
k, pts, tot = synth_msm_terms(eng, n, seed=11)
lo, hi = shard_range(n, rank, world)
got = msm_sharded(eng, k[lo:hi], pts[lo:hi], dist)
want = eng.scalar_base_mult_batch([int(tot).to_bytes(32, "big")])[0].tobytes()
out["msm_ok"] = got == want
out["msm_sum"] = hashlib.sha256(got).hexdigest()[:16]

# ---- BIP-340 whole-batch check, sharded: all valid, then one bad signature in the LAST rank's shard ----
pk, msgs, sig = synth_schnorr_batch(eng, n, max(n >> 4, 1), seed=12)
out["schnorr_all_valid"] = schnorr_batch_verify_sharded(eng, pk[lo:hi], msgs[lo:hi], sig[lo:hi], b"multirank", dist)
bad = sig.copy()
bad[n - 3, 40] ^= 0x10
out["schnorr_one_bad"] = schnorr_batch_verify_sharded(eng, pk[lo:hi], msgs[lo:hi], bad[lo:hi], b"multirank", dist)

# ---- ECDSA verification shards: device-resident flow, bitmap all-gather + count all-reduce ----
per = n                                                     # equal shards of n signatures per rank
pub, dig, r, s = (np.array(a) for a in synth_batch(eng, per, max(per >> 4, 1), seed=100 + rank))
rng = np.random.default_rng(1000 + rank)
badidx = np.sort(rng.choice(per, size=max(per // 50, 1), replace=False))
r[badidx, 31] ^= 1
d = [torch.from_numpy(a).to(dev) for a in (pub, dig, r, s)]
valid = torch.zeros(per, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream
eng.ecdsa_verify_batch_device(per, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), valid.data_ptr(), 0, st)
bitmap, count = gather_valid_device(valid, per * world, dist, eng)
bits = np.unpackbits(bitmap.cpu().numpy(), bitorder="little")
exp = np.ones(per * world, dtype=np.uint8)
for rk in range(world):                                      # every rank can rebuild every rank's pattern
    bi = np.sort(np.random.default_rng(1000 + rk).choice(per, size=max(per // 50, 1), replace=False))
    exp[rk * per + bi] = 0
out["ecdsa_bitmap_ok"] = bool(np.array_equal(bits, exp))
out["ecdsa_count_ok"] = int(count.item()) == int(exp.sum())
st_ = eng.key_grouping_stats()
out["keyed"] = st_["keyed"]
print(json.dumps(out), flush=True)
dist.barrier()
dist.destroy_process_group()

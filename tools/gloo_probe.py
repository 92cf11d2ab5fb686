#!/usr/bin/env python3
"""Two processes on one box, gloo: where the time of an all-gather of host buffers goes, by payload.
Prints per payload the D2H copy, the collective and the H2D copy (ms, median of 7) as rank 0 sees them."""
import os, sys, time, statistics
import torch, torch.distributed as dist
import torch.multiprocessing as mp

def main(rank, world, port, use_gpu):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0) if use_gpu else torch.device("cpu")
    for kib in (8, 16, 32, 64, 256):
        nb = kib * 1024 + 8
        src = torch.zeros(nb, dtype=torch.uint8, device=dev)
        rows = {k: [] for k in ("d2h", "gather", "gather_list", "h2d", "gather_pinned", "gather_clone", "gather_fresh_host")}
        pin_mine = torch.empty(nb, dtype=torch.uint8, pin_memory=True) if use_gpu else torch.empty(nb, dtype=torch.uint8)
        pin_full = torch.empty(nb * world, dtype=torch.uint8, pin_memory=True) if use_gpu else torch.empty(nb * world, dtype=torch.uint8)
        for it in range(9):
            dist.barrier()
            t0 = time.perf_counter(); mine = src.cpu(); t1 = time.perf_counter()
            full = torch.empty(nb * world, dtype=torch.uint8)
            dist.all_gather_into_tensor(full, mine); t2 = time.perf_counter()
            parts = [torch.empty(nb, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(parts, mine); t3 = time.perf_counter()
            back = full.to(dev)
            if use_gpu: torch.cuda.synchronize()
            t4 = time.perf_counter()
            pin_mine.copy_(src)
            if use_gpu: torch.cuda.synchronize()
            t5 = time.perf_counter(); dist.all_gather_into_tensor(pin_full, pin_mine); t6 = time.perf_counter()
            c = mine.clone(); cf = torch.empty(nb * world, dtype=torch.uint8)
            t7 = time.perf_counter(); dist.all_gather_into_tensor(cf, c); t8 = time.perf_counter()
            h = torch.zeros(nb, dtype=torch.uint8); hf = torch.empty(nb * world, dtype=torch.uint8)    # never seen by the HIP runtime
            t9 = time.perf_counter(); dist.all_gather_into_tensor(hf, h); t10 = time.perf_counter()
            if it >= 2:
                rows["gather_pinned"].append(t6 - t5); rows["gather_clone"].append(t8 - t7); rows["gather_fresh_host"].append(t10 - t9)
                rows["d2h"].append(t1 - t0); rows["gather"].append(t2 - t1); rows["gather_list"].append(t3 - t2); rows["h2d"].append(t4 - t3)
        if rank == 0:
            print(f"{kib:4d} KiB per rank: " + ", ".join(f"{k} {statistics.median(v) * 1e3:8.3f} ms" for k, v in rows.items()), flush=True)
    dist.destroy_process_group()

if __name__ == "__main__":
    use_gpu = len(sys.argv) > 1 and sys.argv[1] == "gpu"
    mp.spawn(main, args=(2, 29631, use_gpu), nprocs=2, join=True)

import time, torch, numpy as np
torch.cuda.init()
h = torch.empty(160 << 20, dtype=torch.uint8).pin_memory()
d = torch.empty(160 << 20, dtype=torch.uint8, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize(); print("H2D 160 MiB pinned: %.2f ms" % ((time.perf_counter() - t0) * 1e3))

#!/usr/bin/env python3
"""Context creation time (generator table build) of the current build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
import secp256k1_voi_amd as S
t0 = time.perf_counter(); e = S.Engine(0); t1 = time.perf_counter()
print("bits", e.generator_window_bits(), "create_s", round(t1 - t0, 3))

#!/usr/bin/env python3
"""Kernel time of the HBM-side stage kernels on one 61,440-ray pass (64 + 128 samples): sample_fine, ray_bias (through dfn_mlp_fine's
profile slot), composite_fine — torch events around back-to-back launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn

dev = "cuda:0"
n, Nc, Ni = 61440, 64, 128
g = torch.Generator(device=dev).manual_seed(0)
sigma = torch.rand(n, Nc, device=dev, generator=g) * 2


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms = timed(lambda: eng.sample_fine(sigma, Ni, 0.0, 2.5))
print("sample_fine      %.4f ms  -> %.2f TB/s of the algorithmic 1024 B/ray" % (ms, n * 1024 / ms / 1e9))
raw = torch.rand(n, Nc + Ni, 9, device=dev, generator=g)
z = torch.sort(torch.rand(n, Nc + Ni, device=dev, generator=g) * 2.5, dim=-1)[0]
ms = timed(lambda: eng.composite_fine(raw, z), 10)
print("composite_fine   %.4f ms  -> %.2f TB/s of the algorithmic 7700 B/ray" % (ms, n * 7700 / ms / 1e9))

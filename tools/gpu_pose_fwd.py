#!/usr/bin/env python3
"""Pose-only DFNet forward (the pose regressor at test time: feature/dfnet.py:168-170 after the encoder) at batch 1 and 4, 480x640: ms per call."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dfnet_amd import engine as eng, synthetic as syn
E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
out = []
for B in (1, 4):
    x = torch.rand(B, 3, 480, 640, device="cuda:0")
    for _ in range(5): E.forward(x, False, True, True, 480, 640)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): E.forward(x, False, True, True, 480, 640)
    torch.cuda.synchronize(); out.append("B=%d %.4f ms" % (B, (time.perf_counter() - t0) / 50 * 1e3))
print("pose-only forward:", ", ".join(out))

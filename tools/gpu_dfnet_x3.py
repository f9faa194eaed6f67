#!/usr/bin/env python3
"""Split-f16 DFNet forward only (B x 480x640, features): ms per image (A/B aid: compare builds via DFN_LIB_PATH)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.rand(B, 3, 480, 640, device="cuda:0")
for _ in range(3):
    E.forward(x, True, True, False, 480, 640, precision="f16x3")
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    E.forward(x, True, True, False, 480, 640, precision="f16x3")
torch.cuda.synchronize()
print({k: v for k, v in os.environ.items() if k.startswith("DFN_")}, "ms/img %.4f" % ((time.time() - t0) / 10 / B * 1e3))

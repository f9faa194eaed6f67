#!/usr/bin/env python3
"""DFNet forward timing (BASELINE configs[3] shape: 480x640 frames, features only) + streamed L2 vs oracle."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
dev = "cuda:0"
w = syn.dfnet_weights(3)
E = eng.DfnetEngine(3, 12).load_numpy(w)
out = {}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.rand(B, 3, 480, 640, device=dev)
for prec in ("f16", "f16x3", "f32"):
    E.forward(x, True, True, False, 480, 640, precision=prec)
    torch.cuda.synchronize()
    n = 2 if prec == "f32" else 5
    t0 = time.time()
    for _ in range(n):
        E.forward(x, True, True, False, 480, 640, precision=prec)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n / B
    out[f"ms_per_image_{prec}"] = dt * 1e3
    out[f"tflops_{prec}"] = 325.3e9 / dt / 1e12
print(json.dumps(out))

#!/usr/bin/env python3
"""DFNet forward timing (BASELINE configs[3] shape: 480x640 frames, features only) for the three arithmetic modes, and
the relative L2 of the fast modes' feature pyramids against the exact-fp32 path."""
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
# relative L2 of the feature pyramid of the fast modes against the exact-fp32 MFMA path on one 480x640 frame, per level
# (the same figures against the CPU oracle: tests/test_gpu_dfnet.py::test_c4_frame_relative_l2_vs_oracle)
x1 = torch.rand(1, 3, 480, 640, generator=torch.Generator().manual_seed(7)).to(dev)
ref = E.forward(x1, True, True, False, 480, 640, precision="f32")[0].clone()
for prec in ("f16", "f16x3"):
    got = E.forward(x1, True, True, False, 480, 640, precision=prec)[0]
    out[f"rel_l2_per_level_{prec}_vs_fp32_path"] = [float((got[l] - ref[l]).norm() / ref[l].norm()) for l in range(3)]
print(json.dumps(out))

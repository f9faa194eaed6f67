#!/usr/bin/env python3
"""One line per library build (DFN_LIB_PATH): split-f16 DFNet forward time at 4 x 480x640 (features only, HIP events over 30
forwards) and the relative L2 of its three pyramid levels against the exact-fp32 path of the same library on one frame.
   tools/gpu_ab_libs.sh "python tools/gpu_dfnet_ab.py" libdfnet_hip.so libvar_X.so"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
dev = "cuda:0"
E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.rand(B, 3, 480, 640, device=dev)
for _ in range(5):
    E.forward(x, True, True, False, 480, 640, precision="f16x3")
torch.cuda.synchronize()
ts = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        E.forward(x, True, True, False, 480, 640, precision="f16x3")
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10 / B)
x1 = torch.rand(1, 3, 480, 640, generator=torch.Generator().manual_seed(7)).to(dev)
ref = [t.clone() for t in E.forward(x1, True, True, False, 480, 640, precision="f32")[0]]
got = E.forward(x1, True, True, False, 480, 640, precision="f16x3")[0]
rl = [float((got[l] - ref[l]).norm() / ref[l].norm()) for l in range(3)]
print("ms/image %s  rel_l2 %s" % (" ".join(f"{t:.4f}" for t in ts), " ".join(f"{r:.2e}" for r in rl)))

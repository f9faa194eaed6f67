#!/usr/bin/env python3
"""netwidth-256 frame timing on the register-resident kernels (bench.py secondary.nerfh_netwidth_256, f16 only; A/B aid)."""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
dev = torch.device("cuda:0")
cw, fw, ea, et = syn.nerfh_weights(0, W=256)
E = eng.NerfHEngine(width=256).load_numpy(cw, fw, ea, et)
pose, hist = torch.from_numpy(syn.orbit_pose(0, 8)).to(dev), torch.from_numpy(syn.HIST_IDX).to(dev)
for prec in sys.argv[1:] or ["f16"]:
    E.render_image(pose, 480, 640, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): E.render_image(pose, 480, 640, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(prec, f"{307200 / dt / 1e6:.3f} M rays/s  {dt * 1e3:.1f} ms/frame  {325.9e6 * 307200 / dt / 1e12:.0f} TFLOP/s")

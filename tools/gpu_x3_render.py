#!/usr/bin/env python3
"""Frames per second of the fp32-grade render paths (640x480, 64+128): split-f16 and exact fp32 rays/s (A/B aid)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
cw, fw, ea, et = syn.nerfh_weights(0)
E = eng.NerfHEngine(precision="f16x3").load_numpy(cw, fw, ea, et)
c2w = torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4].cuda()
hist = torch.from_numpy(syn.HIST_IDX).cuda()
for prec, n in (("f16x3", 6), ("f16", 10)):
    for _ in range(2):
        E.render_image(c2w, 480, 640, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        E.render_image(c2w, 480, 640, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(prec, "%.3f M rays/s  %.2f ms/frame" % (307200 / dt / 1e6, dt * 1e3))

#!/usr/bin/env python3
"""The N1 step (tools/gpu_nerf_train_step.py) replayed from ONE captured hipGraph against the same step enqueued launch by launch:
is the ~50-launch step bound by launch gaps?  Prints one JSON line."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc

dev = torch.device("cuda:0")
R, Nc, Ni, iters = 1536, 64, 128, 60
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
opt = torch.optim.Adam(tr.params, lr=5e-4, betas=(0.9, 0.999), capturable=True)
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
target = torch.rand(R, 3, device=dev)


def step():
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
    opt.step()


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
eager_ms = timed(step)
out = {"eager_step_ms": eager_ms}
try:
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    out["graph_step_ms"] = timed(g.replay)
    out["eager_again_ms"] = timed(step)
except Exception as e:
    out["graph_error"] = repr(e)[:600]
print(json.dumps(out))

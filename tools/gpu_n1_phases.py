#!/usr/bin/env python3
"""Device-side phase intervals of the N1 step loop (HIP events behind draw / forward / loss / backward / Adam), 1 536 rays at 64 + 128:
prints `<DFN_TRAIN_BWD_PAIR> step <ms> | forward <ms> loss <ms> backward <ms> adam <ms> draw <ms>` (medians over 60 steps)."""
import os, sys, time
import numpy as np
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from dfnet_amd import optim
from oracle import nerfh_oracle as orc
dev = torch.device("cuda:0")
R, Nc, Ni = 1536, 64, 128
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
opt = (torch.optim.Adam if os.environ.get('DFN_TORCH_ADAM') == '1' else optim.Adam)(tr.params, lr=5e-4, betas=(0.9, 0.999))
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
target = torch.rand(R, 3, device=dev)
marks = []
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        r = f(*a, **k); e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, e)); return r
    setattr(obj, name, g)
for n in ("draw", "forward", "loss", "backward"):
    wrap(tr, n)
for _ in range(5):
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.); opt.step()
torch.cuda.synchronize(); marks.clear()
N = 60
for _ in range(N):
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
    opt.step(); e = torch.cuda.Event(enable_timing=True); e.record(); marks.append(("adam", e))
torch.cuda.synchronize()
acc = {}
for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
    acc.setdefault(n1, []).append(e0.elapsed_time(e1))
tot = marks[4][1].elapsed_time(marks[-1][1]) / (N - 1)
print(os.environ.get("DFN_TRAIN_BWD_PAIR", "default"), f"step {tot:.3f} ms |", " ".join(f"{k} {np.median(v):.3f}" for k, v in acc.items()))

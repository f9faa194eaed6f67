import json, os, sys, time
import numpy as np
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc
dev = torch.device("cuda:0")
R, Nc, Ni, iters = 1536, 64, 128, 60
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
opt = torch.optim.Adam(tr.params, lr=5e-4, betas=(0.9, 0.999))
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
target = torch.rand(R, 3, device=dev)
def timed(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / iters * 1e3
def step():
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
    opt.step()
def step_noadam():
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
draws = tr.draw(R, Nc, Ni, 1., dev)
res = {}
res["fwd"] = timed(lambda: tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2]))
out = tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2])
loss5, gs, gts = tr.loss(out, target)
res["bwd"] = timed(lambda: tr.backward(*gs, gts))
res["step_skip"] = timed(step)
res["step_skip_noadam"] = timed(step_noadam)
tr.range_check = None
res["step_nocheck"] = timed(step)
res["step_nocheck_noadam"] = timed(step_noadam)
tr.range_check = "skip"
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); res["host_enqueue"] = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
print(os.environ.get("DFN_TRAIN_BWD_PAIR", "default"), " ".join(f"{k} {v:.3f}" for k, v in res.items()))

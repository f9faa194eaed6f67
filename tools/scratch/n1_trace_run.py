import os, sys, time
import numpy as np
import torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc
dev = torch.device("cuda:0")
R, Nc, Ni = 1536, 64, 128
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
opt = torch.optim.Adam(tr.params, lr=5e-4, betas=(0.9, 0.999))
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
target = torch.rand(R, 3, device=dev)
for _ in range(12):
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
    opt.step()
torch.cuda.synchronize()

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
for _ in range(5):
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.); opt.step()
acc = {}
def host(name, fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.) + (time.perf_counter() - t0); return r
N = 30
for _ in range(N):
    draws = host("draw", lambda: tr.draw(R, Nc, Ni, 1., dev))
    out = host("forward", lambda: tr.forward(o, d, hist, Nc, Ni, 0., 2.5, draws[0], draws[1], 0., draws[2]))
    l = host("loss", lambda: tr.loss(out, target))
    host("backward", lambda: tr.backward(*l[1], l[2]))
    host("adam", lambda: opt.step())
    host("train_step_whole", lambda: tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.))
print(os.environ.get("DFN_TRAIN_BWD_PAIR", "default"), " ".join(f"{k} {v / N * 1e3:.3f}" for k, v in acc.items()), "ms of host time per call, device idle")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.); opt.step(); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

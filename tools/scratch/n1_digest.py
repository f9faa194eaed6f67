import hashlib, json, sys, os
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc
dev = torch.device("cuda:0")
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
for R in (1536, 200, 4096):
    Nc, Ni = 64, 128
    rng = np.random.default_rng(0)
    ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
    sel = rng.choice(480 * 640, R, replace=False)
    o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
    hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
    g = torch.Generator().manual_seed(3)
    target = torch.rand(R, 3, generator=g).to(dev)
    draws = tuple(t.to(dev) for t in (torch.rand(R, Nc, generator=g), torch.randn(R, Nc, generator=g), torch.rand(R, Ni, generator=g)))
    for rep in range(2):
        for p in tr.params: p.grad = None
        ld, _, _ = tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0., draws=draws)
        torch.cuda.synchronize()
        h = hashlib.sha256()
        for k, p in zip(tr.names, tr.params):
            if "embedding" not in k: h.update(p.grad.cpu().numpy().tobytes())
        print(os.environ.get("DFN_TRAIN_BWD_PAIR"), R, rep, h.hexdigest()[:16], float(ld["f_l"]))

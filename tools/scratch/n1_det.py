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
R, Nc, Ni = 1536, 64, 128
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
g = torch.Generator().manual_seed(3)
target = torch.rand(R, 3, generator=g).to(dev)
draws = tuple(t.to(dev) for t in (torch.rand(R, Nc, generator=g), torch.randn(R, Nc, generator=g), torch.rand(R, Ni, generator=g)))
prev = None
for rep in range(4):
    for p in tr.params:
        p.grad = None
    ld, _, _ = tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0., draws=draws)
    torch.cuda.synchronize()
    cur = {k: p.grad.cpu().numpy().copy() for k, p in zip(tr.names, tr.params)}
    cur.update({"loss_" + k: np.float32(float(v)) for k, v in ld.items()})
    if prev is not None:
        diff = [(k, float(np.abs(cur[k] - prev[k]).max()), float(np.abs(prev[k]).max())) for k in cur if not np.array_equal(cur[k], prev[k])]
        print("rep", rep, "differs from previous in", len(diff), "tensors:", diff[:12])
    prev = cur
print("flagged steps", getattr(tr, "flagged_steps", None))

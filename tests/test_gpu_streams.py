"""The side-stream schedules of the DFNet training step change WHEN kernels run, never what they compute.

`forward_train_keep` starts a pyramid level's adaptation branch on the handle's side stream beside the encoder, `backward_params_split`
runs the levels' and the encoder's weight gradients there beside the data-gradient chain (dfnet_api.hip; LABBOOK R6.10, R6.11).  Every
kernel sums in a fixed order, so the step must be BIT-identical to the one-stream schedule (the A/B switches DFN_ADAPT_FWD_SIDE=0,
DFN_ADAPT_WGRAD_SIDE=0), run after run — a missing event (a buffer overwritten while the other stream still reads it) shows up here as
a differing digest.  The switches are read once per process: each schedule runs in its own interpreter."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_STEP = r"""
import hashlib, json, sys
import torch
sys.path.insert(0, %(root)r)
from dfnet_amd import feature_misc as fm, synthetic as syn
from dfnet_amd.dfnet import DFNet
dev = "cuda:0"
m = DFNet()
m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.dfnet_weights(3).items()}, strict=False)
m.to(dev).train()
m.pyramid_features = True
g = torch.Generator().manual_seed(5)
B, H, W = 4, 120, 160
x = torch.rand(3 * B, 3, H, W, generator=g).to(dev)            # [target, render, synthesised views] in one encoder pass
Gp = torch.randn(3 * B, 12, generator=g).to(dev)
digests = []
for rep in range(3):
    feats, pose = m(x, True, upsampleH=H, upsampleW=W, feature_images=2 * B)
    loss = fm.triplet_loss_hard_negative_mining_plus(feats[1], feats[0], margin=1.0) + (pose * Gp).sum()
    loss.backward()
    torch.cuda.synchronize()
    h = hashlib.sha256()
    h.update(loss.detach().cpu().numpy().tobytes()); h.update(pose.detach().cpu().numpy().tobytes())
    n = 0
    for name, q in sorted(m.named_parameters()):
        if q.grad is not None:
            h.update(q.grad.cpu().numpy().tobytes()); n += 1
        q.grad = None
    for name, b in sorted(m.named_buffers()):
        if "running" in name:
            h.update(b.cpu().numpy().tobytes())
    digests.append(h.hexdigest())
    m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.dfnet_weights(3).items()}, strict=False)   # (running statistics back)
print(json.dumps({"digests": digests, "n_grads": n, "loss": float(loss)}))
"""


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", _STEP % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_side_stream_schedules_are_bit_identical_to_one_stream():
    beside = _run({})
    in_line = _run({"DFN_ADAPT_FWD_SIDE": "0", "DFN_ADAPT_WGRAD_SIDE": "0"})
    assert beside["n_grads"] == in_line["n_grads"] == 46
    assert len(set(beside["digests"])) == 1, beside          # run after run
    assert len(set(in_line["digests"])) == 1, in_line
    assert beside["digests"][0] == in_line["digests"][0], (beside["loss"], in_line["loss"])


# ---------------------------------------------------------------------- N1: both networks' data-gradient chains in one grid
_N1_STEP = r"""
import hashlib, json, sys
import numpy as np
import torch
sys.path.insert(0, %(root)r)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc
dev = torch.device("cuda:0")
E, mods, _ = modules(W=128)
tr = nerf_train.NerfHTrainer(E, *mods)
R, Nc, Ni = %(rays)d, 64, 128
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
g = torch.Generator().manual_seed(3)
target = torch.rand(R, 3, generator=g).to(dev)
draws = tuple(t.to(dev) for t in (torch.rand(R, Nc, generator=g), torch.randn(R, Nc, generator=g), torch.rand(R, Ni, generator=g)))
digests, emb = [], []
for rep in range(3):
    for p in tr.params:
        p.grad = None
    ld, _, _ = tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0., draws=draws)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for k in sorted(ld):
        h.update(np.float32(float(ld[k])).tobytes())
    n = 0
    for name, p in zip(tr.names, tr.params):
        if "embedding" in name:      # scattered with atomic adds: the last bit depends on the arrival order, in every schedule
            emb.append(p.grad.double().abs().sum().item())
        else:
            h.update(p.grad.cpu().numpy().tobytes()); n += 1
    digests.append(h.hexdigest())
print(json.dumps({"digests": digests, "n_grads": n, "emb": emb, "loss": {k: float(v) for k, v in ld.items()}, "fused": not tr.exact}))
"""


@pytest.mark.parametrize("rays", [1536, 200])
def test_n1_backward_chain_pair_is_bit_identical_to_two_launches(rays):
    """nerfh_fused_api.hip: train_backward runs the fine and the coarse network's data-gradient chains as the two halves of one grid
    (train_bwd_chain_pair_kernel: the coarse chain's workgroups fill the fine chain's half-tile tail); DFN_TRAIN_BWD_PAIR=0 launches
    them one after the other.  The loss and the 62 weight / bias gradients of the step bit for bit, run after run and across the two
    schedules; the two embedding gradients (atomic scatter) to round-off."""
    def run(env_extra):
        env = dict(os.environ, **env_extra)
        out = subprocess.run([sys.executable, "-c", _N1_STEP % {"root": ROOT, "rays": rays}], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])
    paired, apart = run({}), run({"DFN_TRAIN_BWD_PAIR": "0"})
    assert paired["fused"] and apart["fused"] and paired["n_grads"] == apart["n_grads"] == 62
    assert len(set(paired["digests"])) == 1, paired
    assert len(set(apart["digests"])) == 1, apart
    assert paired["digests"][0] == apart["digests"][0], (paired["loss"], apart["loss"])
    for a, b in zip(paired["emb"], apart["emb"]):
        assert abs(a - b) <= 1e-5 * abs(b)

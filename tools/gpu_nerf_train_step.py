#!/usr/bin/env python3
"""NeRF-H optimisation step timing at the reference's defaults (N_rand 1536 rays, 64 + 128 samples, netwidth 128; run_nerf.py:32-80):
forward (training-mode render), fused NerfWLoss, backward (every gradient), Adam.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules  # seeded modules on the GPU
from dfnet_amd import nerf_train, synthetic as syn
from dfnet_amd import optim
from oracle import nerfh_oracle as orc

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
W = int(sys.argv[2]) if len(sys.argv) > 2 else 128
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
Nc, Ni = 64, 128
E, mods, _ = modules(W=W)
tr = nerf_train.NerfHTrainer(E, *mods)
tr.exact = os.environ.get("DFN_TRAIN_EXACT", "0") == "1"   # DFN_TRAIN_EXACT=1: time the layer-by-layer exact-fp32 step instead
opt = (torch.optim.Adam if os.environ.get('DFN_TORCH_ADAM') == '1' else optim.Adam)(tr.params, lr=5e-4, betas=(0.9, 0.999))
rng = np.random.default_rng(0)
ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
sel = rng.choice(480 * 640, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
hist = torch.from_numpy(syn.HIST_IDX)[None].to(dev)
target = torch.rand(R, 3, device=dev)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def step():
    tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0.)
    opt.step()


draws = tr.draw(R, Nc, Ni, 1., dev)
fwd_ms = timed(lambda: tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2]))
out = tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2])
loss5, gs, gts = tr.loss(out, target)
bwd_ms = timed(lambda: tr.backward(*gs, gts))
step_ms = timed(step)
# host time to ENQUEUE one step on an idle device (nothing to wait for): how far the host is from being the bound
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); host_ms = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
macs = R * (Nc * (130944 + 16384 + 64 * (W + 27) + 192) + (Nc + Ni) * 182720) * (W / 128) ** 2   # ~ algorithmic MAC per step forward
print(json.dumps({"workload": f"NeRF-H optimisation step: {R} rays, {Nc}+{Ni} samples, netwidth {W}, perturb 1", "forward_ms": fwd_ms,
                  "backward_ms": bwd_ms, "step_ms_with_adam": step_ms, "host_enqueue_ms_of_one_step_on_an_idle_device": host_ms, "rays_per_s": R / step_ms * 1e3,
                  "approx_forward_TFLOPs": 2 * macs / fwd_ms / 1e9, "approx_step_TFLOPs": 6 * macs / step_ms / 1e9,
                  "arithmetic": ("exact fp32 MFMA (v_mfma_f32_32x32x2_f32), layer by layer" if tr.exact or W != 128 else
                                 "split-f16 MFMA, fused register-resident chains + weight-gradient stream")}))

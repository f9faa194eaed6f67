#!/usr/bin/env python3
"""Fused NeRF-H training step (csrc/nerfh_fused_*.hip) against the layer-by-layer exact-fp32 step on the same rays, weights and draws:
forward outputs and every gradient tensor (relative L2), then timing of both.  usage: gpu_fused_debug.py [rays] [Nc] [Ni] [trained]
("trained": the trained-like weights of tests/golden/trained_nerfh_weights.npz on the scene they were trained on, targets = its
analytic images: small, cancelling gradients — the regime a converged run spends its time in — instead of seeded random weights)"""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc

dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
Nc = int(sys.argv[2]) if len(sys.argv) > 2 else 64
Ni = int(sys.argv[3]) if len(sys.argv) > 3 else 128
TRAINED = len(sys.argv) > 4 and sys.argv[4] == "trained"
E, mods, _ = modules()
rng = np.random.default_rng(0)
if TRAINED:
    cw, fw, ea, et = syn.trained_nerfh_weights()
    mods[0].load_state_dict({k: torch.from_numpy(v) for k, v in cw.items()})
    mods[1].load_state_dict({k: torch.from_numpy(v) for k, v in fw.items()})
    mods[2].weight.data.copy_(torch.from_numpy(ea))
    mods[3].weight.data.copy_(torch.from_numpy(et))
    E.load_numpy(cw, fw, ea, et)
    H, W, focal = 60, 80, 585.0 / 8
    pose = syn.orbit_pose(7, 16)[:3, :4]
    ro, rd = orc.get_rays(H, W, focal, torch.from_numpy(pose))
    sel = rng.choice(H * W, R, replace=False)
    target = torch.from_numpy(syn.analytic_scene_image(pose, H, W, focal, 2.5)).reshape(-1, 3)[sel].contiguous().to(dev)
    hist = torch.from_numpy(syn.HIST_IDX)[None].repeat(R, 1).contiguous().to(dev)
else:
    ro, rd = orc.get_rays(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8))[:3, :4])
    sel = rng.choice(480 * 640, R, replace=False)
    hist = torch.from_numpy(rng.integers(0, 40, (R, 10)).astype(np.float32)).to(dev)
    target = torch.rand(R, 3, device=dev)
tr = nerf_train.NerfHTrainer(E, *mods)
tr.range_check = "repeat"
o, d = ro.reshape(-1, 3)[sel].contiguous().to(dev), rd.reshape(-1, 3)[sel].contiguous().to(dev)
draws = tr.draw(R, Nc, Ni, 1., dev, torch.Generator(device=dev).manual_seed(1))


def run(mode):
    tr.exact = bool(mode)   # 1: layer-by-layer exact fp32, 0: fused chains
    for p in tr.params:
        p.grad = None
    ld, psnr, out = tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=1., draws=draws)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in ld.items()}, {k: v.clone() for k, v in out.items()}, [p.grad.clone() for p in tr.params]


def rel(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-300))


ld1, out1, g1 = run(1)
ld0, out0, g0 = run(0)
print("range flags after the fused step:", E.range_flags() if hasattr(E, "range_flags") else "n/a")
print("loss exact", ld1)
print("loss fused", ld0)
for k in out1:
    print(f"out {k:18s} rel L2 {rel(out0[k], out1[k]):.3e}  max abs diff {float((out0[k] - out1[k]).abs().max()):.3e}  nan {int(torch.isnan(out0[k]).sum())}")
worst = 0.
for name, a, b in zip(tr.names, g0, g1):
    e = rel(a, b)
    worst = max(worst, e if np.isfinite(e) else 1e9)
    print(f"grad {name:42s} rel L2 {e:.3e}  |exact| {float(b.norm()):.3e}  nan {int(torch.isnan(a).sum())}")
print("worst gradient rel L2:", worst)


def timed(fn, iters=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


res = {}
for mode, tag in ((1, "exact"), (0, "fused")):
    tr.exact = bool(mode)
    fwd = timed(lambda: tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2]))
    out = tr.forward(o, d, hist, Nc, Ni, 0., 2.5, *draws[:2], 0., draws[2])
    loss5, gs, gts = tr.loss(out, target)
    bwd = timed(lambda: tr.backward(*gs, gts))
    step = timed(lambda: tr.train_step(o, d, hist, target, Nc, Ni, 0., 2.5, perturb=1., raw_noise_std=0., draws=draws))
    res[tag] = dict(forward_ms=fwd, backward_ms=bwd, step_ms=step)
print(json.dumps({"rays": R, "Nc": Nc, "Ni": Ni, **res}))

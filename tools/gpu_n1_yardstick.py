#!/usr/bin/env python3
"""N1 step on the trained-like weights against the float64 oracle: per gradient tensor the distance from float64 autograd of (a) torch's own
fp32 autograd (the CPU oracle in fp32: the yardstick), (b) the exact-fp32 HIP step, (c) the fused step with the fine network's stored operands
as one f16 plane (default), (d) as hi | lo planes.  usage: gpu_n1_yardstick.py [rays] [random]   ("random": seeded random weights instead)"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_gpu_train import modules
from tests.yardstick import float64_default, rel_l2, to64
from dfnet_amd import nerf_train, synthetic as syn
from oracle import nerfh_oracle as orc

T = torch.from_numpy
dev = torch.device("cuda:0")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 256
RANDOM = len(sys.argv) > 2 and sys.argv[2] == "random"
Nc, Ni, NEAR, FAR = 64, 128, 0., 2.5
E, mods, (cw, fw, ea, et) = modules()
rng = np.random.default_rng(0)
H, W, focal = 60, 80, 585.0 / 8
pose = syn.orbit_pose(7, 16)[:3, :4]
if not RANDOM:
    cw, fw, ea, et = syn.trained_nerfh_weights()
    mods[0].load_state_dict({k: T(v) for k, v in cw.items()})
    mods[1].load_state_dict({k: T(v) for k, v in fw.items()})
    mods[2].weight.data.copy_(T(ea)); mods[3].weight.data.copy_(T(et))
    E.load_numpy(cw, fw, ea, et)
ro, rd = orc.get_rays(H, W, focal, T(pose))
sel = rng.choice(H * W, R, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
target = T(syn.analytic_scene_image(pose, H, W, focal, FAR)).reshape(-1, 3)[sel].contiguous() if not RANDOM else torch.rand(R, 3)
hist = T(syn.HIST_IDX)[None].repeat(R, 1).contiguous()
gen = torch.Generator().manual_seed(9)
draws = (torch.rand(R, Nc, generator=gen), torch.randn(R, Nc, generator=gen), torch.rand(R, Ni, generator=gen))
rows = torch.cat([o, d, torch.zeros(R, 1), torch.full((R, 1), FAR), d / d.norm(dim=-1, keepdim=True), hist], 1)
c, f = {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}
_, _, g32, _ = orc.train_step(rows, target, c, f, T(ea), T(et), Nc, Ni, *draws, perturb=1., raw_noise_std=1.)
with float64_default():
    _, _, g64, _ = orc.train_step(to64(rows), to64(target), to64(c), to64(f), to64(T(ea)), to64(T(et)), Nc, Ni, *to64(draws), perturb=1., raw_noise_std=1.)
tr = nerf_train.NerfHTrainer(E, *mods)
tr.range_check = "repeat"
got = {}
for tag, exact, split in (("exact", True, False), ("fused", False, False), ("fused_split", False, True)):
    tr.exact, tr.fused_split = exact, split
    for p in tr.params:
        p.grad = None
    tr.train_step(o.to(dev), d.to(dev), hist.to(dev), target.to(dev), Nc, Ni, NEAR, FAR, perturb=1., raw_noise_std=1., draws=tuple(t.to(dev) for t in draws))
    got[tag] = {k: p.grad.detach().cpu().clone() for k, p in zip(tr.names, tr.params)}
print(f"{'tensor':44s} {'torch fp32':>10s} {'HIP exact':>10s} {'fused':>10s} {'fused hi|lo':>11s}   |g64|")
worst = {"yard": 0., "exact": 0., "fused": 0., "fused_split": 0.}
ratio = 0.
for k in tr.names:
    if k not in g64:
        continue
    e = {"yard": rel_l2(g32[k], g64[k]), **{t: rel_l2(got[t][k], g64[k]) for t in ("exact", "fused", "fused_split")}}
    for t in e:
        worst[t] = max(worst[t], e[t])
    ratio = max(ratio, e["fused"] / max(e["yard"], e["exact"], 1e-30))
    if max(e.values()) > 5e-5:
        print(f"{k:44s} {e['yard']:10.2e} {e['exact']:10.2e} {e['fused']:10.2e} {e['fused_split']:11.2e}   {float(g64[k].norm()):.2e}")
print("worst:", {k: f"{v:.2e}" for k, v in worst.items()}, " largest fused / max(torch fp32, HIP exact):", f"{ratio:.1f}")

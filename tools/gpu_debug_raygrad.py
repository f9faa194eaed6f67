#!/usr/bin/env python3
"""Where does the ray gradient of the generic-width render lose accuracy?  Netwidth 32 (tests/test_gpu_train.py:
test_generic_width_render_gradient_vs_oracle): stage A = the compositing backward alone (HIP vs float64 / fp32 autograd of the oracle's
compositing on the SAME raw, z); whole = the full gradient, per-ray error distribution vs float64."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
from oracle import nerfh_oracle as orc
from tests.yardstick import float64_default, to64
T = torch.from_numpy
DEV = torch.device("cuda:0")
width = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cw, fw, ea, et = syn.nerfh_weights(4, W=width)
E = eng.NerfHEngine(width=width).load_numpy(cw, fw, ea, et)
c, f = {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}
rng = np.random.default_rng(11)
ro, rd = orc.get_rays(480, 640, 585.0, T(syn.orbit_pose(3, 8))[:3, :4])
n = 150
sel = rng.choice(480 * 640, n, replace=False)
o, d = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
hist = T(rng.integers(0, 40, (n, 10)).astype(np.float32))
G = T(rng.standard_normal((n, 3)).astype(np.float32))
rl = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm())
# ---- stage A: compositing backward on the HIP forward's own raw / z
rgb, disp, acc, raw = E.generic_render_rays(o.to(DEV), d.to(DEV), hist.to(DEV), 16, 32, 0., 2.5, retraw=True)
rows = orc.pack_ray_rows(o, d, 0., 2.5, hist)
st = {}
out = orc.render_rays(rows, c, f, T(ea), T(et), 16, 32, retraw=True, stages=st)
z = st.get("z_fine") if "z_fine" in st else None
print("stage keys:", list(st.keys()))
zz = None
for k in st:
    if torch.is_tensor(st[k]) and st[k].shape == (n, 48): zz = st[k]; print("z from stage", k)
if zz is not None:
    graw = eng.composite_fine_backward(raw, zz.to(DEV), G.to(DEV))
    for dt, name in ((torch.float32, "fp32 autograd"), (torch.float64, "float64 autograd")):
        r = raw.detach().cpu().to(dt).requires_grad_(True)
        prev = torch.get_default_dtype(); torch.set_default_dtype(dt)
        res = orc.composite_fine(r, zz.to(dt))
        (res["rgb"] * G.to(dt)).sum().backward()
        torch.set_default_dtype(prev)
        if dt == torch.float64: g64 = r.grad
        else: g32 = r.grad
    for ch, nm in ((slice(0, 3), "d rgb_s"), (3, "d sigma_s"), (slice(4, 7), "d rgb_t"), (7, "d sigma_t")):
        print(f"compositing backward {nm}: HIP vs float64 {rl(graw[..., ch], g64[..., ch]):.2e}, torch fp32 vs float64 {rl(g32[..., ch], g64[..., ch]):.2e}")
# ---- whole gradient
_, r32o, r32d = orc.render_grad_rays(o, d, G, c, f, T(ea), T(et), 16, 32, 0., 2.5, hist)
with float64_default():
    _, o64, d64 = orc.render_grad_rays(*to64((o, d, G, c, f, T(ea), T(et))), 16, 32, 0., 2.5, to64(hist))
go, gd, _ = E.render_rays_backward(o.to(DEV), d.to(DEV), hist.to(DEV), 16, 32, 0., 2.5, G.to(DEV), precision="generic")
per_h = ((go.cpu().double() - o64).norm(dim=1) / o64.norm(dim=1))
per_t = ((r32o.double() - o64).norm(dim=1) / o64.norm(dim=1))
print("whole: HIP", rl(go, o64), "torch32", rl(r32o, o64))
print("per-ray HIP quantiles", [float(per_h.quantile(q)) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
print("per-ray t32 quantiles", [float(per_t.quantile(q)) for q in (0.1, 0.5, 0.9, 0.99, 1.0)])
abs_h = (go.cpu().double() - o64).norm(dim=1); nrm = o64.norm(dim=1)
idx = abs_h.argsort(descending=True)[:5]
print("worst rays (abs err, |g|, |g| rank):", [(int(i), float(abs_h[i]), float(nrm[i])) for i in idx], "max |g|", float(nrm.max()))
# forward agreement
print("forward rgb HIP vs oracle", rl(rgb, out["rgb_map"]), " raw", rl(raw, out["raw"]))
# ---- is the worst ray sitting on a kink (ReLU gate within round-off of zero)?  float64 gradient under 1e-6 perturbations of its origin
i = int(idx[0])
with float64_default():
    base = o64[i]
    for trial in range(6):
        pert = torch.randn(3, dtype=torch.float64, generator=torch.Generator().manual_seed(trial)) * 2e-6
        op = to64(o)[i:i + 1] + pert
        _, gp, _ = orc.render_grad_rays(op, to64(d)[i:i + 1], to64(G)[i:i + 1], *to64((c, f, T(ea), T(et))), 16, 32, 0., 2.5, to64(hist)[i:i + 1])
        print(f"ray {i}: float64 gradient moves by {float((gp[0] - base).norm() / base.norm()):.2e} under a {float(pert.norm()):.1e} shift of its origin;"
              f" HIP error on this ray {float((go[i].cpu().double() - base).norm() / base.norm()):.2e}")

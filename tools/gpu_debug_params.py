#!/usr/bin/env python3
"""Per-tensor error of dfn_dfnet_backward_params at one seed: split-f16 HIP, exact-fp32 HIP, the oracle's fp32 autograd, all against the
oracle's float64 autograd — where along the chain (fc_pose -> conv5_3 -> ... -> conv1_1) an error enters, and how many elements carry it.
usage: gpu_debug_params.py SEED [B 3 H W]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
from oracle import dfnet_oracle as dor
from tests.yardstick import float64_default
T = torch.from_numpy
DEV = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shape = tuple(int(v) for v in sys.argv[2:6]) if len(sys.argv) > 5 else (2, 3, 64, 96)
w = syn.dfnet_weights(3)
E, p = eng.DfnetEngine(3, 12).load_numpy(w), {k: T(v) for k, v in w.items()}
rng = np.random.default_rng(seed)
x = T(rng.uniform(0, 1, shape).astype(np.float32))
G = T(rng.standard_normal((shape[0], 12)).astype(np.float32))
tr = lambda k: k.startswith("encoder.") or k.startswith("fc_pose.")
pp = {k: v.clone().requires_grad_(tr(k)) for k, v in p.items()}
_, pose = dor.dfnet_forward(pp, x, False, True, True)
(pose * G).sum().backward()
with float64_default():
    p64 = {k: v.double().requires_grad_(tr(k)) for k, v in p.items()}
    _, pose64 = dor.dfnet_forward(p64, x.double(), False, True, True)
    (pose64 * G.double()).sum().backward()
x3 = E.backward_params(x.to(DEV), G.to(DEV), precision="f16x3")
h32 = E.backward_params(x.to(DEV), G.to(DEV), precision="f32")
rl = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30))
print(f"seed {seed} {shape}: relative L2 against the float64 autograd — split-f16 HIP | exact-fp32 HIP | oracle fp32 autograd ; elements of the split-f16 tensor off by > 1e-3 of its max")
for k in reversed(list(x3.keys())):
    t = p64[k].grad
    d = (x3[k].double().cpu() - t).abs() / t.abs().max()
    print(f"{k:22s} {rl(x3[k], t):.2e} | {rl(h32[k], t):.2e} | {rl(pp[k].grad, t):.2e} ; {int((d > 1e-3).sum())} of {d.numel()}")
print("pose fp32 vs fp64", rl(pose.detach(), pose64.detach()))
k = os.environ.get("DBG_TENSOR", "encoder.21.weight")
t = p64[k].grad
d = ((x3[k].double().cpu() - t).abs() / t.abs().max())
bad = (d > 1e-3).nonzero()
print(k, "bad elements", bad.shape[0], "distinct co", bad[:, 0].unique().numel(), "distinct ci", bad[:, 1].unique().tolist()[:20], "taps", (bad[:, 2] * 3 + bad[:, 3]).unique().tolist())
ci = int(bad[:, 1].mode().values)
print("most frequent ci", ci, "count", int((bad[:, 1] == ci).sum()), "max rel err", float(d.max()), "rel L2 of that ci slice", rl(x3[k][:, ci], t[:, ci]))
# ratio got / truth on the bad slice: a common factor (scale error) or scattered?
r = (x3[k].double().cpu()[:, ci] / t[:, ci]).reshape(-1)
print("got / truth on that slice: median %.6f, 5%% %.6f, 95%% %.6f" % (float(r.median()), float(r.quantile(0.05)), float(r.quantile(0.95))))

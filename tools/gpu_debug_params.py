#!/usr/bin/env python3
"""Per-tensor error of dfn_dfnet_backward_params (split-f16 vs exact fp32 vs oracle autograd) at a small shape."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
from oracle import dfnet_oracle as dor
T = torch.from_numpy
DEV = torch.device("cuda:0")
shape = tuple(int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 3, 64, 96)
w = syn.dfnet_weights(3)
E, p = eng.DfnetEngine(3, 12).load_numpy(w), {k: T(v) for k, v in w.items()}
for seed in range(31, 41):
    rng = np.random.default_rng(seed)
    x = T(rng.uniform(0, 1, shape).astype(np.float32))
    G = T(rng.standard_normal((shape[0], 12)).astype(np.float32))
    pp = {k: v.clone().requires_grad_(k.startswith("encoder.") or k.startswith("fc_pose.")) for k, v in p.items()}
    _, pose = dor.dfnet_forward(pp, x, False, True, True)
    (pose * G).sum().backward()
    got = E.backward_params(x.to(DEV), G.to(DEV), precision="f16x3")
    ref32 = E.backward_params(x.to(DEV), G.to(DEV), precision="f32")
    worst = {"x3": (0, ""), "f32": (0, ""), "x3_vs_f32": (0, "")}
    for k, g in got.items():
        r = pp[k].grad.double()
        e = float((g.double().cpu() - r).abs().max() / r.abs().max())
        e32 = float((ref32[k].double().cpu() - r).abs().max() / r.abs().max())
        ex = float((g.double() - ref32[k].double()).abs().max() / ref32[k].double().abs().max())
        if e > worst["x3"][0]: worst["x3"] = (e, k)
        if e32 > worst["f32"][0]: worst["f32"] = (e32, k)
        if ex > worst["x3_vs_f32"][0]: worst["x3_vs_f32"] = (ex, k)
    print(seed, {k: f"{v[0]:.2e} {v[1]}" for k, v in worst.items()})

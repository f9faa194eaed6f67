#!/usr/bin/env python3
"""GPU diagnostic: per-stage error of the HIP path vs the CPU oracle + quick timings.
Writes gpurun_out/diag.json.  (Development aid; the parity tests proper are tests/test_gpu_*.py.)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn  # noqa: E402
from oracle import nerfh_oracle as orc  # noqa: E402

T = torch.from_numpy
out = {}


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    d = (a - b).abs()
    return dict(max_abs=float(d.max()), max_rel_to_max=float(d.max() / b.abs().max().clamp_min(1e-30)),
                nan=int(torch.isnan(a).sum()))


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine().load_numpy(cw, fw, ea, et)
    tt = lambda d: {k: T(v) for k, v in d.items()}
    c, f, tea, tet = tt(cw), tt(fw), T(ea), T(et)

    # posenc
    x = (torch.rand(20000, 3) * 6 - 3)
    ref64 = torch.cat([x.double()] + [fn(x.double() * 2.0 ** k) for k in range(10) for fn in (torch.sin, torch.cos)], -1)
    out["posenc_precise_vs_f64"] = err(eng.posenc(x.to(dev), 10, fast=False), ref64)
    out["posenc_fast_vs_f64"] = err(eng.posenc(x.to(dev), 10, fast=True), ref64)
    out["posenc_torchcpu_vs_f64"] = err(orc.posenc(x, 10), ref64)

    # rays
    H, W, focal = 48, 64, 58.5
    c2w = T(syn.orbit_pose(3, 8))
    o, d, v = eng.raygen(H, W, focal, c2w.to(dev))
    ro, rd = orc.get_rays(H, W, focal, c2w[:3, :4])
    out["raygen_d"] = err(d, rd)
    out["raygen_o"] = err(o, ro)
    rows = orc.pack_ray_rows(ro, rd, 0., 2.5, syn.HIST_IDX)
    out["viewdirs"] = err(v.reshape(-1, 3), rows[:, 8:11])

    R = 700
    sel = torch.randperm(H * W)[:R]
    rows = rows[sel]
    Nc, Ni = 64, 128
    st = {}
    with torch.no_grad():
        ref = orc.render_rays(rows, c, f, tea, tet, Nc, Ni, retraw=True, stages=st)
    o_, d_, v_ = rows[:, 0:3].to(dev).contiguous(), rows[:, 3:6].to(dev).contiguous(), rows[:, 8:11].to(dev).contiguous()
    hist = T(syn.HIST_IDX).to(dev)
    for prec in ("f32", "f16"):
        sig = E.mlp_coarse(o_, d_, Nc, 0., 2.5, precision=prec)
        out[f"coarse_sigma_{prec}"] = err(sig, st["sigma_coarse"])
        raw = E.mlp_fine(o_, d_, v_, hist, st["z_fine"].to(dev), precision=prec)
        out[f"fine_raw_{prec}"] = err(raw, st["raw"])
        for ch, nm in ((slice(0, 3), "rgb_s"), (3, "sigma_s"), (slice(4, 7), "rgb_t"), (7, "sigma_t"), (8, "beta")):
            out[f"fine_raw_{prec}_{nm}"] = err(raw[..., ch], st["raw"][..., ch])
        rgb, disp, acc, raw2 = E.render_rays(o_, d_, hist, Nc, Ni, 0., 2.5, retraw=True, precision=prec)
        out[f"render_rgb_{prec}"] = err(rgb, ref["rgb_map"])
        out[f"render_disp_{prec}"] = err(disp, ref["disp_map"])
        out[f"render_acc_{prec}"] = err(acc, ref["acc_map"])
        out[f"render_raw_{prec}"] = err(raw2, ref["raw"])
    z, w, zs = eng.sample_fine(st["sigma_coarse"].to(dev), Ni, 0., 2.5, want_aux=True)
    out["sample_fine_z"] = err(z, st["z_fine"])
    out["sample_fine_w"] = err(w, st["weights_coarse"])
    out["sample_fine_zs"] = err(zs, st["z_samples"])
    comp = eng.composite_fine(st["raw"].to(dev), st["z_fine"].to(dev), want_aux=True)
    cref = orc.composite_fine(st["raw"], st["z_fine"])
    for k in ("rgb", "disp", "acc", "depth", "weights", "beta"):
        out["composite_" + k] = err(comp[k], cref[k])

    # timing: full 640x480 frame
    Hh, Ww = 480, 640
    c2w = T(syn.orbit_pose(0, 8)).to(dev)
    for prec in ("f16", "f32"):
        E.render_image(c2w, Hh, Ww, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
        torch.cuda.synchronize()
        t0 = time.time()
        n = 3 if prec == "f16" else 1
        for _ in range(n):
            E.render_image(c2w, Hh, Ww, 585.0, hist, 64, 128, 0., 2.5, precision=prec)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        out[f"frame_ms_{prec}"] = dt * 1e3
        out[f"rays_per_s_{prec}"] = Hh * Ww / dt
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "diag.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out.items():
        print(k, v)


if __name__ == "__main__":
    main()

"""GPU parity tests of the render GRADIENT path (d L/d rgb -> d L/d rays / pose) — what loss.backward()
runs through render(c2w = pose) in the DFNet_dm step (feature/direct_feature_matching.py:340-376).

Checkers: the golden gradients captured from the reference's own autograd (tests/golden/g9_*), and
torch autograd through the CPU oracle.  Tolerances, relative to the largest gradient entry:
  * exact-fp32 MFMA path, and "f16x3" (forward recompute in split-f16 — fp32-grade activations and ReLU gates —
    with the gradient chain in fp32; the default of the gradient path): 1e-5 for the network alone, 2e-4 per ray
    through the whole render, 2e-3 for d c2w — a signed sum over all rays of per-ray terms that largely
    cancel, so fp32 round-off of the terms is amplified (the oracle and the reference themselves differ
    by ~1e-4 there, tests/test_oracle_golden.py);
  * plain-f16 gradient arithmetic is not offered (round 2): a hidden unit whose pre-activation is within f16 rounding of zero flips
    its ReLU gate, which put that mode at 3e-2 of autograd; the library refuses DFN_PREC_F16 on every gradient entry point
    (checked at the end of test_quarter_res_pose_gradient_vs_oracle)."""
import os

import numpy as np
import pytest
import torch

from dfnet_amd import engine as eng
from dfnet_amd import synthetic as syn
from oracle import nerfh_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"f32": 2e-4, "f16x3": 2e-4}
TOL_NET = {"f32": 1e-5, "f16x3": 1e-5}
TOL_L2 = {"f32": 1e-5, "f16x3": 1e-5}
TOL_C2W = {"f32": 2e-3, "f16x3": 2e-3}   # d c2w: a signed sum over all rays that cancels (G9: oracle vs reference 1e-4)


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


def relmax(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert not torch.isnan(a).any()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def tt(d):
    return {k: T(v) for k, v in d.items()}


def dev(x):
    return torch.as_tensor(x).float().to(DEV).contiguous()


@pytest.fixture(scope="module")
def scene():
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine().load_numpy(cw, fw, ea, et)
    return E, tt(cw), tt(fw), T(ea), T(et)


def test_composite_backward_vs_autograd(gold):
    g = gold("g4_composite")
    rng = np.random.default_rng(5)
    for raw_np, z_np in ((g["raw"], g["z"]),
                         (np.abs(rng.standard_normal((37, 192, 9))).astype(np.float32) * 2,
                          np.sort(rng.uniform(0, 2.5, (37, 192)).astype(np.float32), -1))):
        raw = T(raw_np.copy()).requires_grad_(True)
        G = T(rng.standard_normal((raw.shape[0], 3)).astype(np.float32))
        out = orc.composite_fine(raw, T(z_np))
        (out["rgb"] * G).sum().backward()
        got = eng.composite_fine_backward(dev(raw_np), dev(z_np), dev(G))
        assert relmax(got, raw.grad) < 2e-4  # S_i = total - prefix cancels for the front samples


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("n_rays,Nf", [(5, 24), (9, 192), (3, 70)])
def test_mlp_fine_backward_vs_autograd(scene, prec, n_rays, Nf):
    """d sum(raw * G) / d (points, viewdirs) of the fine network, all nine output channels weighted."""
    E, c, f, ea, et = scene
    rng = np.random.default_rng(11)
    o = T(rng.uniform(-.3, .3, (n_rays, 3)).astype(np.float32))
    d = T(rng.standard_normal((n_rays, 3)).astype(np.float32))
    v = d / d.norm(dim=-1, keepdim=True)
    z = T(np.sort(rng.uniform(0, 2.5, (n_rays, Nf)).astype(np.float32), -1))
    G = T(rng.standard_normal((n_rays, Nf, 9)).astype(np.float32))
    pts = (o[:, None] + d[:, None] * z[..., None]).requires_grad_(True)
    vv = v.clone().requires_grad_(True)
    raw = orc.query_fine(f, ea, et, pts, vv, T(syn.HIST_IDX)[None].repeat(n_rays, 1))
    (raw * G).sum().backward()
    got = E.mlp_fine_backward(dev(o), dev(d), dev(v), dev(syn.HIST_IDX), dev(z), dev(G), precision=prec)
    e_pts, e_v = relmax(got[..., :3], pts.grad), relmax(got[..., 3:].sum(1), vv.grad)
    l2 = rel_l2(got[..., :3], pts.grad)
    print(f"{prec} n={n_rays} Nf={Nf}: d pts {e_pts:.2e} (L2 {l2:.2e})  d viewdirs {e_v:.2e}")
    assert e_pts < TOL_NET[prec] and e_v < TOL_NET[prec] and l2 < TOL_L2[prec]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_render_rays_backward_golden(scene, gold, prec):
    E = scene[0]
    for tag in "ab":
        g = gold("g9_render_grad_rays_" + tag)
        go, gd, gv = E.render_rays_backward(dev(g["rays_o"]), dev(g["rays_d"]), dev(g["hist"]), int(g["Nc"]), int(g["Ni"]),
                                            float(g["near"]), float(g["far"]), dev(g["G"]), precision=prec)
        assert gv is None
        scale = max(np.abs(g["grad_rays_o"]).max(), np.abs(g["grad_rays_d"]).max())
        eo = float((go.cpu() - T(g["grad_rays_o"])).abs().max() / scale)
        ed = float((gd.cpu() - T(g["grad_rays_d"])).abs().max() / scale)
        print(f"{prec} {tag}: d rays_o {eo:.2e}  d rays_d {ed:.2e}")
        assert eo < TOL[prec] and ed < TOL[prec]


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
def test_render_image_backward_golden(scene, gold, prec):
    E = scene[0]
    g = gold("g9_render_grad_c2w")
    gc = E.render_image_backward(dev(g["c2w"]), int(g["H"]), int(g["W"]), float(g["focal"]), dev(g["hist"]), int(g["Nc"]),
                                 int(g["Ni"]), float(g["near"]), float(g["far"]), dev(g["G"]), precision=prec)
    e = relmax(gc, g["grad_c2w"])
    print(f"{prec}: d c2w {e:.2e}")
    assert e < TOL_C2W[prec]


def test_explicit_viewdirs_gradient(scene):
    """viewdirs passed explicitly are an independent input: their gradient comes back separately and
    grad_rays_d carries only the point path; folding it through d/|d| reproduces the derived-viewdirs result."""
    E = scene[0]
    o, d, v = eng.raygen(6, 8, 9.0, T(syn.orbit_pose(2, 8)).to(DEV))
    o, d, v = o.reshape(-1, 3), d.reshape(-1, 3), v.reshape(-1, 3)
    G = dev(np.random.default_rng(3).standard_normal((48, 3)))
    hist = dev(syn.HIST_IDX)
    go, gd, _ = E.render_rays_backward(o, d, hist, 64, 128, 0., 2.5, G, precision="f32")
    go2, gd2, gv2 = E.render_rays_backward(o, d, hist, 64, 128, 0., 2.5, G, viewdirs=v, precision="f32")
    n = d.norm(dim=-1, keepdim=True)
    fold = gd2 + (gv2 - v * (v * gv2).sum(-1, keepdim=True)) / n
    assert torch.equal(go, go2) and relmax(fold, gd.cpu()) < 1e-5


def test_quarter_res_pose_gradient_vs_oracle(scene):
    """The DFNet_dm geometry: 60x80 render (240x320 / 4) at 64+128, d L/d c2w against autograd through the oracle."""
    E, c, f, ea, et = scene
    H, W, focal = 60, 80, 585.0 / 8
    c2w = T(syn.orbit_pose(6, 8))[:3, :4]
    G = T(np.random.default_rng(9).standard_normal((H, W, 3)).astype(np.float32))
    _, ref = orc.render_grad_c2w(H, W, focal, c2w, G, c, f, ea, et, 64, 128, 0., 2.5, syn.HIST_IDX)
    for prec in ("f32", "f16x3"):
        gc = E.render_image_backward(c2w.to(DEV), H, W, focal, dev(syn.HIST_IDX), 64, 128, 0., 2.5, G.to(DEV), precision=prec)
        e = relmax(gc, ref)
        print(f"{prec}: 60x80 d c2w {e:.2e}")
        assert e < TOL_C2W[prec]
    gc2 = E.render_image_backward(c2w.to(DEV), H, W, focal, dev(syn.HIST_IDX), 64, 128, 0., 2.5, G.to(DEV), precision="f16x3")
    assert torch.equal(gc, gc2)  # deterministic
    with pytest.raises(Exception, match="F16X3 or DFN_PREC_F32"):   # the plain-f16 gradient mode is gone (it was 3e-2 off autograd)
        E.render_image_backward(c2w.to(DEV), H, W, focal, dev(syn.HIST_IDX), 64, 128, 0., 2.5, G.to(DEV), precision="f16")


def test_render_autograd_drop_in(scene, gold):
    """dfnet_amd.rendering.render under autograd: loss.backward() reaches c2w / rays through the HIP gradient path,
    with the reference's call shapes (direct_feature_matching.py:342-349, run_nerf.py:47-51)."""
    from dfnet_amd import nerfw, rendering
    E = scene[0]
    kw = dict(network_query_fn=nerfw.HipQuery(E, 65536), perturb=False, N_importance=128, N_samples=64, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False)
    g = gold("g9_render_grad_c2w")
    pose = dev(g["c2w"]).requires_grad_(True)
    # default: the tracked forward runs in the engine's precision (f16) and the gradient is that forward's
    rgb = rendering.render(int(g["H"]), int(g["W"]), float(g["focal"]), c2w=pose, near=0., far=2.5, img_idx=dev(g["hist"]), **kw)[0]
    (rgb * dev(g["G"])).sum().backward()
    # (f16 coarse net + split-f16 fine net and gradient; random per-pixel weights on a 12x16 image make d c2w a badly conditioned sum)
    assert relmax(pose.grad, g["grad_c2w"]) < 3e-2
    pose.grad = None
    rendering.GRAD_FORWARD_PRECISION = "f32"   # everything tracked in fp32: reference-grade pose gradient
    rgb, disp, acc, extras = rendering.render(int(g["H"]), int(g["W"]), float(g["focal"]), c2w=pose, near=0., far=2.5,
                                              img_idx=dev(g["hist"]), **kw)
    assert rgb.requires_grad and not disp.requires_grad and extras == {}
    assert relmax(rgb, g["rgb"]) < 1e-3
    (rgb * dev(g["G"])).sum().backward()
    assert relmax(pose.grad, g["grad_c2w"]) < TOL_C2W["f32"]
    g = gold("g9_render_grad_rays_b")
    rays = torch.stack([dev(g["rays_o"]), dev(g["rays_d"])]).requires_grad_(True)
    rgb = rendering.render(480, 640, 585., rays=rays, near=0., far=2.5, img_idx=dev(g["hist"])[None], **kw)[0]
    (rgb * dev(g["G"])).sum().backward()
    scale = np.abs(g["grad_rays_d"]).max()
    assert float((rays.grad[0].cpu() - T(g["grad_rays_o"])).abs().max()) < TOL["f32"] * scale
    assert float((rays.grad[1].cpu() - T(g["grad_rays_d"])).abs().max()) < TOL["f32"] * scale
    rendering.GRAD_FORWARD_PRECISION = None
    with torch.no_grad():  # and nothing is tracked without grad
        assert not rendering.render(12, 16, 14.6, c2w=pose, near=0., far=2.5, img_idx=dev(g["hist"]), **kw)[0].requires_grad


def test_render_retraw_under_autograd(scene):
    """render(retraw=True) with rays / a pose that require grad (rendering.py:353-400: extras['raw'] is part of the autograd graph
    there): a loss on rgb AND on all nine raw channels against autograd through the oracle — per ray through rays=(o, d), and
    through c2w (get_rays' node in front)."""
    from dfnet_amd import nerfw, rendering
    E, c, f, ea, et = scene
    kw = dict(network_query_fn=nerfw.HipQuery(E, 65536), perturb=False, N_importance=32, N_samples=16, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, retraw=True)
    H, W, focal, Nc, Ni = 6, 8, 7.3, 16, 32
    rng = np.random.default_rng(77)
    c2w = T(syn.orbit_pose(2, 8))[:3, :4].contiguous()
    G = T(rng.standard_normal((H, W, 3)).astype(np.float32))
    Gr = T((rng.standard_normal((H, W, Nc + Ni, 9)) / (Nc + Ni)).astype(np.float32))
    hist = T(syn.HIST_IDX)

    def oracle_loss(o, d):
        rows = orc.pack_ray_rows(o, d, 0., 2.5, syn.HIST_IDX)
        out = orc.render_rays(rows, c, f, ea, et, Nc, Ni, retraw=True)
        return out, (out["rgb_map"] * G.reshape(-1, 3)).sum() + (out["raw"] * Gr.reshape(-1, Nc + Ni, 9)).sum()

    rendering.GRAD_FORWARD_PRECISION = "f32"
    try:
        # rays
        ro, rd = orc.get_rays(H, W, focal, c2w)
        o_ref, d_ref = ro.reshape(-1, 3).clone().requires_grad_(True), rd.reshape(-1, 3).clone().requires_grad_(True)
        out_ref, loss_ref = oracle_loss(o_ref, d_ref)
        loss_ref.backward()
        rays = torch.stack([dev(ro), dev(rd)]).requires_grad_(True)
        rgb, disp, acc, extras = rendering.render(H, W, focal, rays=rays, near=0., far=2.5, img_idx=dev(hist)[None], **kw)
        raw = extras["raw"]
        assert raw.shape == (H, W, Nc + Ni, 9) and raw.requires_grad and rgb.requires_grad and not disp.requires_grad
        assert relmax(raw.reshape(-1, Nc + Ni, 9), out_ref["raw"].detach()) < 2e-5
        ((rgb * dev(G)).sum() + (raw * dev(Gr)).sum()).backward()

        def per_ray(got, go_ref, gd_ref):
            """largest error of each ray's six gradient entries, of the largest entry of the batch"""
            e = torch.maximum((got[0].cpu().reshape(-1, 3) - go_ref).abs().max(-1).values, (got[1].cpu().reshape(-1, 3) - gd_ref).abs().max(-1).values)
            return (e / max(float(go_ref.abs().max()), float(gd_ref.abs().max()))).numpy()

        err_both = per_ray(rays.grad, o_ref.grad, d_ref.grad)
        # a loss on raw ALONE (no d L / d rgb reaches the node's first output; no compositor in the gradient)
        rays.grad = None
        o2, d2 = ro.reshape(-1, 3).clone().requires_grad_(True), rd.reshape(-1, 3).clone().requires_grad_(True)
        rows = orc.pack_ray_rows(o2, d2, 0., 2.5, syn.HIST_IDX)
        (orc.render_rays(rows, c, f, ea, et, Nc, Ni, retraw=True)["raw"] * Gr.reshape(-1, Nc + Ni, 9)).sum().backward()
        extras = rendering.render(H, W, focal, rays=rays, near=0., far=2.5, img_idx=dev(hist)[None], **kw)[3]
        (extras["raw"] * dev(Gr)).sum().backward()
        err_raw = per_ray(rays.grad, o2.grad, d2.grad)
        print(f"retraw under autograd, per-ray gradient error of the largest entry (48 rays): raw-only loss median {np.median(err_raw):.1e} "
              f"worst {err_raw.max():.1e}, rgb + raw loss median {np.median(err_both):.1e} worst {err_both.max():.1e}")
        # Measured: median 1.9e-5 of the largest entry — what the oracle's own fp32 autograd sits from a float64 evaluation on the same
        # samples (2e-5) — and a tail up to 2.9e-4 on a few rays (the gradient kernel takes its ReLU gates from the tracked split-f16
        # forward, the oracle from its fp32 one, and a 16 + 32 ray has few samples to dilute a unit that gates differently; the
        # 192-sample rays of test_render_autograd_drop_in stay inside 2e-4).  A wrong or missing raw channel moves EVERY ray by percents.
        for err in (err_raw, err_both):
            assert np.median(err) < 6e-5 and int((err > 2e-4).sum()) <= 3 and err.max() < 3e-3, err
        # pose
        p_ref = c2w.clone().requires_grad_(True)
        ro, rd = orc.get_rays(H, W, focal, p_ref)
        oracle_loss(ro.reshape(-1, 3), rd.reshape(-1, 3))[1].backward()
        pose = dev(c2w).requires_grad_(True)
        rgb, _, _, extras = rendering.render(H, W, focal, c2w=pose, near=0., far=2.5, img_idx=dev(hist), **kw)
        assert extras["raw"].shape == (H, W, Nc + Ni, 9)
        ((rgb * dev(G)).sum() + (extras["raw"] * dev(Gr)).sum()).backward()
        assert relmax(pose.grad, p_ref.grad) < TOL_C2W["f32"]
    finally:
        rendering.GRAD_FORWARD_PRECISION = None


# ---------------------------------------------------------------------- DFNet input gradient / bicubic adjoint
@pytest.fixture(scope="module")
def dfnet():
    from oracle import dfnet_oracle  # noqa: F401
    w = syn.dfnet_weights(3)
    return eng.DfnetEngine(3, 12).load_numpy(w), {k: T(v) for k, v in w.items()}


@pytest.mark.parametrize("levels", [(0,), (0, 1, 2), (2,)])
@pytest.mark.parametrize("shape,up", [((2, 3, 32, 48), (32, 48)), ((1, 3, 72, 104), (60, 90))])
def test_dfnet_backward_input_vs_autograd(dfnet, levels, shape, up):
    """d sum(features * G) / d x against torch autograd through the CPU oracle (frozen weights, eval-mode BN),
    including odd sizes (max-pool remainders, upsample to a different size).  Gradient arithmetic is fp32-grade only
    (exact fp32 or split-f16); the plain-f16 mode is refused by the library."""
    from oracle import dfnet_oracle as dor
    E, p = dfnet
    # The map is piecewise linear: a pre-activation (or a max-pool pair) within round-off of a tie gates differently
    # under a different summation order, and ONE such flip at a coarse level moves the whole gradient by percents —
    # torch's own fp32 and fp64 gradients differ by 7e-2 on some inputs of this very test.  With ~2M gated units a
    # flip somewhere is common, so: over three seeded inputs the fp32 path must match the oracle (evaluated in fp32
    # or fp64) to round-off on at least one, and stay within 5e-2 relative L2 on all of them.
    best = {"f32": 1.0, "f16x3": 1.0}
    for seed in (21, 22, 23):
        rng = np.random.default_rng(seed)
        x0 = rng.uniform(0, 1, shape).astype(np.float32)
        G = T(rng.standard_normal((3, shape[0], 128, *up)).astype(np.float32))
        for t in range(3):
            if t not in levels:
                G[t] = 0
        refs = []
        for dt in (torch.float32, torch.float64):
            x = T(x0).to(dt).requires_grad_(True)
            feats, _ = dor.dfnet_forward({k: v.to(dt) for k, v in p.items()}, x, return_feature=True, isSingleStream=True,
                                         return_pose=False, upsampleH=up[0], upsampleW=up[1])
            (feats[0] * G.to(dt)).sum().backward()
            refs.append(x.grad)
        for prec, tol_l2 in (("f32", 5e-2), ("f16x3", 5e-2)):
            gx = E.backward_input(T(x0).to(DEV), G.to(DEV), levels=levels, precision=prec)
            e = min(relmax(gx, r) for r in refs)
            l2 = min(rel_l2(gx, r) for r in refs)
            print(f"{prec} seed {seed} levels={levels} {shape}: d x {e:.2e}  (L2 {l2:.2e})")
            assert l2 < tol_l2
            best[prec] = min(best[prec], e)
        if best["f32"] < 5e-5 and best["f16x3"] < 5e-5:
            break
    # "f16x3": forward AND gradient convs in split-f16 (gradient tensors scaled by a measured power of two): fp32-grade
    assert best["f32"] < 5e-5 and best["f16x3"] < 5e-5


def test_bicubic_backward_is_the_adjoint():
    rng = np.random.default_rng(4)
    for (H, W, UH, UW) in ((60, 80, 240, 320), (7, 5, 20, 33), (30, 40, 30, 40)):
        g = T(rng.standard_normal((UH, UW, 3)).astype(np.float32))
        x = T(rng.standard_normal((H, W, 3)).astype(np.float32)).requires_grad_(True)
        up = torch.nn.functional.interpolate(x.permute(2, 0, 1)[None], size=(UH, UW), mode="bicubic", align_corners=False)
        (up[0].permute(1, 2, 0) * g).sum().backward()
        got = eng.upsample_bicubic_backward(g.to(DEV), H, W)
        assert relmax(got, x.grad) < 1e-5


def test_dm_step_pose_gradient_vs_oracle():
    """C5's metric: d loss / d predicted pose of the DFNet_dm step (direct_feature_matching.py:322-370) — HIP gradient
    kernels under torch's loss reductions — against autograd through the composition of the two CPU oracles."""
    from types import SimpleNamespace
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.direct_feature_matching import matching_step_grad
    from dfnet_amd.nerfw import HipQuery
    from oracle import dfnet_oracle as dor
    H, W, focal = 64, 96, 80.0
    w = syn.dfnet_weights(3)
    sd = {k: T(v) for k, v in w.items()}
    model, feat_model = DFNet().eval(), DFNet().eval()
    model.load_state_dict(sd, strict=False)
    feat_model.load_state_dict(sd, strict=False)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f32").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=16, N_samples=8, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    setup = dict(pose_scale=0.7, pose_scale2=1.2, move_all_cam_vec=[0., 0.1, 1.0])
    args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=True,
                           combine_loss_w=[0.3, 0.2, 1.0])
    g = torch.Generator().manual_seed(1)
    data = torch.rand(2, 3, H, W, generator=g)
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(2)])
    hist = T(syn.HIST_IDX).repeat(2, 1)
    out = matching_step_grad(args, data, model, feat_model, gt, hist, [H, W, focal], True, DEV, setup, **kw)
    assert out["grad_pose"].shape == (2, 3, 4)
    # oracle composition under autograd, starting from the same predicted pose
    pose_ = out["pose_pred"].cpu().clone().requires_grad_(True)
    pn = pose_.clone()
    pn[:, :3, 3] *= setup["pose_scale"]
    pn[:, :3, 3] += torch.tensor(setup["move_all_cam_vec"])
    pn[:, :3, 3] *= setup["pose_scale2"]
    c, f = {k: T(x) for k, x in cw.items()}, {k: T(x) for k, x in fw.items()}
    rgbs = []
    for b in range(2):
        r = orc.render(H // 4, W // 4, focal / 4, 1 << 30, c, f, T(ea), T(et), 8, 16, 0., 2.5, syn.HIST_IDX, c2w=pn[b])[0]
        rgbs.append(torch.nn.Upsample(size=(H, W), mode='bicubic')(r.permute(2, 0, 1)[None])[0])
    rgb = torch.stack(rgbs)
    feats, _ = dor.dfnet_forward(sd, torch.cat([data, rgb]), True, False, False, H, W)
    ft, fr = feats[0][[0]].permute(1, 0, 2, 3, 4).reshape(2, 128, H, W), feats[1][[0]].permute(1, 0, 2, 3, 4).reshape(2, 128, H, W)
    fl = torch.stack([1 - torch.nn.functional.cosine_similarity(fr[b].reshape(128, -1), ft[b].reshape(128, -1), dim=1, eps=1e-6).mean()
                      for b in range(2)]).mean()
    loss = 0.3 * torch.nn.functional.mse_loss(pose_.reshape(2, 12), gt) + 0.2 * ((rgb - data) ** 2).mean() + 1.0 * fl
    loss.backward()
    assert abs(float(out["loss"]) - float(loss.detach())) < 5e-4 * max(1.0, abs(float(loss.detach())))
    e = relmax(out["grad_pose"], pose_.grad)
    print(f"d loss / d pose: {e:.2e}  |grad| max {float(pose_.grad.abs().max()):.3e}")
    assert e < 2e-3


def _pose_grads64(p, x, G, rel_scale, gen):
    """float64 autograd of loss = sum(pose * G) w.r.t. the encoder / fc_pose parameters, inputs and weights perturbed by round-off."""
    from oracle import dfnet_oracle as dor
    from tests.yardstick import float64_default
    jit = (lambda t: t * (1 + rel_scale * (2 * torch.rand(t.shape, generator=gen, dtype=torch.float64) - 1))) if rel_scale else (lambda t: t)
    with float64_default():
        pp = {k: jit(v.double()).requires_grad_(k.startswith("encoder.") or k.startswith("fc_pose.")) for k, v in p.items()}
        _, pose = dor.dfnet_forward(pp, jit(x.double()), False, True, True)
        (pose * G.double()).sum().backward()
    return {k: v.grad for k, v in pp.items() if v.grad is not None}


@pytest.mark.parametrize("shape", [(2, 3, 64, 96), (1, 3, 72, 104)])
def test_dfnet_parameter_gradients_vs_autograd(dfnet, shape):
    """Weight / bias gradients of the 13 encoder convs and fc_pose for loss = sum(pose * G): HIP wgrad kernels vs torch autograd through
    the CPU oracle, at least FOUR inputs (on until one is at round-off), EVERY ONE of which must either match to 5e-5 or be demonstrably on a gate.  A ReLU / max-pool tie within
    round-off resolves differently under another summation order and moves a whole gradient element (measured over ten seeds,
    tools/gpu_debug_params.py: either arithmetic path is at round-off on about half of them, 3e-3 ... 1.6e-2 on the others — different
    halves).  Since round 6 that explanation is CHECKED per input instead of being granted to seven of eight: the float64 gradient is
    re-evaluated with inputs and weights perturbed by 3e-7 ... 3e-6 relative — no more than the forward's own parity tolerance
    (tests/yardstick.gradient_on_a_gate); an error the float64 gradient's own sensitivity does not reach half of is unexplained and fails.  Every input within 5e-2 relative L2."""
    from oracle import dfnet_oracle as dor
    from tests.yardstick import gradient_on_a_gate
    E, p = dfnet
    tight = 0
    for n_seen, seed in enumerate(range(31, 41)):
        if n_seen >= 4 and tight >= 1:     # at least four inputs, and on until one of them is at round-off
            break
        rng = np.random.default_rng(seed)
        x = T(rng.uniform(0, 1, shape).astype(np.float32))
        G = T(rng.standard_normal((shape[0], 12)).astype(np.float32))
        pp = {k: v.clone().requires_grad_(k.startswith("encoder.") or k.startswith("fc_pose.")) for k, v in p.items()}
        _, pose = dor.dfnet_forward(pp, x, False, True, True)
        (pose * G).sum().backward()
        got = E.backward_params(x.to(DEV), G.to(DEV), precision="f16x3")
        assert len(got) == 28
        worst, worst_l2 = 0.0, 0.0
        for k, g in got.items():
            worst = max(worst, relmax(g, pp[k].grad))
            worst_l2 = max(worst_l2, rel_l2(g, pp[k].grad))
            assert rel_l2(g, pp[k].grad) < 5e-2, k
        if worst < 5e-5:
            tight += 1
            print(f"seed {seed} {shape}: worst parameter-gradient error {worst:.2e} (round-off)")
            continue
        on_gate, moved, at = gradient_on_a_gate(lambda sc, gen: _pose_grads64(p, x, G, sc, gen), worst_l2, seed=seed)
        print(f"seed {seed} {shape}: worst parameter-gradient error {worst:.2e} (relative L2 {worst_l2:.2e}); the float64 gradient itself moves by "
              f"{moved:.2e} under perturbations of {at:.0e} -> {'on a gate' if on_gate else 'UNEXPLAINED'}")
        assert on_gate, (seed, worst, worst_l2, moved)
    assert tight >= 1, "no input at round-off at all"


@pytest.mark.parametrize("bn_batch,with_pose", [(False, True), (False, False), (True, True), (True, False)])
def test_dfnet_all_parameter_gradients_vs_autograd(dfnet, bn_batch, with_pose):
    """Training DFNet itself (run_feature.py:166-230): train() mode (BatchNorm on batch statistics, affine trained) and
    --freezeBN.  Features / pose / batch statistics of the training forward and the gradients of every trained tensor
    (13 encoder convs, fc_pose, 1x1 and 5x5 adaptation convs [+ BatchNorm affine] of the three levels) for
    loss = sum(features * Gf) [+ sum(pose * Gp)], vs torch autograd through the CPU oracle."""
    from oracle import dfnet_oracle as dor
    from tests.yardstick import float64_default, gradient_on_a_gate
    E, p = dfnet
    shape, uH, uW = (2, 3, 48, 64), 24, 40
    tight = 0
    trained = lambda k: k.endswith(("weight", "bias")) and (bn_batch or ".3." not in k)

    def grads64(x, Gf, Gp, rel_scale, gen):   # float64 autograd of the same loss, inputs and weights perturbed by `rel_scale`
        jit = (lambda t: t * (1 + rel_scale * (2 * torch.rand(t.shape, generator=gen, dtype=torch.float64) - 1))) if rel_scale else (lambda t: t)
        with float64_default():
            q = {k: (jit(v.double()) if trained(k) else v.double()).requires_grad_(trained(k)) for k, v in p.items()}
            maps, pose = dor.dfnet_forward(q, jit(x.double()), True, True, with_pose, uH, uW, bn_stats=[] if bn_batch else None)
            loss = (maps[0] * Gf.double()).sum()
            if with_pose:
                loss = loss + (pose * Gp.double()).sum()
            loss.backward()
        return {k: v.grad for k, v in q.items() if v.grad is not None and float(v.grad.abs().max()) > 0 and not (bn_batch and "adapt" in k and k.endswith(".2.bias"))}

    for n_seen, seed in enumerate(range(41, 49)):
        if n_seen >= 3 and tight >= 1:     # at least three inputs, and on until one of them is at round-off; EVERY one tight or on a gate
            break
        rng = np.random.default_rng(seed)
        x = T(rng.uniform(0, 1, shape).astype(np.float32))
        Gf = T(rng.standard_normal((3, shape[0], 128, uH, uW)).astype(np.float32))
        Gp = T(rng.standard_normal((shape[0], 12)).astype(np.float32)) if with_pose else None
        pp = {k: v.clone().requires_grad_(trained(k)) for k, v in p.items()}
        stats = [] if bn_batch else None
        maps, pose = dor.dfnet_forward(pp, x, True, True, with_pose, uH, uW, bn_stats=stats)
        loss = (maps[0] * Gf).sum()
        if with_pose:
            loss = loss + (pose * Gp).sum()
        loss.backward()
        f, po, st = E.forward_train(x.to(DEV), True, with_pose, bn_batch, uH, uW, precision="f16x3")
        assert rel_l2(f, maps[0].detach()) < 5e-6 and relmax(f, maps[0].detach()) < 2e-5
        if with_pose:
            assert relmax(po, pose.detach()) < 1e-5
        if bn_batch:
            for t in range(3):
                assert relmax(st[t, 0], stats[t][0]) < 1e-5 and relmax(st[t, 1], stats[t][1]) < 1e-5
        else:
            assert st is None
        got = E.backward_all_params(x.to(DEV), None if Gp is None else Gp.to(DEV), Gf.to(DEV), bn_batch=bn_batch, precision="f16x3")
        assert len(got) == (46 if bn_batch else 40)
        worst, worst_l2 = 0.0, 0.0
        for k, g in got.items():
            ref = pp[k].grad if pp[k].grad is not None else torch.zeros_like(pp[k])
            if float(ref.abs().max()) == 0.0:
                assert float(g.abs().max()) == 0.0, k
                continue
            if bn_batch and "adapt" in k and k.endswith(".2.bias"):   # d L/d bias of a conv followed by batch-statistics BatchNorm is exactly 0:
                assert float(g.abs().max()) < 1e-3 * float(got[k.replace(".2.bias", ".3.bias")].abs().max()), k   # rounding noise only
                continue
            worst = max(worst, relmax(g, ref))
            worst_l2 = max(worst_l2, rel_l2(g, ref))
            assert rel_l2(g, ref) < 5e-2, (k, rel_l2(g, ref))
        if worst < 5e-5:
            tight += 1
            print(f"seed {seed} bn_batch={bn_batch} pose={with_pose}: worst parameter-gradient error {worst:.2e} (round-off)")
            continue
        on_gate, moved, at = gradient_on_a_gate(lambda sc, gen: grads64(x, Gf, Gp, sc, gen), worst_l2, seed=seed)
        print(f"seed {seed} bn_batch={bn_batch} pose={with_pose}: worst parameter-gradient error {worst:.2e} (relative L2 {worst_l2:.2e}); the float64 "
              f"gradient itself moves by {moved:.2e} under perturbations of {at:.0e} -> {'on a gate' if on_gate else 'UNEXPLAINED'}")
        assert on_gate, (seed, worst, worst_l2, moved)
    assert tight >= 1, "no input at round-off at all"


@pytest.mark.parametrize("bn_batch", [True, False])
def test_dfnet_kept_forward_matches_recompute(dfnet, bn_batch):
    """forward_train(keep) leaves its activations in the caller's workspace; the backward on that tape must give exactly
    the gradients of the recomputing backward (same kernels, same order), and a tape that does not match is refused."""
    E, _ = dfnet
    rng = np.random.default_rng(77)
    x = T(rng.uniform(0, 1, (2, 3, 64, 80)).astype(np.float32)).to(DEV)
    Gf = T(rng.standard_normal((3, 2, 128, 32, 40)).astype(np.float32)).to(DEV)
    Gp = T(rng.standard_normal((2, 12)).astype(np.float32)).to(DEV)
    f0, p0, s0 = E.forward_train(x, True, True, bn_batch, 32, 40)
    f1, p1, s1, tape = E.forward_train(x, True, True, bn_batch, 32, 40, keep=True)
    assert torch.equal(f0, f1) and torch.equal(p0, p1) and (s0 is None or torch.equal(s0, s1))
    ref = E.backward_all_params(x, Gp, Gf, bn_batch=bn_batch)
    got = E.backward_all_params(x, Gp, Gf, bn_batch=bn_batch, tape=tape)
    assert set(ref) == set(got)
    for k in ref:
        assert torch.equal(ref[k], got[k]), k
    with pytest.raises(RuntimeError, match="forward_train"):   # a workspace that no kept forward filled
        E.backward_all_params(x, Gp, Gf, bn_batch=bn_batch, tape=torch.empty_like(tape))
    with pytest.raises(RuntimeError, match="forward_train"):   # right tape, other shape
        E.backward_all_params(x[:1], Gp[:1], Gf[:, :1].contiguous(), bn_batch=bn_batch, tape=tape)


@pytest.mark.parametrize("mode", ["train", "freezebn"])
def test_dfnet_training_step_vs_reference_golden(mode):
    """The same step against the numbers captured from the reference's DFNet module itself (G10: siamese batch, train()
    and --freezeBN): features, pose, batch statistics -> running-statistics update, gradient norms and samples."""
    from dfnet_amd.engine import DfnetEngine
    g = np.load(os.path.join(GOLD, f"g10_dfnet_train_{mode}.npz"))
    r10 = np.random.default_rng(int(g["seed"]))
    x = r10.uniform(0, 1, (4, 3, 32, 48)).astype(np.float32)
    Gt = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    Gr = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    wts = syn.dfnet_weights(seed=3)
    E = DfnetEngine(3, 12, "f16x3").load_numpy(wts)
    batch = mode == "train"
    xd = T(x).to(DEV)
    (ft, fr), pose, st = E.forward_train(xd, False, True, batch, 24, 40)
    cs = int(g["cstride"])
    assert relmax(ft[:, :, ::cs], g["feat_t"]) < 5e-5 and relmax(fr[:, :, ::cs], g["feat_r"]) < 5e-5
    assert relmax(pose, g["pose"]) < 2e-5
    for t in range(3):
        pre = f"adaptation_layers.adapt_layer_{t}.3."
        rm, rv = T(wts[pre + "running_mean"]), T(wts[pre + "running_var"])
        if batch:
            q = 4 * 32 * 48 // 16 ** t
            rm, rv = 0.9 * rm + 0.1 * st[t, 0].cpu(), 0.9 * rv + 0.1 * st[t, 1].cpu() * q / (q - 1)
        assert relmax(rm, g[f"rm{t}"]) < 1e-5 and relmax(rv, g[f"rv{t}"]) < 1e-5
    Gf = torch.cat([T(Gt), T(Gr)], 1).contiguous().to(DEV)   # siamese halves in batch order
    got = E.backward_all_params(xd, T(g["Gp"]).to(DEV), Gf, bn_batch=batch)
    n = 0
    for k, v in got.items():
        flat = v.reshape(-1).cpu()
        ref_n = float(g["gn:" + k])
        if batch and "adapt" in k and k.endswith(".2.bias"):
            continue   # exactly zero in exact arithmetic; the reference's value is rounding noise too
        assert abs(float(flat.norm()) - ref_n) <= 5e-4 * ref_n, (k, float(flat.norm()), ref_n)
        sub = flat[:: max(1, flat.numel() // 256)][:256]
        assert float((sub - T(g["gs:" + k])).abs().max()) <= 2e-3 * float(np.abs(g["gs:" + k]).max()), k
        n += 1
    assert n == (43 if batch else 40)


def test_dfnet_module_trains_pose_path():
    """nn.Module surface: loss.backward() fills .grad of the regressor's conv / fc parameters (and only those), and a
    small gradient step through a torch optimizer lowers the loss of the HIP forward."""
    from dfnet_amd.dfnet import DFNet
    from oracle import dfnet_oracle as dor
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    m = DFNet().to(DEV)
    m.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    m.eval()
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(2)).to(DEV)
    target = torch.zeros(2, 12, device=DEV)
    _, pose = m(x)
    loss0 = ((pose - target) ** 2).mean()
    loss0.backward()
    with_grad = {k for k, q in m.named_parameters() if q.grad is not None}
    assert with_grad == set(m._pose_param_names())
    pp = {k: v.clone().requires_grad_(k in with_grad) for k, v in sd.items()}
    _, rp = dor.dfnet_forward(pp, x.cpu(), False, True, True)
    ((rp - target.cpu()) ** 2).mean().backward()
    assert rel_l2(dict(m.named_parameters())["fc_pose.weight"].grad, pp["fc_pose.weight"].grad) < 1e-4
    assert rel_l2(dict(m.named_parameters())["encoder.0.weight"].grad, pp["encoder.0.weight"].grad) < 5e-2
    g2 = sum(float((q.grad ** 2).sum()) for q in m.parameters() if q.grad is not None)
    opt = torch.optim.SGD(m.parameters(), lr=0.1 * float(loss0.detach()) / g2)   # first-order prediction: loss drops by ~10 %
    opt.step()
    with torch.no_grad():
        _, pose1 = m(x)
    loss1 = float(((pose1 - target) ** 2).mean())
    assert 0.8 * float(loss0) < loss1 < 0.97 * float(loss0), (float(loss0), loss1)


@pytest.mark.parametrize("mode", ["train", "freezebn"])
def test_dfnet_module_training_step_vs_reference_golden(mode):
    """nn.Module surface of DFNet's own training: the reference's recipe line by line (model.train() [+ freeze_bn_layer /
    freeze_bn_layer_train], siamese forward, loss.backward()) fills .grad of every trained tensor with the values
    captured from the reference module (G10), moves the running statistics like nn.BatchNorm2d, and an optimizer step
    followed by eval() re-commits the folded inference weights."""
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.feature_misc import freeze_bn_layer, freeze_bn_layer_train
    g = np.load(os.path.join(GOLD, f"g10_dfnet_train_{mode}.npz"))
    r10 = np.random.default_rng(int(g["seed"]))
    x = T(r10.uniform(0, 1, (4, 3, 32, 48)).astype(np.float32)).to(DEV)
    Gt = T(r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)).to(DEV)
    Gr = T(r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)).to(DEV)
    m = DFNet()
    m.load_state_dict({k: T(v) for k, v in syn.dfnet_weights(seed=3).items()}, strict=False)
    if mode == "freezebn":
        m = freeze_bn_layer(m)
    m.to(DEV)
    m.train()
    if mode == "freezebn":
        m = freeze_bn_layer_train(m)
    feats, pose = m(x, return_feature=True, isSingleStream=False, return_pose=True, upsampleH=24, upsampleW=40)
    assert relmax(feats[0][:, :, ::8], g["feat_t"]) < 5e-5 and relmax(feats[1][:, :, ::8], g["feat_r"]) < 5e-5
    loss = (feats[0] * Gt).sum() + (feats[1] * Gr).sum() + (pose * T(g["Gp"]).to(DEV)).sum()
    loss.backward()
    n = 0
    for k, q in m.named_parameters():
        if "gn:" + k not in g:
            assert q.grad is None, k
            continue
        if mode == "train" and "adapt" in k and k.endswith(".2.bias"):
            continue
        ref_n = float(g["gn:" + k])
        assert abs(float(q.grad.norm()) - ref_n) <= 5e-4 * ref_n, (k, float(q.grad.norm()), ref_n)
        n += 1
    assert n == (43 if mode == "train" else 40)
    sd = m.state_dict()
    for t in range(3):
        pre = f"adaptation_layers.adapt_layer_{t}.3."
        assert relmax(sd[pre + "running_mean"], g[f"rm{t}"]) < 1e-5 and relmax(sd[pre + "running_var"], g[f"rv{t}"]) < 1e-5
        assert int(sd[pre + "num_batches_tracked"]) == (1 if mode == "train" else 0)
    # optimizer step, then a second training forward (device re-pack) and an eval forward (host re-commit of the folded
    # weights): both must agree with the CPU oracle on the UPDATED parameters
    from oracle import dfnet_oracle as dor
    torch.optim.SGD([q for q in m.parameters() if q.requires_grad], lr=1e-7).step()
    p1 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
    with torch.no_grad():
        f2, _ = m(x, return_feature=True, isSingleStream=True, return_pose=False, upsampleH=24, upsampleW=40)
        ref2, _ = dor.dfnet_forward(p1, x.cpu(), True, True, False, 24, 40, bn_stats=[] if mode == "train" else None)
        assert rel_l2(f2[0], ref2[0]) < 5e-6
        m.eval()
        p2 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
        f3, pose3 = m(x, return_feature=True, isSingleStream=True, return_pose=True, upsampleH=24, upsampleW=40)
        ref3, rp3 = dor.dfnet_forward(p2, x.cpu(), True, True, True, 24, 40)
        assert rel_l2(f3[0], ref3[0]) < 5e-6 and relmax(pose3, rp3) < 1e-5

@pytest.mark.gpu
def test_running_statistics_do_not_repack_the_engine():
    """A train()-mode forward moves the BatchNorm running statistics.  Nothing in a batch-statistics step reads the engine's copy
    of them, so that must not trigger the 52-tensor device re-pack before the step's next kernel call; a frozen-BatchNorm forward
    afterwards does read them and must see the moved values (oracle on the module's current buffers)."""
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.feature_misc import freeze_bn_layer_train
    from oracle import dfnet_oracle as dor
    m = DFNet()
    m.load_state_dict({k: T(v) for k, v in syn.dfnet_weights(seed=3).items()}, strict=False)
    m.to(DEV).train()
    x = torch.rand(2, 3, 32, 48, generator=torch.Generator().manual_seed(5)).to(DEV)

    def step():
        feats, pose = m(x, return_feature=True, isSingleStream=True, return_pose=True, upsampleH=24, upsampleW=40)
        (feats[0].square().mean() + pose.square().mean()).backward()

    step()                                             # commits the engine, moves the running statistics
    calls = []
    orig = m._engine.refresh_train_params_device
    m._engine.refresh_train_params_device = lambda ts, *a, **k: (calls.append(1), orig(ts, *a, **k))[1]
    step()                                             # no parameter changed: forward and backward re-pack nothing
    assert calls == []
    m = freeze_bn_layer_train(m)                       # BatchNorm layers to eval(): the next forward normalises with the running statistics
    feats, _ = m(x, return_feature=True, isSingleStream=True, return_pose=True, upsampleH=24, upsampleW=40)
    assert calls == [1]
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
    ref, _ = dor.dfnet_forward(sd, x.cpu(), True, True, False, 24, 40)
    assert rel_l2(feats[0], ref[0]) < 5e-6


def test_dfnet_s_module_training_step_vs_oracle():
    """DFNet_s (one pyramid level, dfnet.py:174-207) through the same training path: train()-mode forward and every
    parameter gradient vs autograd through the oracle, then the device re-pack after an optimizer step."""
    from dfnet_amd.dfnet import DFNet_s
    from oracle import dfnet_oracle as dor
    from tests.yardstick import float64_default, gradient_on_a_gate
    wts = {k: T(v) for k, v in syn.dfnet_weights(seed=3, taps=(64,)).items()}
    tr = lambda k: k.endswith(("weight", "bias"))

    def grads64(x, Gt, Gr, Gp, rel_scale, gen):
        jit = (lambda t: t * (1 + rel_scale * (2 * torch.rand(t.shape, generator=gen, dtype=torch.float64) - 1))) if rel_scale else (lambda t: t)
        with float64_default():
            q = {k: (jit(v.double()) if tr(k) else v.double()).requires_grad_(tr(k)) for k, v in wts.items()}
            maps, rp = dor.dfnet_forward(q, jit(x.double()), True, False, True, 24, 32, taps=(2,), bn_stats=[])
            ((maps[0] * Gt.double()).sum() + (maps[1] * Gr.double()).sum() + (rp * Gp.double()).sum()).backward()
        return {k: v.grad for k, v in q.items() if v.grad is not None and not ("adapt" in k and k.endswith(".2.bias"))}

    tight = 0
    # the gate-flip caveat of test_dfnet_parameter_gradients_vs_autograd, CHECKED per input (tests/yardstick.gradient_on_a_gate): at least
    # three inputs, on until one is at round-off, every one either at round-off or demonstrably on a gate
    for n_seen, seed in enumerate(range(9, 17)):
        if n_seen >= 3 and tight >= 1:
            break
        m = DFNet_s()
        m.load_state_dict(wts, strict=False)
        m.to(DEV).train()
        rng = np.random.default_rng(seed)
        x = T(rng.uniform(0, 1, (4, 3, 48, 64)).astype(np.float32))
        Gt, Gr = (T(rng.standard_normal((1, 2, 128, 24, 32)).astype(np.float32)) for _ in range(2))
        Gp = T(rng.standard_normal((4, 12)).astype(np.float32))
        feats, pose = m(x.to(DEV), return_feature=True, isSingleStream=False, return_pose=True, upsampleH=24, upsampleW=32)
        ((feats[0] * Gt.to(DEV)).sum() + (feats[1] * Gr.to(DEV)).sum() + (pose * Gp.to(DEV)).sum()).backward()
        pp = {k: v.clone().requires_grad_(tr(k)) for k, v in wts.items()}
        stats = []
        maps, rp = dor.dfnet_forward(pp, x, True, False, True, 24, 32, taps=(2,), bn_stats=stats)
        ((maps[0] * Gt).sum() + (maps[1] * Gr).sum() + (rp * Gp).sum()).backward()
        assert rel_l2(feats[0], maps[0].detach()) < 5e-6 and relmax(pose, rp.detach()) < 1e-5
        worst, worst_l2 = 0.0, 0.0
        for k, q in m.named_parameters():
            ref = pp[k].grad
            if "adapt" in k and k.endswith(".2.bias"):
                continue
            assert q.grad is not None, k
            assert rel_l2(q.grad, ref) < 5e-2, k
            worst = max(worst, relmax(q.grad, ref))
            worst_l2 = max(worst_l2, rel_l2(q.grad, ref))
        if worst < 5e-5:
            tight += 1
            print(f"DFNet_s training step, seed {seed}: worst parameter-gradient error {worst:.2e} (round-off)")
            continue
        on_gate, moved, at = gradient_on_a_gate(lambda sc, gen: grads64(x, Gt, Gr, Gp, sc, gen), worst_l2, seed=seed)
        print(f"DFNet_s training step, seed {seed}: worst parameter-gradient error {worst:.2e} (relative L2 {worst_l2:.2e}); the float64 gradient "
              f"itself moves by {moved:.2e} under perturbations of {at:.0e} -> {'on a gate' if on_gate else 'UNEXPLAINED'}")
        assert on_gate, (seed, worst, worst_l2, moved)
    assert tight >= 1, "no input at round-off at all"
    torch.optim.SGD(m.parameters(), lr=1e-7).step()
    p1 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items() if "num_batches" not in k}
    with torch.no_grad():
        f2, _ = m(x.to(DEV), return_feature=True, isSingleStream=True, return_pose=False, upsampleH=24, upsampleW=32)
        ref2, _ = dor.dfnet_forward(p1, x, True, True, False, 24, 32, taps=(2,), bn_stats=[])
    assert rel_l2(f2[0], ref2[0]) < 5e-6


def test_mlp_two_pass_gradient_equals_one_pass(scene):
    """dfn_mlp_fine_saving + dfn_mlp_fine_backward_saved (forward recording the ReLU masks, backward from raw + masks) vs the
    split-f16 forward and the one-pass gradient kernel: same raw, same d L/d (point, viewdir)."""
    E = scene[0]
    rng = np.random.default_rng(21)
    n, Nf = 300, 64       # 19200 points: 75 tiles, the last ray count not a multiple of anything
    o = T(rng.uniform(-0.3, 0.3, (n, 3)).astype(np.float32)).to(DEV)
    d = T(rng.standard_normal((n, 3)).astype(np.float32)).to(DEV)
    v = d / d.norm(dim=-1, keepdim=True)
    hist = T(syn.HIST_IDX).to(DEV)
    z = torch.sort(T(rng.uniform(0.05, 2.4, (n, Nf)).astype(np.float32)), -1)[0].to(DEV)
    graw = T(rng.standard_normal((n, Nf, 9)).astype(np.float32)).to(DEV)
    raw1 = E.mlp_fine(o, d, v, hist, z, precision="f16x3")
    raw2, masks = E.mlp_fine_saving(o, d, v, hist, z)
    assert relmax(raw2, raw1.cpu()) < 1e-6
    g1 = E.mlp_fine_backward(o, d, v, hist, z, graw, precision="f16x3")
    g2 = E.mlp_fine_backward_saved(o, d, v, z, raw2, masks, graw)
    e = relmax(g2, g1.cpu())
    print(f"two-pass vs one-pass MLP gradient: {e:.2e}")
    assert e < 2e-6


def test_render_frames_equals_per_frame_render():
    """rendering.render_frames (B frames as one ray batch, the DFNet_dm step's form) gives the frames and pose gradients of
    B separate render(c2w=...) calls."""
    from dfnet_amd import rendering
    from dfnet_amd.nerfw import HipQuery
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=32, N_samples=16, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    H, W, focal, B = 12, 16, 14.6, 3
    poses = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4] for k in range(B)]).to(DEV)
    hists = torch.stack([T(np.roll(syn.HIST_IDX, k)) for k in range(B)]).to(DEV)
    G = torch.randn(B, H, W, 3, generator=torch.Generator().manual_seed(3)).to(DEV)
    pa = poses.clone().requires_grad_(True)
    rgb_a = rendering.render_frames(H, W, focal, pa, hists, **kw)
    (rgb_a * G).sum().backward()
    pb = poses.clone().requires_grad_(True)
    rgb_b = torch.stack([rendering.render(H, W, focal, c2w=pb[b], img_idx=hists[b], **kw)[0] for b in range(B)])
    (rgb_b * G).sum().backward()
    e_rgb, e_g = relmax(rgb_a, rgb_b.detach().cpu()), relmax(pa.grad, pb.grad.cpu())
    print(f"render_frames vs per-frame: rgb {e_rgb:.2e}, d c2w {e_g:.2e}")
    assert e_rgb < 5e-6 and e_g < 1e-3   # d c2w: a cancelling sum over rays (see the module docstring)


def test_dm_train_step_parameter_gradients_vs_oracle():
    """The whole DFNet_dm optimisation step (direct_feature_matching.py:322-376): gradients that reach the pose
    regressor's parameters through SVD -> scene rescale -> render -> bicubic -> feature extractor -> losses, HIP path vs
    autograd through the composition of the CPU oracles; then the optimizer step itself."""
    from types import SimpleNamespace
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.direct_feature_matching import train_on_batch
    from dfnet_amd.nerfw import HipQuery
    from oracle import dfnet_oracle as dor
    H, W, focal = 64, 96, 80.0
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    model, feat_model = DFNet().to(DEV).eval(), DFNet().to(DEV).eval()
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    feat_model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    for q in feat_model.parameters():
        q.requires_grad_(False)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16x3").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=16, N_samples=8, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    setup = dict(pose_scale=0.7, pose_scale2=1.2, move_all_cam_vec=[0., 0.1, 1.0])
    args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=True,
                           combine_loss_w=[0.3, 0.2, 1.0])
    data = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(1))
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(2)])
    hist = T(syn.HIST_IDX).repeat(2, 1)

    class Capture:  # stands in for the optimizer: keeps the gradients loss.backward() produced
        def __init__(self, m):
            self.m, self.grads = m, None

        def step(self):
            self.grads = {k: q.grad.detach().cpu().clone() for k, q in self.m.named_parameters() if q.grad is not None}

        def zero_grad(self):
            for q in self.m.parameters():
                q.grad = None

    cap = Capture(model)
    loss, psnr = train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], cap, True, DEV, setup, **kw)
    assert set(cap.grads) == set(model._pose_param_names())
    # oracle composition, everything tracked from the regressor's parameters on
    pp = {k: v.clone().requires_grad_(k in cap.grads) for k, v in sd.items()}
    _, pr = dor.dfnet_forward(pp, data, False, True, True)
    pose_ = pr.reshape(2, 3, 4).clone()
    u, s_, v = torch.svd(pose_[:, :3, :3].clone())
    pose_[:, :3, :3] = u @ v.transpose(-2, -1)
    pn = pose_.clone()
    pn[:, :3, 3] *= setup["pose_scale"]
    pn[:, :3, 3] += torch.tensor(setup["move_all_cam_vec"])
    pn[:, :3, 3] *= setup["pose_scale2"]
    c, f = {k: T(x) for k, x in cw.items()}, {k: T(x) for k, x in fw.items()}
    rgbs = []
    for b in range(2):
        r = orc.render(H // 4, W // 4, focal / 4, 1 << 30, c, f, T(ea), T(et), 8, 16, 0., 2.5, syn.HIST_IDX, c2w=pn[b])[0]
        rgbs.append(torch.nn.Upsample(size=(H, W), mode='bicubic')(r.permute(2, 0, 1)[None])[0])
    rgb = torch.stack(rgbs)
    feats, _ = dor.dfnet_forward(sd, torch.cat([data, rgb]), True, False, False, H, W)
    ft, fr = feats[0][[0]].permute(1, 0, 2, 3, 4).reshape(2, 128, H, W), feats[1][[0]].permute(1, 0, 2, 3, 4).reshape(2, 128, H, W)
    fl = torch.stack([1 - torch.nn.functional.cosine_similarity(fr[b].reshape(128, -1), ft[b].reshape(128, -1), dim=1, eps=1e-6).mean()
                      for b in range(2)]).mean()
    ref_loss = 0.3 * torch.nn.functional.mse_loss(pose_.reshape(2, 12), gt) + 0.2 * ((rgb - data) ** 2).mean() + 1.0 * fl
    ref_loss.backward()
    assert abs(float(loss[0]) - float(ref_loss.detach())) < 5e-4 * max(1.0, abs(float(ref_loss.detach())))
    worst = 0.0
    for k, g in cap.grads.items():
        worst = max(worst, rel_l2(g, pp[k].grad))
    print(f"DFNet_dm step: worst relative-L2 error over the 28 parameter gradients {worst:.2e}")
    assert worst < 1e-3


def test_dm_train_step_is_identical_with_and_without_level_pruning():
    """direct_feature_matching.PRUNE_FEATURE_LEVELS: the feature extractor computes only the levels of args.feature_matching_lvl — the
    step's loss, PSNR and all 28 regressor gradients are bit-identical to the all-levels form (what the reference computes and then
    index_selects, direct_feature_matching.py:354-357)."""
    from types import SimpleNamespace
    import dfnet_amd.direct_feature_matching as dfm
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.nerfw import HipQuery
    H, W, focal = 64, 96, 80.0
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16x3").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=16, N_samples=8, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    setup = dict(pose_scale=0.7, pose_scale2=1.2, move_all_cam_vec=[0., 0.1, 1.0])
    data = torch.rand(2, 3, H, W, generator=torch.Generator().manual_seed(1))
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(2)])
    hist = T(syn.HIST_IDX).repeat(2, 1)

    class Capture:
        def __init__(self, m):
            self.m, self.grads = m, None

        def step(self):
            self.grads = {k: q.grad.detach().cpu().clone() for k, q in self.m.named_parameters() if q.grad is not None}

        def zero_grad(self):
            for q in self.m.parameters():
                q.grad = None

    for lvl in ([0], [1], [0, 2]):
        args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=lvl, per_channel=False, combine_loss=True,
                               combine_loss_w=[0.3, 0.2, 1.0])
        out = {}
        # "poison": the pruned step once more with every feature / gradient plane that the "only these levels are read" hints leave
        # unwritten filled with NaN (engine.POISON_UNREAD, DFN_DEBUG_POISON_UNREAD): a consumer that read an unhinted level would turn
        # the loss or a gradient into NaN — they must stay the pruned step's bits
        for prune in (True, False, "poison"):
            model, feat_model = DFNet().to(DEV).eval(), DFNet().to(DEV).eval()
            model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
            feat_model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
            for q in feat_model.parameters():
                q.requires_grad_(False)
            cap = Capture(model)
            dfm.PRUNE_FEATURE_LEVELS = bool(prune)
            eng.POISON_UNREAD = prune == "poison"
            try:
                loss, psnr = dfm.train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], cap, True, DEV, setup, **kw)
            finally:
                dfm.PRUNE_FEATURE_LEVELS = True
                eng.POISON_UNREAD = False
            out[prune] = (float(loss[0]), float(psnr[0]), cap.grads)
        for other in (False, "poison"):
            assert out[True][0] == out[other][0] and out[True][1] == out[other][1], (lvl, other)
            assert set(out[True][2]) == set(out[other][2])
            for k in out[True][2]:
                assert torch.equal(out[True][2][k], out[other][2][k]), (lvl, other, k)
        assert np.isfinite(out["poison"][0]) and all(bool(torch.isfinite(g).all()) for g in out["poison"][2].values())
    with pytest.raises(NameError, match="combine_loss"):   # the reference's step defines `loss` under --combine_loss only (:371-376)
        dfm.train_on_batch(SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=False,
                                           combine_loss_w=[0.3, 0.2, 1.0]), data, model, feat_model, gt, hist, [H, W, focal], cap, True, DEV, setup, **kw)


def test_dm_train_step_at_c5_size_vs_oracle():
    """BASELINE configs[4] at its per-GPU shape — batch 4, 240x320 frames, NeRF-H render 60x80 at 64+128 + bicubic x4, level-0
    feature loss — through train_on_batch with the production precisions (f16 coarse net, split-f16 fine net / gradients /
    DFNet): the loss, d loss / d predicted pose and all 28 regressor gradients against autograd through the composition of
    the CPU oracles (frame by frame: the loss is separable over frames, which bounds the oracle's autograd memory)."""
    from types import SimpleNamespace
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.direct_feature_matching import matching_step_grad, train_on_batch
    from dfnet_amd.nerfw import HipQuery
    from oracle import dfnet_oracle as dor
    B, H, W, focal, Nc, Ni = 4, 240, 320, 585.0 / 2, 64, 128
    torch.set_num_threads(min(32, torch.get_num_threads()))
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    model, feat_model = DFNet().to(DEV).eval(), DFNet().to(DEV).eval()
    model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    feat_model.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    for q in feat_model.parameters():
        q.requires_grad_(False)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=Ni, N_samples=Nc, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    setup = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0., 0., 1.0])
    args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=True,
                           combine_loss_w=[0.3, 0.2, 1.0])
    data = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1))
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(B)])
    hist = T(syn.HIST_IDX).repeat(B, 1)

    class Capture:
        def __init__(self, m):
            self.m, self.grads = m, None

        def step(self):
            self.grads = {k: q.grad.detach().cpu().clone() for k, q in self.m.named_parameters() if q.grad is not None}

        def zero_grad(self):
            for q in self.m.parameters():
                q.grad = None

    out = matching_step_grad(args, data, model, feat_model, gt, hist, [H, W, focal], True, DEV, setup, **kw)
    cap = Capture(model)
    loss, _ = train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], cap, True, DEV, setup, **kw)
    # ---- oracle composition: regressor (tracked) -> SVD -> rescale -> per frame [render -> bicubic -> features -> losses]
    pp = {k: v.clone().requires_grad_(k in cap.grads) for k, v in sd.items()}
    _, pr = dor.dfnet_forward(pp, data, False, True, True)
    pose_ = pr.reshape(B, 3, 4).clone()
    u, s_, v = torch.svd(pose_[:, :3, :3].clone())
    pose_[:, :3, :3] = u @ v.transpose(-2, -1)
    pose_leaf = pose_.detach().clone().requires_grad_(True)   # d loss / d predicted pose is collected here, frame by frame
    c, f = {k: T(x) for k, x in cw.items()}, {k: T(x) for k, x in fw.items()}
    with torch.no_grad():
        ft_all = dor.dfnet_forward(sd, data, True, True, False, H, W)[0][0][0]   # target features, level 0: [B,128,H,W]
    total = 0.3 * torch.nn.functional.mse_loss(pose_leaf.reshape(B, 12), gt)
    total.backward()
    ref_loss = float(total.detach())
    for b in range(B):
        pn = pose_leaf[b].clone()
        t3 = (pn[:3, 3] * setup["pose_scale"] + torch.tensor(setup["move_all_cam_vec"])) * setup["pose_scale2"]
        pn = torch.cat([pn[:3, :3], t3[:, None]], 1)
        r = orc.render(H // 4, W // 4, focal / 4, 1 << 30, c, f, T(ea), T(et), Nc, Ni, 0., 2.5, syn.HIST_IDX, c2w=pn)[0]
        rgb = torch.nn.Upsample(size=(H, W), mode='bicubic')(r.permute(2, 0, 1)[None])
        fr = dor.dfnet_forward(sd, rgb, True, True, False, H, W)[0][0][0][0]       # [128,H,W]
        fl = 1 - torch.nn.functional.cosine_similarity(fr.reshape(128, -1), ft_all[b].reshape(128, -1), dim=1, eps=1e-6).mean()
        part = 0.2 * ((rgb[0] - data[b]) ** 2).sum() / (B * 3 * H * W) + 1.0 * fl / B
        part.backward()
        ref_loss += float(part.detach())
    assert abs(float(loss[0]) - ref_loss) < 2e-4 * max(1.0, abs(ref_loss)), (float(loss[0]), ref_loss)
    e_pose = relmax(out["grad_pose"], pose_leaf.grad)
    pose_.backward(pose_leaf.grad)    # continue into the regressor's parameters (fp32 autograd)
    # The regressor's backward is piecewise linear in ~10^7 ReLU / max-pool gates at this size: units whose pre-activation is
    # within round-off of zero gate differently under another summation order.  Yardstick: the same backward in fp64 — the
    # HIP gradients must stay within 4x of what torch's own fp32 autograd loses against fp64 (measured: conv1_1's weight gradient —
    # a cancelling sum over 307,200 pixels — 2.0e-3 here vs 5.8e-4 for torch fp32; every other tensor 3e-4..6e-4), never above 5e-3.
    p64 = {k: v.double().clone().requires_grad_(k in cap.grads) for k, v in sd.items()}
    _, pr64 = dor.dfnet_forward(p64, data.double(), False, True, True)
    q64 = pr64.reshape(B, 3, 4).clone()
    u, s_, v = torch.svd(q64[:, :3, :3].clone())
    q64[:, :3, :3] = u @ v.transpose(-2, -1)
    q64.backward(pose_leaf.grad.double())
    e_hip = {k: min(rel_l2(g, pp[k].grad), rel_l2(g, p64[k].grad)) for k, g in cap.grads.items()}
    e_t32 = {k: rel_l2(pp[k].grad, p64[k].grad) for k in cap.grads}
    worst_k = max(e_hip, key=e_hip.get)
    print(f"C5 size: loss {float(loss[0]):.6f} vs oracle {ref_loss:.6f}; d loss / d pose {e_pose:.2e}; worst relative L2 over the 28 "
          f"parameter gradients {e_hip[worst_k]:.2e} ({worst_k}); torch fp32 vs fp64 autograd on the same tensor {e_t32[worst_k]:.2e}, "
          f"worst {max(e_t32.values()):.2e}")
    assert e_pose < 2e-3
    assert all(e_hip[k] < max(1e-3, 4 * max(e_t32.values())) and e_hip[k] < 5e-3 for k in e_hip), e_hip


def test_split_f16_gradient_chain_survives_density_only_gradients(scene):
    """Points whose colour / transient gradients vanish next to their density gradient (near-duplicate samples: alpha ~ 0, so only
    d sigma is non-zero, and tiny): the per-point power-of-two scale of the split-f16 chain used to be chosen from the colour
    branches alone and overflowed the d sigma_s slot (inf -> NaN for the whole ray and, through the pose reduction, the frame)."""
    E = scene[0]
    g = torch.Generator().manual_seed(4)
    n, Nf = 7, 64
    o, d = torch.randn(n, 3, generator=g) * 0.3, torch.randn(n, 3, generator=g)
    v = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(n, Nf, generator=g) * 2.5)[0]
    G = torch.zeros(n, Nf, 9)
    G[..., 3] = torch.randn(n, Nf, generator=g) * 2.5e-10        # d sigma_s only ...
    G[..., 7] = torch.randn(n, Nf, generator=g) * 1.2e-12        # ... and a much smaller d sigma_t
    G[0, :, :] = torch.randn(Nf, 9, generator=g)                 # one ordinary ray beside them
    G[1, :, 3] = 1e-30                                           # and one at the edge of fp32
    args = [t.to(DEV) for t in (o, d, v)] + [dev(syn.HIST_IDX), z.to(DEV), G.to(DEV)]
    ref = E.mlp_fine_backward(*args, precision="f32")
    got = E.mlp_fine_backward(*args, precision="f16x3")
    assert bool(torch.isfinite(got).all())
    for r in range(n):
        scale = float(ref[r].abs().max())
        assert float((got[r] - ref[r]).abs().max()) <= 2e-5 * scale + 1e-37, r


# ---------------------------------------------------------------------- fused cosine feature loss of the DFNet_dm step
def _feature_loss_reference(fr, ft, levels):
    """The reference's composition (direct_feature_matching.py:41-50, 114-136, 352-358) in torch on the CPU."""
    idx = torch.tensor(levels)
    def prep(f):
        f = torch.index_select(f, 0, idx).permute(1, 0, 2, 3, 4)
        return f.reshape(f.shape[0], -1, f.shape[3], f.shape[4])
    f_r, f_t = prep(fr), prep(ft)
    cos = torch.nn.CosineSimilarity(dim=1, eps=1e-6)
    per = []
    for b in range(f_r.shape[0]):
        C = f_r.shape[1]
        per.append(1 - cos(f_r[b].reshape(C, -1), f_t[b].reshape(C, -1)).mean())
    return torch.stack(per).mean()


@pytest.mark.parametrize("shape,levels", [((3, 2, 16, 12, 20), [0, 1, 2]), ((3, 4, 8, 7, 9), [0]), ((2, 1, 5, 3, 5), [1]),
                                          ((3, 2, 32, 60, 80), [2, 0])])
def test_feature_cosine_loss_vs_reference_composition(shape, levels):
    """dfn_feature_cosine_forward / _backward against the reference's torch composition (fp64 on the CPU): value and gradient,
    vector and scalar paths (H*W not a multiple of 4), level subsets in any order, a zero row (norm below eps)."""
    from dfnet_amd import feature_misc as fm
    g = torch.Generator().manual_seed(sum(shape))
    fr, ft = torch.randn(*shape, generator=g), torch.randn(*shape, generator=g)
    fr[levels[0], 0, 1] = 0.                                    # ||x|| = 0 <= eps: cos = 0, gradient = y_n / eps
    ft = 0.3 * fr + ft                                          # correlated, like features of a render and its target
    ref_in = fr.double().requires_grad_(True)
    ref = _feature_loss_reference(ref_in, ft.double(), levels)
    ref.backward()
    x = fr.to(DEV).requires_grad_(True)
    loss = fm.feature_loss_batch(x, ft.to(DEV), levels)
    assert loss.shape == () and abs(float(loss.detach()) - float(ref.detach())) < 2e-6
    (3. * loss).backward()
    gref = 3. * ref_in.grad
    got = x.grad.cpu().double()
    zr = (levels[0], 0, 1)                                      # the zero row's gradient is 1e6 times the others': compared on its own
    e0 = float((got[zr] - gref[zr]).abs().max() / gref[zr].abs().max())
    got[zr], gref[zr] = 0., 0.
    e = float((got - gref).abs().max() / gref.abs().max())
    assert e < 2e-6 and e0 < 2e-6, (e, e0)
    for l in range(shape[0]):
        if l not in levels:
            assert float(x.grad[l].abs().max()) == 0.
    # the two halves of one siamese stack [L, 2B, C, H, W], addressed in place
    both = torch.cat([ft, fr], 1).to(DEV)
    B = shape[1]
    loss2 = fm.feature_loss_batch(both[:, B:], both[:, :B], levels)
    assert abs(float(loss2) - float(loss.detach())) < 1e-6     # (a half may start unaligned: scalar instead of 16-byte loads)
    # per_channel=True keeps the reference's own composition (one cosine per pixel over the channels)
    pc = fm.feature_loss_batch(fr.to(DEV), ft.to(DEV), levels, per_channel=True)
    cosd0 = torch.nn.CosineSimilarity(dim=0, eps=1e-6)
    idx = torch.tensor(levels)
    f_r = torch.index_select(fr, 0, idx).permute(1, 0, 2, 3, 4).reshape(shape[1], -1, shape[3] * shape[4])
    f_t = torch.index_select(ft, 0, idx).permute(1, 0, 2, 3, 4).reshape(shape[1], -1, shape[3] * shape[4])
    want = torch.stack([1 - cosd0(f_r[b], f_t[b]).mean() for b in range(shape[1])]).mean()
    assert abs(float(pc) - float(want)) < 2e-6

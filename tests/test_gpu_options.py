"""The remaining keyword options of render() / render_rays on the HIP path (models/rendering.py:245-256, 353-400):
lindisp (depths linear in disparity), ndc (ndc_rays at near = 1), c2w_staticcam, and white_bkgd's error behaviour.
Checked against the REFERENCE's outputs (tests/golden/g14_*.npz) and the CPU oracle; every call goes through the C ABI."""
import numpy as np
import pytest
import torch

from dfnet_amd import engine as eng
from dfnet_amd._lib import DfnError, check
from dfnet_amd import nerf_train, nerfw, rendering
from dfnet_amd import synthetic as syn
from oracle import nerfh_oracle as orc

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def relmax(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert not torch.isnan(a).any()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def dev(x):
    return torch.as_tensor(x).float().to(DEV).contiguous()


@pytest.fixture()
def scene():
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine().load_numpy(cw, fw, ea, et)
    yield E, {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}, T(ea), T(et)
    E.set_render_options(lindisp=False)


def kwargs(E, Nc, Ni, **over):
    kw = dict(network_query_fn=nerfw.HipQuery(E, 65536), perturb=False, N_importance=Ni, N_samples=Nc, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False)
    kw.update(over)
    return kw


TOL = {"f32": 2e-5, "f16x3": 2e-5, "f16": 1e-3}


@pytest.mark.parametrize("prec", ["f32", "f16x3", "f16", "generic"])
def test_lindisp_render_rays_vs_reference_golden(scene, gold, prec):
    """render_rays with lindisp=True against the reference's own output (G14), every arithmetic of the engine."""
    E = scene[0]
    g = gold("g14_render_lindisp")
    Nc, Ni, near, far = int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"])
    E.set_render_options(lindisp=True)
    rgb, disp, acc, raw = E.render_rays(dev(g["rays_o"]), dev(g["rays_d"]), dev(g["hist"])[None], Nc, Ni, near, far, retraw=True,
                                        precision=prec)
    tol = TOL.get(prec, 2e-5)
    errs = dict(rgb=relmax(rgb, g["rgb"]), disp=relmax(disp, g["disp"]), acc=relmax(acc, g["acc"]), raw=relmax(raw, g["raw"]))
    print(f"lindisp {prec}: {errs}")
    assert max(errs.values()) < tol, errs
    E.set_render_options(lindisp=False)   # and the option is really what made the difference
    rgb_lin = E.render_rays(dev(g["rays_o"]), dev(g["rays_d"]), dev(g["hist"])[None], Nc, Ni, near, far, precision=prec)[0]
    assert relmax(rgb_lin, g["rgb"]) > 5e-4   # (this random scene saturates within the first samples: 1.3e-3 apart)


def test_lindisp_depths_bit_exact_vs_oracle(scene):
    """The sampler with lindisp: z_fine against the oracle's sort of cat([z, z_samples]) on the kernel's own densities."""
    E, cw, fw, ea, et = scene
    R, Nc, Ni, near, far = 300, 64, 128, 0.3, 4.0
    sigma = torch.rand(R, Nc, generator=torch.Generator().manual_seed(3)) * 4
    sigma[:5] = 0.                           # empty rays: uniform pdf
    sigma[5:10, 20] = 1e3                    # a wall: every sample in one bin
    sigma[10:80] = sigma[10:80] * 0.05 + 0.05   # thin media: every bin keeps weight although the first bins are 1/180 of the last
    z, w, zs = eng.sample_fine(dev(sigma), Ni, near, far, want_aux=True, lindisp=True)
    zc = orc.coarse_z(torch.full((R, 1), near), torch.full((R, 1), far), Nc, R, lindisp=True)
    assert torch.equal(z.cpu()[:, :1], zc[:, :1]) and float((z.cpu()[:, -1] - zc[:, -1]).abs().max()) <= 0.
    _, w_ref = orc.coarse_weights(sigma, zc)
    mid = .5 * (zc[:, 1:] + zc[:, :-1])
    zs_ref = orc.sample_pdf(mid, w_ref[:, 1:-1], Ni, det=True)
    assert relmax(w, w_ref) < 3e-6
    pdf = (w_ref[:, 1:-1] + 1e-5) / (w_ref[:, 1:-1] + 1e-5).sum(-1, keepdim=True)
    good = pdf.min(-1)[0] > 5e-4             # inverse-CDF sampling is ill-conditioned where the pdf is ~1e-5 (tests/test_gpu_nerfh.py)
    good[:5] = True
    assert int(good.sum()) > 5 and relmax(zs[good.to(DEV)], zs_ref[good]) < 5e-5   # (the error scales with 1 / min pdf: 5e-6 at 2e-3)
    assert bool((zs.cpu() >= mid[:, :1]).all()) and bool((zs.cpu() <= mid[:, -1:]).all())
    srt = torch.sort(torch.cat([zc, zs.cpu()], -1), -1)[0]     # exact merge of the kernel's own samples with the coarse depths
    assert torch.equal(z.cpu(), srt)
    with pytest.raises(DfnError, match="near > 0"):
        eng.sample_fine(dev(sigma), Ni, 0., far, lindisp=True)


def test_lindisp_drop_in_render_and_gradient(scene, gold):
    """rendering.render(lindisp=True): forward equals the golden; d loss / d rays equals autograd through the oracle."""
    E, cw, fw, ea, et = scene
    g = gold("g14_render_lindisp")
    Nc, Ni, near, far = int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"])
    kw = kwargs(E, Nc, Ni, lindisp=True)
    rays = torch.stack([dev(g["rays_o"]), dev(g["rays_d"])])
    rgb, disp, acc, extras = rendering.render(480, 640, 585., rays=rays, near=near, far=far, img_idx=dev(g["hist"])[None], retraw=True, **kw)
    assert relmax(rgb, g["rgb"]) < 1e-3 and relmax(extras["raw"], g["raw"]) < 1e-3      # the engine's default f16 arithmetic
    assert E.lindisp
    G = torch.randn(rays.shape[1], 3, generator=torch.Generator().manual_seed(1))
    rays_t = rays.clone().requires_grad_(True)
    rendering.GRAD_FORWARD_PRECISION = "f32"      # fp32-grade tracked forward (the default tracks the engine's f16 coarse net)
    try:
        rgb_t = rendering.render(480, 640, 585., rays=rays_t, near=near, far=far, img_idx=dev(g["hist"])[None], **kw)[0]
        (rgb_t * G.to(DEV)).sum().backward()
    finally:
        rendering.GRAD_FORWARD_PRECISION = None
    assert relmax(rgb_t, g["rgb"]) < 2e-5
    o_ref, d_ref = T(g["rays_o"]).clone().requires_grad_(True), T(g["rays_d"]).clone().requires_grad_(True)
    view = d_ref / torch.norm(d_ref, dim=-1, keepdim=True)
    n = o_ref.shape[0]
    rows = torch.cat([o_ref, d_ref, torch.full((n, 1), near), torch.full((n, 1), far), view, T(g["hist"])[None].repeat(n, 1)], 1)
    out = orc.render_rays(rows, cw, fw, ea, et, Nc, Ni, lindisp=True)
    (out["rgb_map"] * G).sum().backward()
    # (torch's cumprod backward divides by 1 - alpha: on this saturating scene single rays of the ORACLE's gradient are off by a percent,
    #  tests/test_gpu_train.py::test_generic_width_render_gradient_vs_oracle; hence relative L2 over the batch)
    # measured tolerance (tests/yardstick.py): the same gradient through the oracle in float64; the HIP gradient within 1.5 x the distance
    # torch's own fp32 autograd sits from it (+ 2e-4)
    from tests.yardstick import float64_default, to64
    with float64_default():
        o64, d64 = T(g["rays_o"]).double().requires_grad_(True), T(g["rays_d"]).double().requires_grad_(True)
        view64 = d64 / torch.norm(d64, dim=-1, keepdim=True)
        rows64 = torch.cat([o64, d64, torch.full((n, 1), near), torch.full((n, 1), far), view64, T(g["hist"]).double()[None].repeat(n, 1)], 1)
        out64 = orc.render_rays(rows64, *to64((cw, fw, ea, et)), Nc, Ni, lindisp=True)
        (out64["rgb_map"] * G.double()).sum().backward()
    for name, got, ref, r64 in (("o", rays_t.grad[0].cpu(), o_ref.grad, o64.grad), ("d", rays_t.grad[1].cpu(), d_ref.grad, d64.grad)):
        yard, e = float((ref.double() - r64).norm() / r64.norm()), float((got.double() - r64).norm() / r64.norm())
        print(f"lindisp d loss / d rays_{name} vs float64: relative L2 {e:.2e} (torch fp32: {yard:.2e})")
        assert e <= 1.5 * yard + 2e-4
    with pytest.raises(ValueError, match="near > 0"):
        rendering.render(480, 640, 585., rays=rays, near=0., far=far, img_idx=dev(g["hist"])[None], **kw)
    rendering.render(480, 640, 585., rays=rays, near=near, far=far, img_idx=dev(g["hist"])[None], **kwargs(E, Nc, Ni))
    assert not E.lindisp                   # the option follows the keyword of each call


def test_lindisp_training_forward_vs_oracle(scene):
    """Training-mode render_rays (perturb, noise, random u) with lindisp against the oracle fed the same draws."""
    E, cw, fw, ea, et = scene
    ew = syn.nerfh_weights(0)
    coarse = nerfw.NeRFW('coarse', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27)
    fine = nerfw.NeRFW('fine', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True,
                       encode_transient=True, in_channels_a=50, in_channels_t=20)
    coarse.load_state_dict({k: T(v) for k, v in ew[0].items()})
    fine.load_state_dict({k: T(v) for k, v in ew[1].items()})
    emb_a, emb_t = torch.nn.Embedding(1000, 5), torch.nn.Embedding(1000, 2)
    emb_a.weight.data.copy_(T(ew[2]))
    emb_t.weight.data.copy_(T(ew[3]))
    mods = [m.to(DEV) for m in (coarse, fine, emb_a, emb_t)]
    tr = nerf_train.NerfHTrainer(E, *mods)
    R, Nc, Ni, near, far = 64, 16, 32, 0.4, 2.5
    rng = np.random.default_rng(8)
    ro, rd = orc.get_rays(480, 640, 585.0, T(syn.orbit_pose(4, 8))[:3, :4])
    sel = rng.choice(480 * 640, R, replace=False)
    o, d = ro.reshape(-1, 3)[sel].contiguous(), rd.reshape(-1, 3)[sel].contiguous()
    hist = T(syn.HIST_IDX)[None].float()
    gen = torch.Generator().manual_seed(4)
    t_rand, noise, u = torch.rand(R, Nc, generator=gen), torch.randn(R, Nc, generator=gen), torch.rand(R, Ni, generator=gen)
    rows = torch.cat([o, d, torch.full((R, 1), near), torch.full((R, 1), far), d / d.norm(dim=-1, keepdim=True), hist.repeat(R, 1)], 1)
    with torch.no_grad():
        ref = orc.render_rays_train(rows, cw, fw, ea, et, Nc, Ni, t_rand, noise, u, perturb=1., raw_noise_std=1., lindisp=True)
    E.set_render_options(lindisp=True)
    out = tr.forward(o.to(DEV), d.to(DEV), hist.to(DEV), Nc, Ni, near, far, t_rand.to(DEV), noise.to(DEV), 1., u.to(DEV))
    for k in ("rgb_map", "disp_map", "acc_map", "rgb0", "disp0", "acc0", "z_std", "beta"):
        e = relmax(out[k], ref[k])
        assert e < 3e-5, (k, e)
    assert relmax(out["raw"], ref["raw"]) < 1e-3


def test_ndc_rays_and_render_vs_reference_golden(scene, gold):
    """ndc_rays on the device, render(ndc=True) and render(c2w_staticcam=...) against the reference (G14)."""
    E = scene[0]
    g = gold("g14_render_ndc_staticcam")
    H, W, focal, Nc, Ni = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["Nc"]), int(g["Ni"])
    o, d = rendering.get_rays(H, W, focal, dev(g["c2w"]))
    no, nd = rendering.ndc_rays(H, W, focal, 1., o, d)
    assert no.shape == (H, W, 3) and relmax(no, g["ndc_rays_o"]) < 1e-6 and relmax(nd, g["ndc_rays_d"]) < 1e-6
    for prec, tol in (("f32", 2e-5), ("f16x3", 2e-5), ("f16", 1e-3)):
        E.precision = prec
        rgb, disp, acc, extras = rendering.render(H, W, focal, c2w=dev(g["c2w"]), near=0., far=1., img_idx=dev(g["hist"])[None],
                                                  **kwargs(E, Nc, Ni, ndc=True))
        errs = (relmax(rgb, g["rgb_ndc"]), relmax(disp, g["disp_ndc"]), relmax(acc, g["acc_ndc"]))
        print(f"ndc {prec}: {errs}")
        assert rgb.shape == (H, W, 3) and extras == {} and max(errs) < tol, errs
        rgb, disp, acc, _ = rendering.render(H, W, focal, c2w=dev(g["c2w"]), c2w_staticcam=dev(g["c2w_staticcam"]), near=0., far=2.5,
                                             img_idx=dev(g["hist"])[None], **kwargs(E, Nc, Ni))
        errs = (relmax(rgb, g["rgb_static"]), relmax(disp, g["disp_static"]), relmax(acc, g["acc_static"]))
        print(f"c2w_staticcam {prec}: {errs}")
        assert max(errs) < tol, errs
    E.precision = "f16"
    pose = dev(g["c2w"]).requires_grad_(True)
    with pytest.raises(NotImplementedError, match="autograd"):
        rendering.render(H, W, focal, c2w=pose, near=0., far=1., img_idx=dev(g["hist"])[None], **kwargs(E, Nc, Ni, ndc=True))


def test_white_bkgd_fails_like_the_reference(scene, gold):
    """white_bkgd=True raises TypeError in the reference's NeRF-H path (G14 records it: rendering.py:295 passes it as
    output_transient); the mirror raises the same exception type instead of inventing a behaviour."""
    E = scene[0]
    g = gold("g14_render_ndc_staticcam")
    assert str(g["white_bkgd_raises"]) == "TypeError"
    with pytest.raises(TypeError, match="white_bkgd"):
        rendering.render(6, 8, 7.3, c2w=dev(g["c2w"]), near=0., far=2.5, img_idx=dev(g["hist"])[None], **kwargs(E, 16, 32, white_bkgd=True))
    with pytest.raises(DfnError, match="unknown option"):
        check(E.lib.dfn_nerfh_set_render_options(E.handle, 6), "dfn_nerfh_set_render_options")


def test_coarse_f16_option_keeps_the_fp32_grade_pixel(scene, gold):
    """DFN_RENDER_COARSE_F16: the coarse network (sample placement only) in f16, the fine network in split-f16.  The pixel stays
    within the fp32-grade tolerance of the golden render; the raw samples sit at slightly different depths (f16 densities move the
    inverse-CDF samples), so they are compared through the composited maps only."""
    E = scene[0]
    g = gold("g7_render_image")
    H, W, focal = int(g["H"]), int(g["W"]), float(g["focal"])
    ref = E.render_image(dev(g["c2w"]), H, W, focal, dev(g["hist"]), int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"]),
                         precision="f16x3")[0].clone()
    E.set_render_options(coarse_f16=True)
    try:
        assert E.coarse_f16
        rgb, disp, acc = E.render_image(dev(g["c2w"]), H, W, focal, dev(g["hist"]), int(g["Nc"]), int(g["Ni"]), float(g["near"]),
                                        float(g["far"]), precision="f16x3")
        e = (relmax(rgb, g["rgb"]), relmax(disp, g["disp"]), relmax(acc, g["acc"]))
        print(f"coarse f16 + fine f16x3 vs the reference's render: {e}")
        assert max(e) < 2e-5, e
        assert not torch.equal(rgb, ref)          # the option is live
        E.set_render_options(lindisp=False)       # other options leave it alone
        assert E.coarse_f16
    finally:
        E.set_render_options(coarse_f16=False)
    rgb2 = E.render_image(dev(g["c2w"]), H, W, focal, dev(g["hist"]), int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"]),
                          precision="f16x3")[0]
    assert torch.equal(rgb2, ref)

"""GPU parity of the DFNet feature extractor (through dfn_dfnet_forward) against the golden vectors
captured from the reference's DFNet.forward and against the CPU oracle.

Tolerance (north_star: 1e-3 relative fp32): the exact-fp32 MFMA path and the split-f16 path
("f16x3": hi/lo f16 operands, three f16 MFMAs per product) are held to 2e-5 of the output range; the f16-input path to 3e-3 max / 1.2e-3 relative-L2 (13 conv layers of f16 rounding sit at
~6-7e-4 relative L2 — measured, see DESIGN.md §6 — which is why f32 is this path's default)."""
import numpy as np
import pytest
import torch

from dfnet_amd import engine as eng
from dfnet_amd import synthetic as syn
from oracle import dfnet_oracle as dor

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"


def relmax(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert not torch.isnan(a).any()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def rel_l2(a, b):
    a, b = a.detach().cpu().double(), torch.as_tensor(b).double()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def net():
    w = syn.dfnet_weights(3)
    return eng.DfnetEngine(3, 12).load_numpy(w), {k: T(v) for k, v in w.items()}


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("f16", 3e-3)])
def test_dfnet_golden_small(net, gold, prec, tol):
    E, _ = net
    g = gold("g8_dfnet_small")
    cs = int(g["cstride"])
    x = T(g["x"]).to(DEV)
    (ft, fr), pose = E.forward(x, True, False, True, 32, 48, precision=prec)
    assert ft.shape == (3, 1, 128, 32, 48) and fr.shape == (3, 1, 128, 32, 48)
    for lvl in range(3):
        assert relmax(ft[lvl, :, ::cs], g["siam_t"][lvl]) < tol, lvl
        assert relmax(fr[lvl, :, ::cs], g["siam_r"][lvl]) < tol, lvl
    l2 = torch.sqrt((ft ** 2).sum((1, 2, 3, 4))).cpu().numpy()
    np.testing.assert_allclose(l2, g["siam_t_l2"], rtol=max(tol, 1e-4))
    assert relmax(pose, g["pose"]) < tol
    fs, none = E.forward(x, True, True, False, 40, 56, precision=prec)
    assert none is None and fs.shape == (3, 2, 128, 40, 56)
    for lvl in range(3):
        assert relmax(fs[lvl, :, ::cs], g["single"][lvl]) < tol, lvl
    none, p_only = E.forward(x, precision=prec)
    assert none is None and relmax(p_only, g["pose_only"]) < tol


@pytest.mark.parametrize("prec,tol", [("f32", 2e-5), ("f16x3", 2e-5), ("f16", 3e-3)])
def test_dfnet_golden_120x160(net, gold, prec, tol):
    E, _ = net
    g = gold("g8_dfnet_120x160")
    fs, _ = E.forward(T(g["x"]).to(DEV), True, True, False, 120, 160, precision=prec)
    full = fs[:, 0]
    for lvl in range(3):
        assert relmax(full[lvl, ::8, ::6, ::8], g["sub"][lvl]) < tol, lvl
    l2 = torch.sqrt((full ** 2).sum((1, 2, 3))).cpu().numpy()
    np.testing.assert_allclose(l2, g["l2"], rtol=max(tol, 1e-4))


def test_dfnet_s_golden(gold):
    w = syn.dfnet_weights(3, taps=(64,))
    E = eng.DfnetEngine(1, 12).load_numpy(w)
    g = gold("g8_dfnet_s_small")
    fs, pose = E.forward(T(g["x"]).to(DEV), True, True, True, 32, 48)
    assert fs.shape == (1, 2, 128, 32, 48)
    assert relmax(fs[0, :, ::int(g["cstride"])], g["single"][0]) < 2e-5 and relmax(pose, g["pose"]) < 2e-5


@pytest.mark.parametrize("B,H,W,uH,uW", [(1, 33, 47, 33, 47), (3, 64, 96, 50, 70), (2, 240, 320, 240, 320), (2, 120, 213, 120, 213)])
def test_dfnet_vs_oracle_shapes(net, B, H, W, uH, uW):
    """Ragged sizes (tile remainders, odd pooling) and an upsample that is not the input size."""
    E, p = net
    x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(H * W))
    with torch.no_grad():
        ref, rpose = dor.dfnet_forward(p, x, True, True, True, uH, uW)
    got, pose = E.forward(x.to(DEV), True, True, True, uH, uW, precision="f32")
    for lvl in range(3):
        assert relmax(got[lvl], ref[0][lvl]) < 2e-5, lvl
    assert relmax(pose, rpose) < 2e-5
    gx3, px3 = E.forward(x.to(DEV), True, True, True, uH, uW, precision="f16x3")  # split-f16: fp32-grade at f16 MFMA rate
    for lvl in range(3):
        assert relmax(gx3[lvl], ref[0][lvl]) < 2e-5, lvl
    assert relmax(px3, rpose) < 2e-5
    g16, p16 = E.forward(x.to(DEV), True, True, True, uH, uW, precision="f16")
    for lvl in range(3):
        assert rel_l2(g16[lvl], ref[0][lvl]) < 1.2e-3 and relmax(g16[lvl], ref[0][lvl]) < 3e-3, lvl


def test_dfnet_module_drop_in(gold):
    """The nn.Module mirror: reference state_dict in, reference return convention out."""
    from dfnet_amd.dfnet import DFNet
    g = gold("g8_dfnet_small")
    m = DFNet().eval()
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    missing = m.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys and all("num_batches_tracked" in k for k in missing.missing_keys)
    with torch.no_grad():
        feats, pose = m(T(g["x"]).to(DEV), return_feature=True, isSingleStream=False, return_pose=True,
                        upsampleH=32, upsampleW=48)
    assert isinstance(feats, list) and len(feats) == 2 and feats[0].shape == (3, 1, 128, 32, 48)
    assert relmax(feats[1][:, :, ::8], g["siam_r"]) < 2e-5 and relmax(pose, g["pose"]) < 2e-5
    feats, pose = m(T(g["x"]).to(DEV), return_feature=True, isSingleStream=True, return_pose=False,
                    upsampleH=40, upsampleW=56)
    assert pose is None and len(feats) == 1 and relmax(feats[0][:, :, ::8], g["single"]) < 2e-5


def test_dm_step_forward_vs_oracle():
    """DFNet_dm forward (direct_feature_matching.py:322-376) against the same composition of the two oracles."""
    from types import SimpleNamespace
    from dfnet_amd.dfnet import DFNet
    from dfnet_amd.direct_feature_matching import matching_step_forward
    from dfnet_amd.nerfw import HipQuery
    from oracle import nerfh_oracle as orc
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
    setup = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0., 0., 1.0])
    args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0, 2], per_channel=False, combine_loss=True,
                           combine_loss_w=[0.3, 0.2, 1.0])
    g = torch.Generator().manual_seed(1)
    data = torch.rand(2, 3, H, W, generator=g)
    gt = torch.stack([T(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(2)])
    hist = T(syn.HIST_IDX).repeat(2, 1)
    out = matching_step_forward(args, data, model, feat_model, gt, hist, [H, W, focal], True, DEV, setup, **kw)
    # oracle composition
    with torch.no_grad():
        _, pp = dor.dfnet_forward(sd, data)
        pose = pp.reshape(2, 3, 4).clone()
        u, s, v = torch.svd(pose[:, :3, :3].clone())
        pose[:, :3, :3] = u @ v.transpose(-2, -1)
        assert relmax(out["pose_pred"], pose) < 1e-4
        pose[:, :3, 3] += torch.tensor([0., 0., 1.0])
        c, f = {k: T(x) for k, x in cw.items()}, {k: T(x) for k, x in fw.items()}
        rgbs = []
        for b in range(2):
            c2w = torch.eye(4); c2w[:3, :4] = pose[b]
            r = orc.render(H // 4, W // 4, focal / 4, 32768, c, f, T(ea), T(et), 8, 16, 0., 2.5, syn.HIST_IDX, c2w=c2w)[0]
            rgbs.append(torch.nn.Upsample(size=(H, W), mode='bicubic')(r.permute(2, 0, 1)[None])[0])
        rgb = torch.stack(rgbs)
        assert relmax(out["rgb"], rgb) < 1e-3
        feats, _ = dor.dfnet_forward(sd, torch.cat([data, rgb]), True, False, False, H, W)
        ft, fr = feats[0][[0, 2]].permute(1, 0, 2, 3, 4).reshape(2, 256, H, W), feats[1][[0, 2]].permute(1, 0, 2, 3, 4).reshape(2, 256, H, W)
        fl = torch.stack([1 - torch.nn.functional.cosine_similarity(fr[b].reshape(256, -1), ft[b].reshape(256, -1), dim=1, eps=1e-6).mean() for b in range(2)]).mean()
        photo = ((rgb - data) ** 2).mean()
        pl = torch.nn.functional.mse_loss(pp.reshape(2, 12) * 0 + pose.reshape(2, 12) * 0 + out["pose_pred"].cpu().reshape(2, 12), gt)
        want = 0.3 * pl + 0.2 * photo + 1.0 * fl
    assert abs(float(out["feat_loss"]) - float(fl)) < 2e-4 * max(1.0, abs(float(fl)))
    assert abs(float(out["photo_loss"]) - float(photo)) < 1e-4
    assert abs(float(out["loss"]) - float(want)) < 5e-4 * max(1.0, abs(float(want)))


def test_dfnet_limits(net):
    """Smallest supported image (32x32: one pixel after four pools), argument errors, and batch invariance."""
    E, p = net
    x = torch.rand(3, 3, 32, 32, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref, rpose = dor.dfnet_forward(p, x, True, True, True, 32, 32)
    got, pose = E.forward(x.to(DEV), True, True, True, 32, 32)
    for lvl in range(3):
        assert relmax(got[lvl], ref[0][lvl]) < 2e-5
    assert relmax(pose, rpose) < 2e-5
    one, _ = E.forward(x[1:2].to(DEV), True, True, False, 32, 32)
    assert torch.equal(one[:, 0], got[:, 1])                       # images are independent: same bits alone or in a batch
    with pytest.raises(Exception, match="32"):
        E.forward(torch.rand(1, 3, 16, 64, device=DEV), True, True, False, 16, 64)
    with pytest.raises(Exception, match="even batch"):
        E.forward(x.to(DEV), True, False, False, 32, 32)           # siamese needs an even batch


def test_device_repack_equals_host_commit():
    """After an in-place update of the pose path's parameters on the GPU the module re-packs them on the device
    (dfn_dfnet_refresh_pose_params_device); a fresh module committed from the host with the same values gives the same
    bits for the pose, the features (all three arithmetic modes) and the input gradient."""
    from dfnet_amd.dfnet import DFNet
    sd = {k: T(v) for k, v in syn.dfnet_weights(3).items()}
    m = DFNet().to(DEV).eval()
    m.load_state_dict({k: v.to(DEV) for k, v in sd.items()}, strict=False)
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(9)).to(DEV)
    with torch.no_grad():
        m(x)                                           # first commit (host)
        g = torch.Generator(device=DEV).manual_seed(4)
        for k in m._pose_param_names():                # an "optimizer step": in place, on the device
            q = dict(m.named_parameters())[k]
            q.mul_(1 + 0.05 * torch.randn(q.shape, device=DEV, generator=g))
        calls = []
        orig = m._engine.refresh_pose_params_device
        m._engine.refresh_pose_params_device = lambda ts, precisions=None: (calls.append(1), orig(ts, precisions))[1]
        E1 = m.engine()
        assert calls == [1]                            # took the device path
    ref = DFNet().eval()
    ref.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()}, strict=False)
    E2 = ref.engine()
    G = torch.randn(3, 2, 128, 64, 96, generator=torch.Generator().manual_seed(1)).to(DEV)

    def same(prec):
        a, pa = E1.forward(x, True, True, True, 64, 96, precision=prec)
        b, pb = E2.forward(x, True, True, True, 64, 96, precision=prec)
        assert torch.equal(a, b) and torch.equal(pa, pb), prec
        if prec != "f16":   # gradients run in split-f16 / exact fp32 only (the library refuses plain f16)
            assert torch.equal(E1.backward_input(x, G, precision=prec), E2.backward_input(x, G, precision=prec)), prec

    same("f16x3")                                      # the module re-packed its own precision only ...
    for prec in ("f32", "f16"):                        # ... the other two are stale and refused, not silently outdated
        with pytest.raises(RuntimeError, match="stale"):
            E1.forward(x, True, True, True, 64, 96, precision=prec)
    pose = [dict(m.named_parameters())[k].detach() for k in m._pose_param_names()]
    E1.refresh_pose_params_device(pose, precisions="all")
    for prec in ("f16x3", "f32", "f16"):
        same(prec)


def test_split_f16_survives_large_activations():
    """Trained VGG16 stacks produce activations in the thousands; the split-f16 convs must neither overflow f16 nor lose
    accuracy there: scale the first conv's weights so that activations reach ~5e3 and compare with the exact-fp32 path."""
    w = {k: v.copy() for k, v in syn.dfnet_weights(3).items()}
    w["encoder.0.weight"] *= 400.0     # activations of a few thousand throughout the stack (representable range: 32 500)
    w["encoder.0.bias"] *= 400.0
    E = eng.DfnetEngine(3, 12).load_numpy(w)
    x = torch.rand(1, 3, 64, 96, generator=torch.Generator().manual_seed(6)).to(DEV)
    a, pa = E.forward(x, True, True, True, 64, 96, precision="f16x3")
    b, pb = E.forward(x, True, True, True, 64, 96, precision="f32")
    assert bool(torch.isfinite(a).all()) and float(b.abs().max()) > 100.0
    print("largest feature", float(b.abs().max()))
    for lvl in range(3):   # two fp32-grade paths against each other at activation magnitudes of several thousand
        assert relmax(a[lvl], b[lvl].cpu()) < 1e-4, lvl
    assert relmax(pa, pb.cpu()) < 1e-4


def test_c4_frame_relative_l2_vs_oracle(net):
    """BASELINE configs[3] frame size (480x640, features only): relative L2 of every pyramid level against the CPU
    oracle — the C4 parity metric — for the three arithmetic modes."""
    E, p = net
    x = torch.rand(1, 3, 480, 640, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        ref = dor.dfnet_forward(p, x, True, True, False, 480, 640)[0][0]
    for prec, tol in (("f16x3", 5e-6), ("f32", 5e-6), ("f16", 1.2e-3)):
        got = E.forward(x.to(DEV), True, True, False, 480, 640, precision=prec)[0].cpu()
        l2 = [float((got[l] - ref[l]).norm() / ref[l].norm()) for l in range(3)]
        print(prec, "relative L2 per level", ["%.2e" % v for v in l2])
        assert max(l2) < tol


def test_triplet_loss_fused_vs_reference_golden_and_oracle():
    """dfn_triplet_loss_forward / backward (the three losses of feature/misc.py:355-435): values and gradients against
    the reference's own functions (G11, all four mining cases, contiguous stacks) and — on the two halves of one siamese
    [L,2B,C,H,W] tensor, addressed in place, W not a multiple of the wave — against autograd through the oracle."""
    import os
    from dfnet_amd import feature_misc as fm
    from oracle import dfnet_oracle as dor
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g11_triplet_losses.npz"))
    fns = (fm.triplet_loss, fm.triplet_loss_hard_negative_mining, fm.triplet_loss_hard_negative_mining_plus)
    for case in range(4):
        margin = float(g[f"c{case}_margin"])
        for mining, fn in enumerate(fns):
            f1 = torch.from_numpy(g[f"c{case}_f1"]).to(DEV).requires_grad_(True)
            f2 = torch.from_numpy(g[f"c{case}_f2"]).to(DEV).requires_grad_(True)
            loss = fn(f1, f2, margin=margin)
            (3.0 * loss).backward()
            ref = float(g[f"c{case}_m{mining}_loss"])
            assert abs(float(loss) - ref) <= 2e-6 * max(abs(ref), 1e-3), (case, mining, float(loss), ref)
            for got, key in ((f1.grad, "g1"), (f2.grad, "g2")):
                want = 3.0 * g[f"c{case}_m{mining}_{key}"]
                assert float((got.cpu() - torch.from_numpy(want)).abs().max()) <= 2e-5 * float(np.abs(want).max()) + 1e-9, (case, mining, key)
    gen = torch.Generator().manual_seed(5)
    L, B, C, H, W = 3, 3, 16, 7, 101
    F = torch.randn(L, 2 * B, C, H, W, generator=gen)
    F[:, B + 1] = F[:, 1] + 0.3 * F[:, B + 1]          # some rows inside the margin, some not
    Fd = F.to(DEV).requires_grad_(True)
    loss = fm.triplet_loss_hard_negative_mining_plus(Fd[:, B:], Fd[:, :B], margin=1.0)   # (render, target) as run_feature.py
    loss.backward()
    Fc = F.clone().requires_grad_(True)
    ref, _ = dor.triplet_loss(Fc[:, B:], Fc[:, :B], 1.0, 2)
    ref.backward()
    assert abs(float(loss) - float(ref)) <= 2e-6 * float(ref)
    assert float((Fd.grad.cpu() - Fc.grad).abs().max()) <= 2e-5 * float(Fc.grad.abs().max())
    with pytest.raises(ValueError):
        fm.triplet_loss(Fd[:, :B, :, :, ::2], Fd[:, B:, :, :, ::2])


def test_frame_prep_device_vs_oracle(tmp_path):
    """dfn_frame_prep (dataset front-end on the device): INTER_AREA downscale + luma histogram vs the CPU restatement —
    integer factor (7-Scenes df=2), no resize, a fractional factor, a flat image (one bin), and the dataset class
    returning device items."""
    from dfnet_amd.engine import frame_prep
    from dfnet_amd import datasets
    from oracle import frame_oracle as fo
    rng = np.random.default_rng(12)
    for (h, w, H, W) in ((48, 64, 24, 32), (30, 40, 30, 40), (45, 64, 20, 28), (480, 640, 240, 320)):
        raw = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if h == 480:   # a smooth image: populated neighbouring bins like a photograph
            yy, xx = np.mgrid[0:h, 0:w]
            raw = np.stack([(yy * 255 // h), (xx * 255 // w), ((yy + xx) * 255 // (h + w))], -1).astype(np.uint8)
        img, hist = frame_prep(torch.from_numpy(raw).to(DEV), H, W, 10)
        if h % H == 0 and w % W == 0:
            want = raw.reshape(H, h // H, W, w // W, 3).astype(np.float64).mean((1, 3)) / 255.0
        else:
            want = fo.area_downscale(raw, H, W)
        got = img.permute(1, 2, 0).cpu().numpy()
        assert np.abs(got - want).max() < 2e-7, (h, w, H, W, np.abs(got - want).max())
        ref_hist = fo.luma_histogram(torch.from_numpy(want.transpose(2, 0, 1).astype(np.float32)), 10)
        assert float(hist.sum()) == pytest.approx(float(ref_hist.sum()), abs=2)
        assert float((hist.cpu() - ref_hist).abs().max()) <= 1.0, (hist, ref_hist)   # a luma on a bin edge may fall either side
        if h == 30:
            assert torch.equal(hist.cpu(), ref_hist)
    flat = np.full((16, 16, 3), 128, np.uint8)
    _, hist = frame_prep(torch.from_numpy(flat).to(DEV), 8, 8, 10)
    assert hist.tolist() == [0, 0, 0, 0, 0, 100, 0, 0, 0, 0]
    from tests.test_host_logic import make_scene
    from dfnet_amd import options
    datadir = make_scene(str(tmp_path))
    args = options.nerf_parser().parse_args(["--datadir", datadir, "--dataset_type", "7Scenes", "--df", "2", "--load_pose_avg_stats",
                                             "--render_test"])
    train_dl, _, hwf, *_ = datasets.load_7Scenes_dataloader_NeRF(args)
    img, pose, hist = next(iter(train_dl))
    assert img.is_cuda and hist.is_cuda and img.shape == (1, 3, 24, 32) and hist.shape == (1, 10)
    host = datasets.SevenScenesFrames(train_dl.dataset.files[0].rsplit("/seq-", 1)[0], True, 1, df=2., hist_bin=10, device_prep=False)
    himg, _, hhist = host[0]
    assert float((img[0].cpu() - himg).abs().max()) < 1e-6 and float((hist[0].cpu() - hhist).abs().max()) <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("levels", [(0,), (1,), (0, 2), (2,)])
@pytest.mark.parametrize("single,shape,up", [(True, (2, 3, 64, 96), (64, 96)), (False, (2, 3, 72, 104), (60, 90)), (True, (4, 3, 240, 320), (240, 320))])
def test_forward_levels_equals_full_forward_on_the_levels_asked_for(levels, single, shape, up):
    """dfn_dfnet_forward_levels (the reference's index_select(features, 0, feature_matching_lvl) folded into the forward,
    direct_feature_matching.py:354-357): the planes of the requested levels hold the SAME BITS as the full forward's, the others
    stay zero — split-f16 (side-stream schedule when level 0 has the requested size) and the exact-fp32 path."""
    import torch
    from dfnet_amd import engine as eng, synthetic as syn
    E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
    x = torch.rand(*shape, generator=torch.Generator().manual_seed(5)).to("cuda:0")
    for prec in ("f16x3", "f32"):
        full, _ = E.forward(x, True, single, False, up[0], up[1], precision=prec)
        full = [t.clone() for t in ((full,) if single else full)]
        part, pose = E.forward(x, True, single, False, up[0], up[1], precision=prec, levels=levels)
        assert pose is None
        for f, p in zip(full, (part,) if single else part):
            for t in range(3):
                if t in levels:
                    assert torch.equal(p[t], f[t]), (prec, t)
                else:
                    assert not bool(p[t].any()), (prec, t)


@pytest.mark.gpu
def test_dfnet_s_features_without_pose_launch_the_tap_conv():
    """DFNet_s, features only: conv1_2 is the last tap and the encoder stops there — its only output is the fused 1x1's (the conv used
    to be skipped when it had no other output and the features came from whatever the buffer held).  Fresh engine, first call."""
    import torch
    from dfnet_amd import engine as eng, synthetic as syn
    w = {k: v for k, v in syn.dfnet_weights(3).items() if not k.startswith("adaptation_layers.adapt_layer_1") and
         not k.startswith("adaptation_layers.adapt_layer_2")}
    x = torch.rand(2, 3, 64, 96, generator=torch.Generator().manual_seed(9)).to("cuda:0")
    a, _ = eng.DfnetEngine(1, 12).load_numpy(w).forward(x, True, True, False, 64, 96, precision="f16x3")
    b, pose = eng.DfnetEngine(1, 12).load_numpy(w).forward(x, True, True, True, 64, 96, precision="f16x3")
    assert pose is not None and torch.equal(a, b) and bool(a.abs().sum() > 0)


@pytest.mark.gpu
def test_side_stream_forward_is_repeatable():
    """The split-f16 forward forks the adaptation branches of levels 1-2 onto the handle's side stream (and the level-restricted
    forward skips some of them): interleaved calls at two sizes on one engine — shared workspace, shared events — must return the same
    bits every time."""
    import torch
    from dfnet_amd import engine as eng, synthetic as syn
    E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
    xa = torch.rand(4, 3, 240, 320, generator=torch.Generator().manual_seed(2)).to("cuda:0")
    xb = torch.rand(2, 3, 96, 128, generator=torch.Generator().manual_seed(3)).to("cuda:0")
    ra = E.forward(xa, True, True, False, 240, 320)[0].clone()
    rb = E.forward(xb, True, True, False, 96, 128)[0].clone()
    for _ in range(8):
        assert torch.equal(E.forward(xa, True, True, False, 240, 320)[0], ra)
        assert torch.equal(E.forward(xb, True, True, False, 96, 128)[0], rb)
        assert torch.equal(E.forward(xa, True, True, False, 240, 320, levels=[0])[0][0], ra[0])
        assert torch.equal(E.forward(xa, True, True, False, 240, 320, levels=[1, 2])[0][1:], ra[1:])

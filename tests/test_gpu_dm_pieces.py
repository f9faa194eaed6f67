"""Batched / fused pieces of the DFNet_dm step against their per-frame and torch forms (through the C ABI):
get_rays and its adjoint for the frames of a mini-batch in one launch (/root/reference/script/models/ray_utils.py:5-15 under
feature/direct_feature_matching.py:340-348), the bicubic x4 enlargement and its adjoint for a batch with the NCHW store
(feature/misc.py:230-237, direct_feature_matching.py:344-346), and the combine_loss block (direct_feature_matching.py:359-370)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _poses(B, seed=0):
    from dfnet_amd import synthetic as syn
    return torch.stack([torch.from_numpy(syn.orbit_pose(k + seed, 8))[:3, :4] for k in range(B)]).float().to(DEV)


def test_raygen_frames_equals_per_frame_raygen_and_adjoint():
    from dfnet_amd import engine as eng
    B, H, W, focal = 3, 13, 22, 73.1
    c2w = _poses(B)
    o, d, v = eng.raygen_frames(H, W, focal, c2w)
    for b in range(B):
        ob, db, vb = eng.raygen(H, W, focal, c2w[b])
        assert torch.equal(o[b], ob) and torch.equal(d[b], db) and torch.equal(v[b], vb)
    g = torch.Generator(device=DEV).manual_seed(3)
    go, gd = torch.randn(B, H * W, 3, device=DEV, generator=g), torch.randn(B, H * W, 3, device=DEV, generator=g)
    gc = eng.raygen_frames_backward(H, W, focal, go, gd)
    for b in range(B):
        assert torch.equal(gc[b], eng.raygen_backward(H, W, focal, go[b].contiguous(), gd[b].contiguous()))


@pytest.mark.parametrize("nchw", [False, True])
def test_bicubic_frames_equals_per_frame_bicubic_and_adjoint(nchw):
    from dfnet_amd import engine as eng
    B, h, w, C, H, W = 3, 15, 20, 3, 60, 80
    g = torch.Generator(device=DEV).manual_seed(5)
    x = torch.rand(B, h, w, C, device=DEV, generator=g)
    y = eng.upsample_bicubic_frames(x, H, W, nchw=nchw)
    ref = torch.stack([eng.upsample_bicubic(x[b], H, W) for b in range(B)])
    assert torch.equal(y, ref.permute(0, 3, 1, 2).contiguous() if nchw else ref)
    # and against torch's own bicubic (the oracle of the single-frame kernel)
    t = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(H, W), mode="bicubic", align_corners=False)
    assert float(((y if nchw else y.permute(0, 3, 1, 2)) - t).abs().max()) < 2e-6
    gy = torch.randn(y.shape, device=DEV, generator=g)
    gx = eng.upsample_bicubic_frames_backward(gy, h, w, nchw=nchw)
    gy_nhwc = gy.permute(0, 2, 3, 1).contiguous() if nchw else gy
    refg = torch.stack([eng.upsample_bicubic_backward(gy_nhwc[b], h, w) for b in range(B)])
    assert torch.equal(gx, refg)


def test_dm_combined_loss_equals_the_torch_expression_and_its_autograd():
    from dfnet_amd.feature_misc import dm_combined_loss
    B, H, W = 4, 48, 64
    g = torch.Generator(device=DEV).manual_seed(9)
    rgb0, data = torch.rand(B, 3, H, W, device=DEV, generator=g), torch.rand(B, 3, H, W, device=DEV, generator=g)
    pose0, gt = torch.randn(B, 12, device=DEV, generator=g), torch.randn(B, 12, device=DEV, generator=g)
    feat0 = torch.rand((), device=DEV, generator=g)
    w = [0.3, 0.2, 1.0]
    res = []
    for fused in (True, False):
        rgb, pose_, feat = (t.clone().double().requires_grad_(True) if not fused else t.clone().requires_grad_(True) for t in (rgb0, pose0, feat0))
        if fused:
            loss, photo, pl = dm_combined_loss(rgb, data, pose_, gt, feat, w)
        else:   # the reference's expression, in float64 as the yardstick
            photo = torch.mean((rgb - data.double()) ** 2)
            pl = torch.nn.functional.mse_loss(pose_, gt.double())
            loss = w[0] * pl + w[1] * photo + w[2] * feat
        (loss * 1.7).backward()
        res.append([t.detach().double() for t in (loss, photo, pl, rgb.grad, pose_.grad, feat.grad)])
    for name, a, b in zip(("loss", "photo", "pose", "d rgb", "d pose", "d feat"), *res):
        err = float((a - b).abs().max() / b.abs().max())
        assert err < 5e-7, (name, err)
    assert not res[0][1].requires_grad

"""DFNet's triplet-loss training step WITHOUT the enlarged feature stacks (the reductions "fused into the upsample";
csrc/dfnet_triplet_pyr.hip, dfnet.FeaturePyramid): the loss of feature/misc.py:355-435 on the stacks of feature/dfnet.py:142-160 is
a closed form of the low-resolution pyramid, because the bilinear enlargement is linear and separable.  Checked against
  * the REFERENCE end to end (G16: its DFNet in train() / --freezeBN, its three triplet functions, its autograd): loss, mining
    sums and every parameter gradient;
  * the materialised path of this library (stacks + the fused stack kernels, themselves pinned by G10 / G11) on inputs built to make
    each of the four mining cases win, enlargements up, down and identity."""
import os

import numpy as np
import pytest
import torch

from dfnet_amd import feature_misc as fm
from dfnet_amd import synthetic as syn
from dfnet_amd.dfnet import DFNet, FeaturePyramid
from dfnet_amd.feature_misc import freeze_bn_layer, freeze_bn_layer_train

pytestmark = pytest.mark.gpu
T = torch.from_numpy
DEV = "cuda:0"
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LOSSES = (fm.triplet_loss, fm.triplet_loss_hard_negative_mining, fm.triplet_loss_hard_negative_mining_plus)


def _module(mode, pyramid):
    m = DFNet()
    m.load_state_dict({k: T(v) for k, v in syn.dfnet_weights(seed=3).items()}, strict=False)
    if mode == "freezebn":
        m = freeze_bn_layer(m)
    m.to(DEV)
    m.train()
    if mode == "freezebn":
        m = freeze_bn_layer_train(m)
    m.pyramid_features = pyramid
    return m


def _step(m, x, Gp, k, margin, size, w_f=0.7):
    feats, pose = m(x, return_feature=True, isSingleStream=False, return_pose=True, upsampleH=size[0], upsampleW=size[1])
    lf = LOSSES[k](feats[1], feats[0], margin=margin)        # (features_rgb, features_target), run_feature.py:154
    (w_f * lf + (pose * Gp).sum()).backward()
    grads = {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None}
    for q in m.parameters():
        q.grad = None
    return feats, float(lf.detach()), pose.detach(), grads


@pytest.mark.parametrize("size", [(32, 48), (24, 40)])
@pytest.mark.parametrize("mode", ["train", "freezebn"])
def test_pyramid_triplet_step_vs_reference_golden(mode, size):
    """G16: the reference's own DFNet + triplet loss + backward on six frames (three per stream); the three loss functions.  The
    pyramid path (no stacks) must give the reference's loss, its pose and every one of its parameter gradients, to the tolerances
    the stack path is held to against G10."""
    g = np.load(os.path.join(GOLD, "g16_dfnet_triplet_step.npz"))
    x, Gp = T(g["x"]).to(DEV), T(g["Gp"]).to(DEV)
    for k in range(3):
        tag = f"{mode}_{size[0]}x{size[1]}_m{k}"
        m = _module(mode, True)
        feats, lf, pose, grads = _step(m, x, Gp, k, float(g[tag + ":margin"]), size, float(g["w_f"]))
        assert isinstance(feats[0], FeaturePyramid) and feats[0].shape == (3, 3, 128) + size
        ref = float(g[tag + ":loss_f"])
        assert abs(lf - ref) <= 2e-5 * abs(ref), (tag, lf, ref)
        assert float((pose.cpu() - T(g[tag + ":pose"])).abs().max()) <= 2e-5 * float(np.abs(g[tag + ":pose"]).max())
        n, worst = 0, 0.0
        for name, q in m.named_parameters():
            if f"{tag}:gn:{name}" not in g.files:
                assert name not in grads, name
                continue
            if "adapt" in name and name.endswith((".2.bias", ".3.bias")):
                # a per-channel SHIFT of a level cancels in every difference the triplet loss takes: exactly zero, rounding noise
                assert float(grads[name].norm()) < 1e-5 and float(g[f"{tag}:gn:{name}"]) < 1e-6, name
                continue
            ref_n = float(g[f"{tag}:gn:{name}"])
            got = grads[name].reshape(-1).cpu()
            err = abs(float(got.norm()) - ref_n) / ref_n
            worst = max(worst, err)
            assert err <= 1e-3, (tag, name, float(got.norm()), ref_n)
            rs = T(g[f"{tag}:gs:{name}"])
            sub = got[:: max(1, got.numel() // 256)][:256]
            assert float((sub - rs).abs().max()) <= 4e-3 * float(rs.abs().max()), (tag, name)
            n += 1
        assert n == (40 if mode == "train" else 37)
        print(f"G16 {tag}: loss {lf:.6f} (reference {ref:.6f}), worst gradient-norm error {worst:.1e}")


def _frames(case, gen):
    """Six frames (three per stream) arranged so that mining case `case` has the smallest of the four mean squared differences:
    0: f1 ~ roll f2, 1: f2 ~ roll f1, 2: f1 ~ roll f1, 3: f2 ~ roll f2   (f1 = rendered stream = second half of the batch)."""
    t = torch.rand(3, 3, 32, 48, generator=gen)          # target stream (f2)
    r = torch.rand(3, 3, 32, 48, generator=gen)          # rendered stream (f1)
    mix = lambda a, b: 0.97 * a + 0.03 * b
    if case == 0:
        r = mix(torch.roll(t, 1, 0), r)
    elif case == 1:
        t = mix(torch.roll(r, 1, 0), t)
    elif case == 2:
        r = mix(r[:1].expand(3, -1, -1, -1), r)
    else:
        t = mix(t[:1].expand(3, -1, -1, -1), t)
    return torch.cat([t, r])


@pytest.mark.parametrize("case,size", [(c, s) for s in ((32, 48), (40, 56), (16, 24)) for c in range(4)] + [(2, (96, 64)), (1, (120, 160))])
def test_pyramid_triplet_equals_the_stack_path(case, size):
    """Every mining case, enlargement to the input size (level 0 = identity), beyond it and below it, and steep enlargements (the
    2 x 3 level-2 map to 96 / 120 rows: the backward's output rows run in chunks of its LDS table): the pyramid path against the
    materialised one — same chosen case, loss to 2e-6, all 46 gradients to 2e-5 relative L2."""
    gen = torch.Generator().manual_seed(100 + case)
    x = _frames(case, gen).to(DEV)
    Gp = torch.randn(6, 12, generator=gen).to(DEV)
    out = {}
    for pyramid in (False, True):
        m = _module("train", pyramid)
        feats, lf, pose, grads = _step(m, x, Gp, 2, 0.3, size)
        out[pyramid] = (lf, pose, grads)
        if not pyramid:
            with torch.no_grad():
                mse = torch.nn.functional.mse_loss
                f1, f2 = feats[1], feats[0]
                c = [mse(f1, torch.roll(f2, 1, 1)), mse(f2, torch.roll(f1, 1, 1)), mse(f1, torch.roll(f1, 1, 1)), mse(f2, torch.roll(f2, 1, 1))]
                assert int(torch.argmin(torch.stack(c))) == case, [float(v) for v in c]
    (la, pa, ga), (lb, pb, gb) = out[False], out[True]
    assert abs(la - lb) <= 2e-6 * abs(la), (la, lb)
    assert torch.equal(pa, pb)
    assert set(ga) == set(gb) and len(ga) == 46
    worst = 0.0
    for k in ga:
        if "adapt" in k and k.endswith((".2.bias", ".3.bias")):
            continue   # zero up to rounding noise on both paths (a shift cancels in the loss)
        e = float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30))
        worst = max(worst, e)
    print(f"case {case}, {size}: loss {lb:.6f} vs {la:.6f}, worst relative-L2 gradient difference {worst:.1e}")
    assert worst <= 2e-5


def test_pyramid_misuse_fails_loudly():
    m = _module("train", True)
    x = torch.rand(4, 3, 32, 48, generator=torch.Generator().manual_seed(1)).to(DEV)
    feats, _ = m(x, return_feature=True, upsampleH=32, upsampleW=48)
    with pytest.raises(ValueError, match="two streams"):
        fm.triplet_loss(feats[0], feats[0])
    fm.triplet_loss(feats[1], feats[0])
    with pytest.raises(RuntimeError, match="one triplet loss"):
        fm.triplet_loss(feats[1], feats[0])
    m2 = _module("train", True)
    fb, _ = m2(x, return_feature=True, upsampleH=32, upsampleW=48)
    with pytest.raises(ValueError, match="ONE siamese"):
        fm.triplet_loss(feats[1], fb[0])
    # without a recorded graph (no_grad) the module returns real stacks
    with torch.no_grad():
        fc, _ = m2(x, return_feature=True, upsampleH=32, upsampleW=48)
    assert torch.is_tensor(fc[0]) and fc[0].shape == (3, 2, 128, 32, 48)


@pytest.mark.parametrize("mode", ["train", "freezebn"])
def test_extra_pose_frames_in_the_siamese_pass_equal_the_second_forward(mode):
    """run_feature.py:211-227 with random view synthesis calls the model twice per step — the siamese pair with features, then
    `feat_model(rgb_perturb, False)` for the synthesised views' poses.  `feature_images` puts the synthesised frames behind the pair in
    ONE encoder pass (adaptation layers, BatchNorm statistics and the triplet loss still see the pair only): losses, poses, running
    statistics and all 46 accumulated gradients must be those of the two calls."""
    gen = torch.Generator().manual_seed(21)
    pair = _frames(2, gen).to(DEV)                                  # 3 + 3 frames
    extra = torch.rand(3, 3, 32, 48, generator=gen).to(DEV)         # synthesised views
    Gp, Gv = torch.randn(6, 12, generator=gen).to(DEV), torch.randn(3, 12, generator=gen).to(DEV)
    out = {}
    for merged in (False, True):
        m = _module(mode, True)
        if merged:
            feats, pose_all = m(torch.cat([pair, extra]), True, upsampleH=32, upsampleW=48, feature_images=6)
            pose, vp = pose_all[:6], pose_all[6:]
        else:
            feats, pose = m(pair, True, upsampleH=32, upsampleW=48)
        lf = fm.triplet_loss_hard_negative_mining_plus(feats[1], feats[0], margin=0.3)
        if not merged:
            _, vp = m(extra, False)
        loss = 0.7 * lf + (pose * Gp).sum() + (vp * Gv).sum()
        loss.backward()
        sd = m.state_dict()
        out[merged] = (float(lf.detach()), pose.detach().clone(), vp.detach().clone(),
                       {n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None},
                       {k: sd[k].clone() for k in sd if "running" in k})
    (la, pa, va, ga, ra), (lb, pb, vb, gb, rb) = out[False], out[True]
    assert abs(la - lb) <= 2e-6 * abs(la)
    assert float((pa - pb).abs().max()) <= 1e-5 * float(pa.abs().max()) and float((va - vb).abs().max()) <= 1e-5 * float(va.abs().max())
    assert set(ga) == set(gb) and len(ga) == (46 if mode == "train" else 40)
    for k in ra:
        assert torch.allclose(ra[k], rb[k], rtol=1e-6, atol=1e-7), k           # BatchNorm saw the pair only, both ways
    worst = 0.0
    for k in ga:
        if "adapt" in k and k.endswith((".2.bias", ".3.bias")):
            continue
        worst = max(worst, float((ga[k] - gb[k]).norm() / ga[k].norm().clamp_min(1e-30)))
    print(f"one encoder pass for pair + synthesised views ({mode}): loss {lb:.6f} vs {la:.6f}, worst relative-L2 gradient difference {worst:.1e}")
    assert worst <= 2e-5
    with pytest.raises(NotImplementedError, match="feature_images"):
        _module(mode, False)(torch.cat([pair, extra]), True, upsampleH=32, upsampleW=48, feature_images=6)

"""Pin the CPU oracle (oracle/*.py) to the reference: every comparison here is against
numbers captured from the reference's own modules by tests/golden/make_golden.py."""
import numpy as np
import pytest
import torch

from dfnet_amd import synthetic as syn
from oracle import dfnet_oracle as dor
from oracle import nerfh_oracle as orc

T = torch.from_numpy


def tt(d):
    return {k: T(np.ascontiguousarray(v)) for k, v in d.items()}


def close(a, b, rtol=1e-6, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def nets(W):
    c, f, ea, et = syn.nerfh_weights(seed=0, W=W)
    return tt(c), tt(f), T(ea), T(et)


def test_g1_get_rays(gold):
    g = gold("g1_get_rays")
    o, d = orc.get_rays(int(g["H"]), int(g["W"]), float(g["focal"]), g["c2w"])
    close(o, g["rays_o"], 0, 0)
    close(d, g["rays_d"], 1e-7, 1e-7)


def test_g2_posenc(gold):
    g = gold("g2_posenc")
    close(orc.posenc(T(g["x"]), 10), g["pe_xyz"], 0, 0)
    close(orc.posenc(T(g["d"]), 4), g["pe_dir"], 0, 0)


def test_g3_network_modes(gold):
    for W in (128, 32):
        g = gold(f"g3_nerfw_w{W}")
        c, f, _, _ = nets(W)
        x = T(g["x"])
        close(orc.nerfh_sigma(c, x[:, :63]), g["coarse_sigma"])
        close(orc.nerfh_static(c, x[:, :63], x[:, 63:90]), g["coarse_static"])
        close(orc.nerfh_fine(f, x[:, :63], x[:, 63:90], x[:, 90:140], x[:, 140:160]), g["fine_raw"])


def test_g4_composite(gold):
    g = gold("g4_composite")
    z = T(g["z"])
    acc, w = orc.coarse_weights(T(g["sigma_coarse"])[..., 0], z)
    close(acc, g["acc_coarse"])
    close(w, g["w_coarse"])
    out = orc.composite_fine(T(g["raw"]), z, test_time=True, static_only=True)
    for k in ("rgb", "disp", "acc", "weights", "depth", "beta"):
        close(out[k], g[k], 1e-6, 1e-6)
    tr = orc.composite_fine(T(g["raw"]), z, test_time=False)
    for k in ("rgb", "disp", "acc", "depth", "beta"):
        close(tr[k], g["train_" + k], 1e-6, 1e-6)


def test_g5_sample_pdf(gold):
    g = gold("g5_sample_pdf")
    ka = orc.sample_pdf(torch.linspace(0, 1, 8)[None], T(np.array([[0, 1, 2, 3, 2, 1, 0]], np.float32)), 5)
    close(ka, g["known_answer"], 0, 1e-7)
    close(ka, np.array([[0, .375, .5, .625, 1]], np.float32), 0, 1e-6)
    b, w = T(g["bins"]), T(g["weights"])
    close(orc.sample_pdf(b, w, 128), g["det128"], 0, 0)
    close(orc.sample_pdf(b, w, 17), g["det17"], 0, 0)
    close(orc.sample_pdf(b, w, 40, det=False, u=T(g["u"])), g["rand40"], 0, 0)


def test_g6_render_rays(gold):
    for tag in "abc":
        g = gold("g6_render_rays_" + tag)
        c, f, ea, et = nets(int(g["W"]))
        rows = orc.pack_ray_rows(T(g["rays_o"]), T(g["rays_d"]), float(g["near"]), float(g["far"]), g["hist"])
        assert rows.shape[1] == 21
        out = orc.render_rays(rows, c, f, ea, et, int(g["Nc"]), int(g["Ni"]), retraw=True)
        close(out["raw"], g["raw"], 2e-5, 2e-6)
        close(out["rgb_map"], g["rgb"], 1e-5, 1e-6)
        close(out["disp_map"], g["disp"], 1e-5, 1e-6)
        close(out["acc_map"], g["acc"], 1e-5, 1e-6)


def test_g7_render_image(gold):
    g = gold("g7_render_image")
    c, f, ea, et = nets(128)
    rgb, disp, acc = orc.render(int(g["H"]), int(g["W"]), float(g["focal"]), 100, c, f, ea, et,
                                int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"]), g["hist"],
                                c2w=g["c2w"])
    close(rgb, g["rgb"], 1e-5, 1e-6)
    close(disp, g["disp"], 1e-5, 1e-6)
    close(acc, g["acc"], 1e-5, 1e-6)


def trained_nets():
    """The trained-like NeRF-H weights of tests/golden/trained_nerfh_weights.npz (tools/gpu_train_scene.py) as oracle inputs."""
    import os
    tw = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "trained_nerfh_weights.npz"))
    c = {k[len("coarse."):]: T(tw[k]) for k in tw.files if k.startswith("coarse.")}
    f = {k[len("fine."):]: T(tw[k]) for k in tw.files if k.startswith("fine.")}
    return c, f, T(tw["embedding_a.weight"]), T(tw["embedding_t.weight"])


def test_g15_trained_weights(gold):
    """The oracle against the REFERENCE on trained-like weights (sharp occupancy: sigma up to the hundreds, saturated alphas)."""
    c, f, ea, et = trained_nets()
    g = gold("g15_trained_render_rays")
    rows = orc.pack_ray_rows(T(g["rays_o"]), T(g["rays_d"]), float(g["near"]), float(g["far"]), g["hist"])
    st = {}
    out = orc.render_rays(rows, c, f, ea, et, int(g["Nc"]), int(g["Ni"]), retraw=True, stages=st)
    assert float(np.max(g["raw"][..., 3])) > 20.0 and float(np.min(g["acc"])) > 0.99      # an occupied scene, not the 0.7-density fog of random init
    close(out["raw"], g["raw"], 5e-5, 5e-5)
    close(out["rgb_map"], g["rgb"], 1e-5, 2e-6)
    close(out["disp_map"], g["disp"], 1e-5, 2e-6)
    close(out["acc_map"], g["acc"], 1e-5, 2e-6)
    # the reference's own intermediates (recorded by make_golden.py around rendering.py:292-304): the oracle's stages on those
    assert np.array_equal(st["z_coarse"].numpy(), g["coarse_z"])
    close(st["sigma_coarse"], g["coarse_raw"][..., 0], 2e-5, 2e-5 * float(np.abs(g["coarse_raw"]).max()))
    close(st["weights_coarse"], g["coarse_weights"], 1e-4, 2e-6)
    zref = T(g["z_vals"])
    assert bool((zref[:, 1:] >= zref[:, :-1]).all()) and zref.shape == (64, 192)
    # the inverse CDF is ill-conditioned where the coarse pdf is ~1e-5 (empty space in front of a surface): most samples agree to
    # round-off, a few move within their bin
    dz = (st["z_fine"] - zref).abs()
    assert float(dz.median()) < 1e-6 and float(dz.max()) < 2.5 / 62
    # the fine network and the compositor ON THE REFERENCE'S SAMPLES: arithmetic only, no sampler in between
    raw = orc.query_fine(f, ea, et, rows[:, 0:3][:, None] + rows[:, 3:6][:, None] * zref[..., None], rows[:, 8:11], rows[:, 11:])
    close(raw, g["raw"], 2e-5, 2e-5 * float(np.abs(g["raw"]).max()))
    comp = orc.composite_fine(T(g["raw"]), zref)
    close(comp["rgb"], g["rgb"], 1e-6, 1e-6)
    close(comp["disp"], g["disp"], 1e-6, 1e-6)
    close(comp["weights"], g["weights"], 1e-6, 1e-6)
    g = gold("g15_trained_render_image")
    rgb, disp, acc = orc.render(int(g["H"]), int(g["W"]), float(g["focal"]), 100, c, f, ea, et, int(g["Nc"]), int(g["Ni"]),
                                float(g["near"]), float(g["far"]), g["hist"], c2w=g["c2w"])
    close(rgb, g["rgb"], 1e-5, 2e-6)
    close(disp, g["disp"], 1e-5, 2e-6)
    close(acc, g["acc"], 1e-5, 2e-6)


def test_g9_render_gradients(gold):
    """Autograd through the oracle's render == autograd through the reference's render (pose / ray gradients)."""
    c, f, ea, et = nets(128)
    for tag in "ab":
        g = gold("g9_render_grad_rays_" + tag)
        rgb, go, gd = orc.render_grad_rays(T(g["rays_o"]), T(g["rays_d"]), T(g["G"]), c, f, ea, et, int(g["Nc"]),
                                           int(g["Ni"]), float(g["near"]), float(g["far"]), g["hist"])
        close(rgb, g["rgb"], 1e-5, 1e-6)
        scale = float(np.abs(g["grad_rays_d"]).max())
        close(go, g["grad_rays_o"], 1e-4, 1e-5 * scale)
        close(gd, g["grad_rays_d"], 1e-4, 1e-5 * scale)
    g = gold("g9_render_grad_c2w")
    rgb, gc = orc.render_grad_c2w(int(g["H"]), int(g["W"]), float(g["focal"]), g["c2w"], T(g["G"]), c, f, ea, et,
                                  int(g["Nc"]), int(g["Ni"]), float(g["near"]), float(g["far"]), g["hist"])
    close(rgb, g["rgb"], 1e-5, 1e-6)
    close(gc, g["grad_c2w"], 1e-4, 1e-5 * float(np.abs(g["grad_c2w"]).max()))


def test_g8_dfnet(gold):
    p = tt(syn.dfnet_weights(seed=3))
    g = gold("g8_dfnet_small")
    cs = int(g["cstride"])
    x = T(g["x"])
    with torch.no_grad():
        maps, pose = dor.dfnet_forward(p, x, True, False, True, 32, 48)
        close(maps[0][:, :, ::cs], g["siam_t"], 1e-4, 1e-5)
        close(maps[1][:, :, ::cs], g["siam_r"], 1e-4, 1e-5)
        close(torch.sqrt((maps[0] ** 2).sum((1, 2, 3, 4))), g["siam_t_l2"], 1e-5, 0)
        close(pose, g["pose"], 1e-4, 1e-5)
        maps, pose = dor.dfnet_forward(p, x, True, True, False, 40, 56)
        assert pose is None and len(maps) == 1
        close(maps[0][:, :, ::cs], g["single"], 1e-4, 1e-5)
        close(torch.sqrt((maps[0] ** 2).sum((1, 2, 3, 4))), g["single_l2"], 1e-5, 0)
        maps, pose = dor.dfnet_forward(p, x)
        assert maps is None
        close(pose, g["pose_only"], 1e-4, 1e-5)
        g2 = gold("g8_dfnet_120x160")
        maps, _ = dor.dfnet_forward(p, T(g2["x"]), True, True, False, 120, 160)
        full = maps[0][:, 0]
        close(full[:, ::8, ::6, ::8], g2["sub"], 1e-4, 1e-5)
        close(torch.sqrt((full ** 2).sum((1, 2, 3))), g2["l2"], 1e-5, 0)
        ps = tt(syn.dfnet_weights(seed=3, taps=(64,)))
        gs = gold("g8_dfnet_s_small")
        maps, pose = dor.dfnet_forward(ps, x, True, True, True, 32, 48, taps=(2,))
        close(maps[0][:, :, ::cs], gs["single"], 1e-4, 1e-5)
        close(pose, gs["pose"], 1e-4, 1e-5)


@pytest.mark.parametrize("mode", ["train", "freezebn"])
def test_g10_dfnet_training_step(gold, mode):
    """One training step of the reference's DFNet (train() mode: BatchNorm on batch statistics; --freezeBN: running
    statistics, affine frozen): features, pose, running-statistics update and every parameter gradient."""
    g = gold("g10_dfnet_train_" + mode)
    r10 = np.random.default_rng(int(g["seed"]))
    x = r10.uniform(0, 1, (4, 3, 32, 48)).astype(np.float32)
    Gt = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    Gr = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    assert np.array_equal(x, g["x"]) and abs(float(np.sqrt((Gt ** 2).sum())) - float(g["Gt_l2"])) < 1e-3
    p = tt(syn.dfnet_weights(seed=3))
    frozen = mode == "freezebn"
    trained = lambda k: k.endswith(("weight", "bias")) and not (frozen and ".3." in k)
    pp = {k: v.clone().requires_grad_(trained(k)) for k, v in p.items()}
    stats = None if frozen else []
    maps, pose = dor.dfnet_forward(pp, T(g["x"]), True, False, True, 24, 40, bn_stats=stats)
    cs = int(g["cstride"])
    close(maps[0][:, :, ::cs], g["feat_t"], 1e-4, 2e-5)
    close(maps[1][:, :, ::cs], g["feat_r"], 1e-4, 2e-5)
    close(pose, g["pose"], 1e-4, 1e-5)
    ((maps[0] * T(Gt)).sum() + (maps[1] * T(Gr)).sum() + (pose * T(g["Gp"])).sum()).backward()
    n = 4 * 32 * 48
    for i in range(3):
        rm, rv = p[f"adaptation_layers.adapt_layer_{i}.3.running_mean"], p[f"adaptation_layers.adapt_layer_{i}.3.running_var"]
        if not frozen:
            q = n // (16 ** i)
            mean, var = stats[i]
            rm, rv = 0.9 * rm + 0.1 * mean, 0.9 * rv + 0.1 * var * q / (q - 1)
        close(rm, g[f"rm{i}"], 1e-5, 1e-6)
        close(rv, g[f"rv{i}"], 1e-5, 1e-6)
    n_checked = 0
    for k, v in pp.items():
        if "gn:" + k not in g:
            assert v.grad is None, k
            continue
        flat = v.grad.reshape(-1)
        ref_n = float(g["gn:" + k])
        assert abs(float(flat.norm()) - ref_n) <= 2e-4 * ref_n + 1e-6, (k, float(flat.norm()), ref_n)
        sub = flat[:: max(1, flat.numel() // 256)][:256]
        close(sub, g["gs:" + k], 0, 3e-4 * max(float(np.abs(g["gs:" + k]).max()), 1e-6))
        n_checked += 1
    assert n_checked == (40 if frozen else 46)


def test_g11_triplet_losses(gold):
    """Triplet losses of DFNet's training, all four mining cases: the oracle's written-out definition vs the values and
    autograd gradients of the reference's own three functions."""
    g = gold("g11_triplet_losses")
    for case in range(4):
        margin = float(g[f"c{case}_margin"])
        for mining in range(3):
            f1, f2 = T(g[f"c{case}_f1"]).requires_grad_(True), T(g[f"c{case}_f2"]).requires_grad_(True)
            loss, chosen = dor.triplet_loss(f1, f2, margin, mining)
            if mining == 2:
                assert chosen == case
            loss.backward()
            close(loss.detach(), g[f"c{case}_m{mining}_loss"], 1e-6, 1e-7)
            close(f1.grad, g[f"c{case}_m{mining}_g1"], 1e-5, 1e-8)
            close(f2.grad, g[f"c{case}_m{mining}_g2"], 1e-5, 1e-8)


@pytest.mark.parametrize("tag,mining", [("train_32x48", 0), ("train_32x48", 2), ("train_24x40", 1), ("freezebn_24x40", 2), ("freezebn_32x48", 2)])
def test_g16_dfnet_triplet_training_step(gold, tag, mining):
    """The triplet-loss training step END TO END (run_feature.py:141-162): the reference's DFNet in train() / --freezeBN, its own
    triplet functions on (features_rgb, features_target) of a six-frame siamese batch enlarged to the input size or to 24 x 40, its
    autograd — against the oracle's dfnet_forward + triplet_loss: loss, the four mining sums, pose, every parameter gradient."""
    g = gold("g16_dfnet_triplet_step")
    mode, size = tag.split("_")
    uh, uw = (int(v) for v in size.split("x"))
    key = f"{tag}_m{mining}"
    frozen = mode == "freezebn"
    p = tt(syn.dfnet_weights(seed=3))
    trained = lambda k: k.endswith(("weight", "bias")) and not (frozen and ".3." in k)
    pp = {k: v.clone().requires_grad_(trained(k)) for k, v in p.items()}
    maps, pose = dor.dfnet_forward(pp, T(g["x"]), True, False, True, uh, uw, bn_stats=None if frozen else [])
    f_t, f_r = maps[0], maps[1]
    close(dor.triplet_loss_cases(f_r, f_t).detach(), g[key + ":mse"], 2e-5, 1e-7)
    loss, _ = dor.triplet_loss(f_r, f_t, float(g[key + ":margin"]), mining)
    close(loss.detach(), g[key + ":loss_f"], 2e-5, 1e-7)
    close(pose.detach(), g[key + ":pose"], 1e-4, 1e-5)
    (float(g["w_f"]) * loss + (pose * T(g["Gp"])).sum()).backward()
    n = 0
    for k, v in pp.items():
        if f"{key}:gn:{k}" not in g:
            assert v.grad is None, k
            continue
        if "adapt" in k and k.endswith((".2.bias", ".3.bias")):
            assert float(v.grad.norm()) < 1e-6 and float(g[f"{key}:gn:{k}"]) < 1e-6
            continue   # a per-channel SHIFT of a level cancels in every difference the triplet loss takes: exactly zero, rounding noise
        ref_n = float(g[f"{key}:gn:{k}"])
        flat = v.grad.reshape(-1)
        assert abs(float(flat.norm()) - ref_n) <= 3e-4 * ref_n + 1e-6, (k, float(flat.norm()), ref_n)
        rs = g[f"{key}:gs:{k}"]
        close(flat[:: max(1, flat.numel() // 256)][:256], rs, 0, 5e-4 * max(float(np.abs(rs).max()), 1e-6))
        n += 1
    assert n == (37 if frozen else 40)


def _train_rows(g):
    o, d = T(g["rays_o"]), T(g["rays_d"])
    return orc.pack_ray_rows(o, d, float(g["near"]), float(g["far"]), g["hist"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g12_render_rays_training_mode(gold, tag):
    """Training-mode render_rays (perturb = 1, test_time = False, raw_noise_std 0 / 1): the reference's own outputs and
    extras with its recorded random draws as inputs."""
    g = gold(f"g12_render_train_{tag}")
    c, f, ea, et = nets(128)
    with torch.no_grad():
        out = orc.render_rays_train(_train_rows(g), c, f, ea, et, int(g["Nc"]), int(g["Ni"]), T(g["t_rand"]), T(g["noise"]),
                                    T(g["u"]), perturb=1., raw_noise_std=float(g["raw_noise_std"]))
    for k_ref, k in (("rgb", "rgb_map"), ("disp", "disp_map"), ("acc", "acc_map"), ("raw", "raw"), ("rgb0", "rgb0"),
                     ("disp0", "disp0"), ("acc0", "acc0"), ("z_std", "z_std"), ("transient_sigmas", "transient_sigmas"),
                     ("beta", "beta")):
        close(out[k], g[k_ref], 2e-5, 2e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g13_training_step(gold, tag):
    """One optimisation step (run_nerf.py:50-66): NerfWLoss terms, PSNR and the gradient of every parameter of both
    networks and both embedding tables (norm + 256 strided samples each, the touched embedding rows in full)."""
    g12, g = gold(f"g12_render_train_{tag}"), gold(f"g13_train_step_{tag}")
    c, f, ea, et = nets(128)
    ld, ps, grads, _ = orc.train_step(_train_rows(g12), T(g["target"]), c, f, ea, et, int(g12["Nc"]), int(g12["Ni"]), T(g12["t_rand"]),
                                      T(g12["noise"]), T(g12["u"]), perturb=1., raw_noise_std=float(g12["raw_noise_std"]))
    for k in ("c_l", "f_l", "b_l", "s_l"):
        close(ld[k], g["loss_" + k], 2e-5, 1e-7)
    close(ps, g["psnr"][0], 1e-5, 1e-5)
    names = [k[3:] for k in g if k.startswith("gn:")]
    assert sorted(names) == sorted(grads), set(names) ^ set(grads)
    for k in names:
        flat = grads[k].reshape(-1)
        gn = float(g["gn:" + k])
        assert abs(float(flat.norm()) - gn) <= 1e-4 * gn + 1e-12, k
        close(flat[:: max(1, flat.numel() // 256)][:256], g["gs:" + k], 0, 2e-4 * gn / np.sqrt(flat.numel()) + 1e-9)
    rows = T(g["emb_rows"])
    close(grads["embedding_a.weight"][rows], g["ga_rows"], 1e-4, 1e-8)
    close(grads["embedding_t.weight"][rows], g["gt_rows"], 1e-4, 1e-8)
    # the same step's gradient w.r.t. the RAYS (the reference's training render is differentiable w.r.t. them): pins the oracle's
    # autograd for tests/test_gpu_train.py::test_training_render_ray_gradients
    go, gd = orc.train_step_grad_rays(T(g12["rays_o"]), T(g12["rays_d"]), float(g12["near"]), float(g12["far"]), g12["hist"], T(g["target"]),
                                      c, f, ea, et, int(g12["Nc"]), int(g12["Ni"]), T(g12["t_rand"]), T(g12["noise"]), T(g12["u"]), perturb=1.,
                                      raw_noise_std=float(g12["raw_noise_std"]))
    ref = T(g["g_rays"])
    assert float((go - ref[0]).norm() / ref[0].norm()) < 2e-4 and float((gd - ref[1]).norm() / ref[1].norm()) < 2e-4


def test_g14_render_options(gold):
    """lindisp, ndc and c2w_staticcam of render() (rendering.py:269-273, 364-376); white_bkgd raises in the reference."""
    c, f, ea, et = nets(128)
    g = gold("g14_render_lindisp")
    rows = orc.pack_ray_rows(T(g["rays_o"]), T(g["rays_d"]), float(g["near"]), float(g["far"]), g["hist"])
    out = orc.render_rays(rows, c, f, ea, et, int(g["Nc"]), int(g["Ni"]), retraw=True, lindisp=True)
    close(out["raw"], g["raw"], 2e-5, 2e-6)
    close(out["rgb_map"], g["rgb"], 1e-5, 1e-6)
    close(out["disp_map"], g["disp"], 1e-5, 1e-6)
    close(out["acc_map"], g["acc"], 1e-5, 1e-6)
    g = gold("g14_render_ndc_staticcam")
    H, W, focal, Nc, Ni = int(g["H"]), int(g["W"]), float(g["focal"]), int(g["Nc"]), int(g["Ni"])
    o, d = orc.get_rays(H, W, focal, g["c2w"])
    no, nd = orc.ndc_rays(H, W, focal, 1., o, d)
    close(no, g["ndc_rays_o"], 1e-6, 1e-6)
    close(nd, g["ndc_rays_d"], 1e-6, 1e-6)
    rgb, disp, acc = orc.render(H, W, focal, 100, c, f, ea, et, Nc, Ni, 0., 1., g["hist"], c2w=g["c2w"], ndc=True)
    close(rgb, g["rgb_ndc"], 1e-5, 1e-6)
    close(disp, g["disp_ndc"], 1e-5, 1e-6)
    close(acc, g["acc_ndc"], 1e-5, 1e-6)
    rgb, disp, acc = orc.render(H, W, focal, 100, c, f, ea, et, Nc, Ni, 0., 2.5, g["hist"], c2w=g["c2w"],
                                c2w_staticcam=g["c2w_staticcam"])
    close(rgb, g["rgb_static"], 1e-5, 1e-6)
    close(disp, g["disp_static"], 1e-5, 1e-6)
    close(acc, g["acc_static"], 1e-5, 1e-6)
    assert str(g["white_bkgd_raises"]) == "TypeError"

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference; never on the GPU box).  It imports
the reference's `models/{ray_utils,nerfw,rendering}.py` and `feature/dfnet.py` from
/root/reference/script, feeds them seeded inputs and seed-generated weights
(dfnet_amd/synthetic.py) and stores inputs + outputs as small .npz files.  Nothing of the
reference's source is stored — only numbers.

Modules the image lacks are registered as empty stand-ins before import, only so that the
reference's *unrelated* import lines succeed: `imageio` (PNG/MP4 writes, unused here).  The
VGG16 layer stack is third-party arithmetic (torchvision==0.10.0, requirements.txt:96, call
site feature/dfnet.py:90-92) that is absent from /root/reference and from this image; the
fixture G8 restates its published architecture (cfg "D": 13 x [Conv3x3 pad1 + ReLU(inplace)]
+ 5 x MaxPool(2,2)) as a plain nn.Sequential so that the reference's DFNet.forward — the code
that IS under /root/reference — runs on top of it.  Parity of that third-party stack is
therefore pinned by construction only ("parity unpinned" by any reference test).

Usage:  python tests/golden/make_golden.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/script"
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

sys.modules.setdefault("imageio", types.ModuleType("imageio"))

from dfnet_amd import synthetic as syn  # noqa: E402


def _install_vgg_stub():
    tv = types.ModuleType("torchvision")
    models = types.ModuleType("torchvision.models")

    class _VGG(torch.nn.Module):
        def __init__(self):
            super().__init__()
            layers, cin = [], 3
            for v in syn.VGG16_CFG:
                if v == "M":
                    layers.append(torch.nn.MaxPool2d(kernel_size=2, stride=2))
                else:
                    layers += [torch.nn.Conv2d(cin, v, kernel_size=3, padding=1), torch.nn.ReLU(inplace=True)]
                    cin = v
            self.features = torch.nn.Sequential(*layers)

    models.vgg16 = lambda pretrained=False, **kw: _VGG()
    tv.models = models
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = models


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def load_into(module, weights):
    sd = {k: t(v) for k, v in weights.items()}
    missing = module.load_state_dict(sd, strict=False)
    extra = [k for k in missing.missing_keys if "num_batches_tracked" not in k]
    assert not extra and not missing.unexpected_keys, (extra, missing.unexpected_keys)


ONLY = os.environ.get("GOLDEN_ONLY", "")  # e.g. GOLDEN_ONLY=g9 rewrites only the fixtures whose name starts with g9


def save(name, **arrs):
    if ONLY and not name.startswith(ONLY):
        return
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  " + " ".join(f"{k}{list(v.shape)}" for k, v in out.items()))


def rand_c2w(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] = -q[:, 0]
    m = np.concatenate([q, rng.uniform(-0.3, 0.3, (3, 1))], 1).astype(np.float32)
    return m


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from models import nerfw, ray_utils, rendering  # the reference

    rng = np.random.default_rng(1234)

    # ---------------- G1: get_rays
    c2w = rand_c2w(rng)
    ro, rd = ray_utils.get_rays(4, 6, 5.0, t(c2w))
    save("g1_get_rays", H=4, W=6, focal=5.0, c2w=c2w, rays_o=ro, rays_d=rd)

    # ---------------- G2: positional encodings
    embed_fn, ch_xyz, _ = nerfw.get_embedder(10, 0)
    embeddirs_fn, ch_dir, _ = nerfw.get_embedder(4, 0)
    assert (ch_xyz, ch_dir) == (63, 27)
    x = rng.uniform(-3, 3, (7, 3)).astype(np.float32)
    x[0] = [0.0, 1.5, -2.75]
    dvec = rng.standard_normal((7, 3)).astype(np.float32)
    dvec /= np.linalg.norm(dvec, axis=1, keepdims=True)
    save("g2_posenc", x=x, pe_xyz=embed_fn(t(x)), d=dvec, pe_dir=embeddirs_fn(t(dvec)))

    # ---------------- networks with seeded weights
    nets = {}
    for W in (128, 32):
        cw, fw, ea, et = syn.nerfh_weights(seed=0, W=W)
        coarse = nerfw.NeRFW("coarse", D=8, W=W, skips=[4], in_channels_xyz=63, in_channels_dir=27)
        fine = nerfw.NeRFW("fine", D=8, W=W, skips=[4], in_channels_xyz=63, in_channels_dir=27,
                           encode_appearance=True, encode_transient=True, in_channels_a=50, in_channels_t=20)
        load_into(coarse, cw)
        load_into(fine, fw)
        assert list(coarse.state_dict().keys()) == list(cw.keys())
        assert list(fine.state_dict().keys()) == list(fw.keys())
        emb_a = torch.nn.Embedding(1000, 5)
        emb_t = torch.nn.Embedding(1000, 2)
        emb_a.weight.data.copy_(t(ea))
        emb_t.weight.data.copy_(t(et))
        nets[W] = (coarse.eval(), fine.eval(), emb_a, emb_t)

    # ---------------- G3: NeRFW forward modes
    with torch.no_grad():
        for W in (128, 32):
            coarse, fine, _, _ = nets[W]
            xin = rng.uniform(-1, 1, (33, 160)).astype(np.float32)
            xin[:, :63] = embed_fn(t(rng.uniform(-2, 2, (33, 3)).astype(np.float32))).numpy()
            save(f"g3_nerfw_w{W}", x=xin,
                 coarse_sigma=coarse(t(xin[:, :63]), sigma_only=True),
                 coarse_static=coarse(t(xin[:, :90]), output_transient=False),
                 fine_raw=fine(t(xin), output_transient=True))

    # ---------------- G4: raw2outputs_NeRFW
    with torch.no_grad():
        R, N = 11, 24
        z = np.sort(rng.uniform(0.0, 2.5, (R, N)).astype(np.float32), -1)
        z[0] = np.linspace(0, 2.5, N, dtype=np.float32)
        raw = rng.uniform(0, 1, (R, N, 9)).astype(np.float32)
        raw[..., 3] = rng.uniform(0, 6, (R, N))  # static sigma
        raw[..., 7] = rng.uniform(0, 3, (R, N))  # transient sigma
        raw[1, :, 3] = 0.0
        raw[2, :, 7] = 0.0
        raw[3, :, 3] = 40.0
        rd_ = t(rng.standard_normal((R, 3)).astype(np.float32))
        sig_c = rng.uniform(-1, 5, (R, N, 1)).astype(np.float32)  # coarse: relu matters
        _, _, acc_c, w_c, _, _, _ = rendering.raw2outputs_NeRFW(t(sig_c), t(z), rd_, 0., False, test_time=True, typ="coarse")
        rgb, disp, acc, w, depth, tsig, beta = rendering.raw2outputs_NeRFW(
            t(raw), t(z), rd_, 0., True, 0.1, False, True, typ="fine")
        rgb2, disp2, acc2, w2, depth2, tsig2, beta2 = rendering.raw2outputs_NeRFW(
            t(raw), t(z), rd_, 0., True, 0.1, False, False, typ="fine")
        save("g4_composite", z=z, raw=raw, sigma_coarse=sig_c, acc_coarse=acc_c, w_coarse=w_c,
             rgb=rgb, disp=disp, acc=acc, weights=w, depth=depth, beta=beta,
             train_rgb=rgb2, train_disp=disp2, train_acc=acc2, train_depth=depth2, train_beta=beta2)

    # ---------------- G5: sample_pdf
    with torch.no_grad():
        ka = rendering.sample_pdf(torch.linspace(0, 1, 8)[None], t(np.array([[0, 1, 2, 3, 2, 1, 0]], np.float32)), 5, det=True)
        R, Nb = 9, 63
        bins = np.sort(rng.uniform(0, 2.5, (R, Nb)).astype(np.float32), -1)
        bins[0] = np.linspace(0.02, 2.48, Nb, dtype=np.float32)
        wts = rng.uniform(0, 1, (R, Nb - 1)).astype(np.float32) ** 4
        wts[1] = 0.0            # all-zero row -> uniform pdf
        wts[2, 5:] = 0.0        # mass only at the front (u=1 edge, flat cdf tail)
        wts[3, :-3] = 0.0       # mass only at the back
        wts[4] = 1e-7
        det = rendering.sample_pdf(t(bins), t(wts), 128, det=True)
        det17 = rendering.sample_pdf(t(bins), t(wts), 17, det=True)
        np.random.seed(0)
        u = np.random.rand(R, 40).astype(np.float32)  # what pytest=True draws (rendering.py:40-47)
        rnd = rendering.sample_pdf(t(bins), t(wts), 40, det=False, pytest=True)
        save("g5_sample_pdf", known_answer=ka, bins=bins, weights=wts, det128=det, det17=det17, u=u, rand40=rnd)

    # ---------------- G6 / G7: render_rays and render
    def kwargs_for(W, Nc, Ni):
        coarse, fine, emb_a, emb_t = nets[W]
        q = lambda inputs, viewdirs, ts, network_fn, typ, embedding_a, embedding_t, output_transient, test_time: \
            nerfw.run_network_NeRFW(inputs, viewdirs, ts, network_fn, embed_fn=embed_fn, embeddirs_fn=embeddirs_fn,
                                    typ=typ, embedding_a=embedding_a, embedding_t=embedding_t,
                                    output_transient=output_transient, netchunk=65536, test_time=test_time)
        return dict(network_query_fn=q, perturb=False, N_importance=Ni, network_fine=fine, N_samples=Nc,
                    network_fn=coarse, use_viewdirs=True, white_bkgd=False, raw_noise_std=0.,
                    embedding_a=emb_a, embedding_t=emb_t, test_time=True, ndc=False, lindisp=False)

    hist = syn.HIST_IDX
    with torch.no_grad():
        for tag, W, R, Nc, Ni in (("a", 128, 64, 8, 16), ("b", 128, 16, 64, 128), ("c", 32, 32, 32, 64)):
            c2w = syn.orbit_pose(3, 8)[:3, :4]
            ro, rd = ray_utils.get_rays(480, 640, 585.0, t(c2w))
            sel = rng.choice(480 * 640, R, replace=False)
            ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
            rgb, disp, acc, extras = rendering.render(480, 640, 585.0, chunk=32768, rays=torch.stack([ro, rd], 0),
                                                       near=0., far=2.5, img_idx=t(hist)[None], retraw=True,
                                                       **kwargs_for(W, Nc, Ni))
            save(f"g6_render_rays_{tag}", W=W, Nc=Nc, Ni=Ni, near=0., far=2.5, hist=hist, rays_o=ro, rays_d=rd,
                 rgb=rgb, disp=disp, acc=acc, raw=extras["raw"])
        H, Wd, focal = 12, 16, 14.6
        c2w = syn.orbit_pose(1, 8)
        rgb, disp, acc, extras = rendering.render(H, Wd, focal, chunk=100, c2w=t(c2w)[:3, :4], near=0., far=2.5,
                                                   img_idx=t(hist)[None], **kwargs_for(128, 64, 128))
        assert extras == {}
        save("g7_render_image", H=H, W=Wd, focal=focal, c2w=c2w, near=0., far=2.5, hist=hist, Nc=64, Ni=128,
             rgb=rgb, disp=disp, acc=acc)

    # ---------------- G9: gradients of the render w.r.t. rays / pose (the autograd path of
    # feature/direct_feature_matching.py:340-376: loss.backward() through render(c2w=pose_nerf)).
    # Loss = sum(rgb * G) with a fixed random G, so every ray and channel is weighted differently.
    grng = np.random.default_rng(77)
    for tag, R, Nc, Ni in (("a", 24, 8, 16), ("b", 8, 64, 128)):
        c2w = syn.orbit_pose(5, 8)[:3, :4]
        ro, rd = ray_utils.get_rays(480, 640, 585.0, t(c2w))
        sel = grng.choice(480 * 640, R, replace=False)
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).clone().requires_grad_(True)
        G = grng.standard_normal((R, 3)).astype(np.float32)
        rgb, disp, acc, _ = rendering.render(480, 640, 585.0, chunk=32768, rays=rays, near=0., far=2.5,
                                             img_idx=t(hist)[None], **kwargs_for(128, Nc, Ni))
        (rgb * t(G)).sum().backward()
        assert all(p.grad is None for p in nets[128][0].parameters())  # the coarse net gets no gradient (z_samples.detach())
        save(f"g9_render_grad_rays_{tag}", Nc=Nc, Ni=Ni, near=0., far=2.5, hist=hist, rays_o=rays[0], rays_d=rays[1],
             G=G, rgb=rgb, grad_rays_o=rays.grad[0], grad_rays_d=rays.grad[1])
        for net in nets[128][:2]:
            net.zero_grad()
    H, Wd, focal = 12, 16, 14.6
    pose = t(syn.orbit_pose(1, 8))[:3, :4].clone().requires_grad_(True)
    G = grng.standard_normal((H, Wd, 3)).astype(np.float32)
    rgb, disp, acc, _ = rendering.render(H, Wd, focal, chunk=100, c2w=pose, near=0., far=2.5, img_idx=t(hist)[None],
                                         **kwargs_for(128, 64, 128))
    (rgb * t(G)).sum().backward()
    save("g9_render_grad_c2w", H=H, W=Wd, focal=focal, c2w=pose, near=0., far=2.5, hist=hist, Nc=64, Ni=128, G=G,
         rgb=rgb, grad_c2w=pose.grad)

    # ---------------- G12: training-mode render_rays (perturb=1, test_time=False; rendering.py:276-331) and
    # ---------------- G13: one NeRF-H optimisation step (run_nerf.py:50-66: render -> NerfWLoss -> backward).
    # The reference draws t_rand = torch.rand (stratified jitter), noise = torch.randn_like (coarse alpha) and
    # u = torch.rand (importance sampling) inside render_rays; the draws are RECORDED here (wrappers around torch.rand /
    # torch.randn_like while the reference runs) and stored, so the oracle and the HIP path take them as inputs.
    from models import losses as ref_losses
    r12 = np.random.default_rng(1212)

    class Recorder:
        def __enter__(self):
            self.rec, self.o_rand, self.o_randn = [], torch.rand, torch.randn_like
            def rand(*a, **k):
                r = self.o_rand(*a, **k)
                self.rec.append(("rand", r.clone()))
                return r
            def randn_like(*a, **k):
                r = self.o_randn(*a, **k)
                self.rec.append(("randn", r.clone()))
                return r
            torch.rand, torch.randn_like = rand, randn_like
            return self

        def __exit__(self, *exc):
            torch.rand, torch.randn_like = self.o_rand, self.o_randn

    def train_kwargs(Nc, Ni, noise_std):
        kw = kwargs_for(128, Nc, Ni)
        kw.update(perturb=1., raw_noise_std=noise_std, test_time=False)
        return kw

    def digest(flat):
        flat = flat.reshape(-1)
        return flat.norm(), flat[:: max(1, flat.numel() // 256)][:256].clone()

    for tag, R, Nc, Ni, noise_std in (("a", 40, 16, 32, 0.), ("b", 24, 64, 128, 1.)):
        c2w = syn.orbit_pose(2, 8)[:3, :4]
        ro, rd = ray_utils.get_rays(480, 640, 585.0, t(c2w))
        sel = r12.choice(480 * 640, R, replace=False)
        # the rays carry gradient: the reference's training render is differentiable w.r.t. them too (G13 `g_rays`)
        rays = torch.stack([ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]], 0).clone().requires_grad_(True)
        target = t(r12.uniform(0, 1, (R, 3)).astype(np.float32))
        coarse, fine, emb_a, emb_t = nets[128]
        for m in (coarse, fine, emb_a, emb_t):
            m.zero_grad()
            m.train()
        torch.manual_seed(100 + R)
        with Recorder() as rc:
            rgb, disp, acc, extras = rendering.render(480, 640, 585.0, chunk=32768, rays=rays, near=0., far=2.5,
                                                       img_idx=t(hist)[None], retraw=True, **train_kwargs(Nc, Ni, noise_std))
        kinds = [k for k, _ in rc.rec]
        assert kinds == ["rand", "randn", "rand"], kinds   # t_rand, coarse noise, u — in this order
        t_rand, noise, u = (r for _, r in rc.rec)
        assert t_rand.shape == (R, Nc) and noise.shape == (R, Nc) and u.shape == (R, Ni)
        assert sorted(extras) == sorted(["raw", "rgb0", "disp0", "acc0", "z_std", "transient_sigmas", "beta"])
        save(f"g12_render_train_{tag}", Nc=Nc, Ni=Ni, near=0., far=2.5, hist=hist, raw_noise_std=noise_std, rays_o=rays[0].detach(),
             rays_d=rays[1].detach(), t_rand=t_rand, noise=noise, u=u, rgb=rgb.detach(), disp=disp.detach(), acc=acc.detach(),
             **{k: v.detach() for k, v in extras.items()})
        # the step: results dict as run_nerf.py:54-58, NerfWLoss(coef=1), loss = sum, backward
        loss_d = ref_losses.loss_dict['nerfw'](coef=1)({'rgb_fine': rgb, 'rgb_coarse': extras['rgb0'], 'beta': extras['beta'],
                                                        'transient_sigmas': extras['transient_sigmas']}, target)
        loss = sum(l for l in loss_d.values())
        with torch.no_grad():
            psnr = nerfw.mse2psnr(nerfw.img2mse(rgb, target))
        loss.backward()
        out = {"target": target, "psnr": psnr, "loss": loss.detach(), "g_rays": rays.grad.clone()}   # d loss / d (rays_o, rays_d) [2,R,3]
        out.update({"loss_" + k: v.detach() for k, v in loss_d.items()})
        none_grads = []
        for pre, mod in (("coarse.", coarse), ("fine.", fine), ("embedding_a.", emb_a), ("embedding_t.", emb_t)):
            for k, q in mod.named_parameters():
                if q.grad is None:
                    none_grads.append(pre + k)
                    continue
                out["gn:" + pre + k], out["gs:" + pre + k] = digest(q.grad)
        out["emb_rows"] = np.unique(hist.astype(np.int64))
        out["ga_rows"] = emb_a.weight.grad[t(out["emb_rows"])]
        out["gt_rows"] = emb_t.weight.grad[t(out["emb_rows"])]
        assert none_grads == [], none_grads   # every parameter of both nets and both tables is reached by the loss
        save(f"g13_train_step_{tag}", **out)
        for m in (coarse, fine, emb_a, emb_t):
            m.zero_grad()
            m.eval()

    # ---------------- G8: DFNet forward (reference feature/dfnet.py on a restated VGG16 stack)
    _install_vgg_stub()
    from feature import dfnet as ref_dfnet
    with torch.no_grad():
        net = ref_dfnet.DFNet()
        wts = syn.dfnet_weights(seed=3)
        load_into(net, wts)
        net.eval()
        assert list(k for k in net.state_dict() if "num_batches" not in k) == list(wts.keys())
        assert net.hypercolumn_indices == [2, 14, 28] and net.scales == [1, 4, 16]
        x = rng.uniform(0, 1, (2, 3, 32, 48)).astype(np.float32)
        f_si, p_si = net(t(x), return_feature=True, isSingleStream=False, return_pose=True, upsampleH=32, upsampleW=48)
        f_ss, p_ss = net(t(x), return_feature=True, isSingleStream=True, return_pose=False, upsampleH=40, upsampleW=56)
        _, p_only = net(t(x), return_feature=False)
        assert p_ss is None
        # channel-subsampled (every 8th of 128) + full-tensor L2 norms per level, to keep the fixture small
        nrm = lambda f: torch.sqrt((f ** 2).sum((1, 2, 3, 4)))
        save("g8_dfnet_small", x=x, cstride=8, siam_t=f_si[0][:, :, ::8], siam_r=f_si[1][:, :, ::8], pose=p_si,
             single=f_ss[0][:, :, ::8], pose_only=p_only,
             siam_t_l2=nrm(f_si[0]), siam_r_l2=nrm(f_si[1]), single_l2=nrm(f_ss[0]))
        x2 = rng.uniform(0, 1, (1, 3, 120, 160)).astype(np.float32)
        f2, _ = net(t(x2), return_feature=True, isSingleStream=True, return_pose=False, upsampleH=120, upsampleW=160)
        # 3x128x120x160 fp32 = 29 MB: store a strided subsample + per-level norms
        full = f2[0][:, 0]
        save("g8_dfnet_120x160", x=x2, sub=full[:, ::8, ::6, ::8], l2=torch.sqrt((full ** 2).sum((1, 2, 3))),
             mean=full.mean((1, 2, 3)))
        net_s = ref_dfnet.DFNet_s()
        wts_s = syn.dfnet_weights(seed=3, taps=(64,))
        load_into(net_s, wts_s)
        net_s.eval()
        fs, ps = net_s(t(x), return_feature=True, isSingleStream=True, return_pose=True, upsampleH=32, upsampleW=48)
        save("g8_dfnet_s_small", x=x, cstride=8, single=fs[0][:, :, ::8], pose=ps, single_l2=nrm(fs[0]))

    # ---------------- G10: one training step of DFNet itself (run_feature.py:166-230): train() mode and --freezeBN
    def grad_digest(net):
        out = {}
        for k, q in net.named_parameters():
            if q.grad is None:
                continue
            flat = q.grad.reshape(-1)
            out["gn:" + k] = flat.norm()
            out["gs:" + k] = flat[:: max(1, flat.numel() // 256)][:256].clone()
        return out

    # --freezeBN (utils/utils.py:18-39; that module's other imports — torchvision.utils, matplotlib — are absent here,
    # so its two helpers' effect is applied to the reference module directly): BatchNorm affine without grad, and the
    # BatchNorm modules back in eval() after net.train().
    def freeze_bn_layer(m):
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.requires_grad_(False)
                mod.bias.requires_grad_(False)
        return m

    def freeze_bn_layer_train(m):
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.eval()
        return m

    r10 = np.random.default_rng(1010)   # the cotangents are regenerated from this seed by the tests (2.9 MB each)
    xb = r10.uniform(0, 1, (4, 3, 32, 48)).astype(np.float32)
    Gt = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    Gr = r10.standard_normal((3, 2, 128, 24, 40)).astype(np.float32)
    Gp = r10.standard_normal((4, 12)).astype(np.float32)
    for mode in ("train", "freezebn"):
        net = ref_dfnet.DFNet()
        load_into(net, syn.dfnet_weights(seed=3))
        if mode == "freezebn":
            net = freeze_bn_layer(net)
        net.train()
        if mode == "freezebn":
            net = freeze_bn_layer_train(net)
        feats, pose = net(t(xb), return_feature=True, isSingleStream=False, return_pose=True, upsampleH=24, upsampleW=40)
        ((feats[0] * t(Gt)).sum() + (feats[1] * t(Gr)).sum() + (pose * t(Gp)).sum()).backward()
        bn = {}
        for i in range(3):
            m = getattr(net.adaptation_layers, "adapt_layer_%d" % i)[3]
            bn["rm%d" % i], bn["rv%d" % i] = m.running_mean.clone(), m.running_var.clone()
        save("g10_dfnet_train_" + mode, seed=1010, x=xb, Gp=Gp, Gt_l2=np.sqrt((Gt ** 2).sum()), Gr_l2=np.sqrt((Gr ** 2).sum()), cstride=8, feat_t=feats[0][:, :, ::8], feat_r=feats[1][:, :, ::8],
             pose=pose, **bn, **grad_digest(net))

    # ---------------- G11: the triplet losses of DFNet's training (feature/misc.py:355-435).  That module cannot be
    # imported here (pytorch3d, torchvision.utils, matplotlib are absent), so the three function definitions are taken
    # from its syntax tree and executed as they stand, with torch / nn as their globals.
    import ast
    src = open(os.path.join(REF, "feature", "misc.py")).read()
    wanted = ("triplet_loss", "triplet_loss_hard_negative_mining", "triplet_loss_hard_negative_mining_plus")
    mod = ast.Module(body=[n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in wanted], type_ignores=[])
    ns = {"torch": torch, "nn": torch.nn}
    exec(compile(mod, "reference:feature/misc.py", "exec"), ns)
    out, seen, seed = {}, set(), 0
    while len(seen) < 4 and seed < 200:
        r11 = np.random.default_rng(1100 + seed)
        f1 = r11.standard_normal((3, 4, 8, 5, 6)).astype(np.float32)
        f2 = (f1 * r11.uniform(0, 1) + r11.standard_normal((3, 4, 8, 5, 6)) * r11.uniform(0.05, 1.5)).astype(np.float32)
        if seed % 2:   # make a neighbouring image nearly identical: pushes the minimum to the roll cases
            f1[:, 1] = f1[:, 0] + 0.01 * f1[:, 1]
        if seed % 3 == 2:
            f2[:, 2] = f2[:, 1] + 0.01 * f2[:, 2]
        a, b = t(f1).requires_grad_(True), t(f2).requires_grad_(True)
        with torch.no_grad():
            mse = torch.nn.MSELoss()
            an, ng = torch.roll(a, 1, 1), torch.roll(b, 1, 1)
            case = int(torch.argmin(torch.stack([mse(a, ng), mse(b, an), mse(a, an), mse(b, ng)])))
        if case not in seen:
            seen.add(case)
            margin = 0.5 + 0.25 * case
            for k, fn in enumerate(wanted):
                a.grad = b.grad = None
                loss = ns[fn](a, b, margin=margin)
                loss.backward()
                out.update({f"c{case}_m{k}_loss": loss.detach(), f"c{case}_m{k}_g1": a.grad.clone(), f"c{case}_m{k}_g2": b.grad.clone()})
            out.update({f"c{case}_f1": f1, f"c{case}_f2": f2, f"c{case}_margin": margin})
        seed += 1
    assert seen == {0, 1, 2, 3}, seen
    save("g11_triplet_losses", **out)

    # ---------------- G16: the triplet-loss training step of DFNet END TO END through the reference (run_feature.py:141-162, :211-227):
    # net.train()(cat([target, rgb])) -> triplet loss of feature/misc.py (the functions executed above) on (features_rgb,
    # features_target) + a linear functional of the pose -> backward: loss, chosen case and every parameter gradient.  Six frames (three
    # per stream, so that the rolled negatives are not the positives), enlarged to the input size (level 0 = identity, as
    # run_feature.py does) and to a size that matches no level (24 x 40); train() mode and --freezeBN; all three loss functions.
    r16 = np.random.default_rng(1616)
    x16 = r16.uniform(0, 1, (6, 3, 32, 48)).astype(np.float32)
    x16[3:] = (0.7 * x16[:3] + 0.3 * x16[3:]).astype(np.float32)      # the rendered stream resembles the target stream
    x16[4] = (0.98 * x16[3] + 0.02 * x16[4]).astype(np.float32)       # two nearly identical neighbours: a roll case can win the mining
    Gp16 = r16.standard_normal((6, 12)).astype(np.float32)
    out16 = {"x": x16, "Gp": Gp16, "w_f": 0.7}
    for mode in ("train", "freezebn"):
        for (uh, uw) in ((32, 48), (24, 40)):
            for k, fn in enumerate(wanted):
                net = ref_dfnet.DFNet()
                load_into(net, syn.dfnet_weights(seed=3))
                if mode == "freezebn":
                    net = freeze_bn_layer(net)
                net.train()
                if mode == "freezebn":
                    net = freeze_bn_layer_train(net)
                feats, pose = net(t(x16), return_feature=True, isSingleStream=False, return_pose=True, upsampleH=uh, upsampleW=uw)
                f_t, f_r = feats[0], feats[1]
                margin = 0.05 + 0.05 * k          # small margins: the hinge is active on part of the rows only
                lf = ns[fn](f_r, f_t, margin=margin)
                with torch.no_grad():
                    mse = torch.nn.MSELoss()
                    an, ng = torch.roll(f_r, 1, 1), torch.roll(f_t, 1, 1)
                    cases = torch.stack([mse(f_r, ng), mse(f_t, an), mse(f_r, an), mse(f_t, ng)])
                (0.7 * lf + (pose * t(Gp16)).sum()).backward()
                tag = f"{mode}_{uh}x{uw}_m{k}"
                out16.update({f"{tag}:loss_f": lf.detach(), f"{tag}:margin": margin, f"{tag}:mse": cases, f"{tag}:pose": pose.detach()})
                out16.update({f"{tag}:{kk}": v for kk, v in grad_digest(net).items()})
    save("g16_dfnet_triplet_step", **out16)

    # ---------------- G14: the remaining render() options of the NeRF-H path (rendering.py:353-400, 269-273):
    # lindisp=True (depths linear in disparity), ndc=True (ndc_rays at near = 1, rendering.py:374-376) and c2w_staticcam
    # (rays from one pose, view directions from another, rendering.py:364-371).  white_bkgd=True is NOT a working option of
    # this path in the reference: render_rays hands it to the coarse compositor in the output_transient slot
    # (rendering.py:295), which then multiplies by transient_sigmas = None -> TypeError at test time.
    r14 = np.random.default_rng(1414)
    with torch.no_grad():
        c2w = syn.orbit_pose(4, 8)[:3, :4]
        ro, rd = ray_utils.get_rays(480, 640, 585.0, t(c2w))
        sel = r14.choice(480 * 640, 48, replace=False)
        ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
        kw = kwargs_for(128, 16, 32)
        kw.update(lindisp=True)
        rgb, disp, acc, extras = rendering.render(480, 640, 585.0, chunk=32768, rays=torch.stack([ro, rd], 0), near=0.4, far=2.5,
                                                   img_idx=t(hist)[None], retraw=True, **kw)
        save("g14_render_lindisp", Nc=16, Ni=32, near=0.4, far=2.5, hist=hist, rays_o=ro, rays_d=rd, rgb=rgb, disp=disp, acc=acc,
             raw=extras["raw"])
        try:
            kw = kwargs_for(128, 16, 32)
            kw.update(white_bkgd=True)
            rendering.render(480, 640, 585.0, chunk=32768, rays=torch.stack([ro, rd], 0), near=0.4, far=2.5, img_idx=t(hist)[None], **kw)
            white = "ok"
        except TypeError as e:
            white = "TypeError"
        assert white == "TypeError"
        H, Wd, focal = 6, 8, 7.3
        pose, pose2 = syn.orbit_pose(1, 8)[:3, :4], syn.orbit_pose(2, 8)[:3, :4]
        o0, d0 = ray_utils.get_rays(H, Wd, focal, t(pose))
        no, nd = ray_utils.ndc_rays(H, Wd, focal, 1., o0, d0)
        kw = kwargs_for(128, 16, 32)
        kw.update(ndc=True)
        rgb, disp, acc, _ = rendering.render(H, Wd, focal, chunk=100, c2w=t(pose), near=0., far=1., img_idx=t(hist)[None], **kw)
        rgb_s, disp_s, acc_s, _ = rendering.render(H, Wd, focal, chunk=100, c2w=t(pose), c2w_staticcam=t(pose2), near=0., far=2.5,
                                                   img_idx=t(hist)[None], **kwargs_for(128, 16, 32))
        save("g14_render_ndc_staticcam", H=H, W=Wd, focal=focal, c2w=pose, c2w_staticcam=pose2, hist=hist, Nc=16, Ni=32,
             ndc_rays_o=no, ndc_rays_d=nd, rgb_ndc=rgb, disp_ndc=disp, acc_ndc=acc, rgb_static=rgb_s, disp_static=disp_s,
             acc_static=acc_s, white_bkgd_raises=white)

    # ---------------- G15: the reference on TRAINED-LIKE weights.  tests/golden/trained_nerfh_weights.npz holds NeRF-H weights trained
    # natively (tools/gpu_train_scene.py: 20 000 fused HIP steps on a synthetic scene with real occupancy — three shaded spheres in
    # front of a checkered wall — held-out PSNR 28.5 dB): numbers only.  SURVEY section 7: random-init weights are contractive, trained
    # checkpoints (sharp sigma) amplify error — here the REFERENCE's own render_rays / render run on such weights: 64 rays and a
    # 12 x 16 frame at 64 + 128 samples of the training scene's cameras.
    wpath = os.path.join(HERE, "trained_nerfh_weights.npz")
    if os.path.exists(wpath):
        tw = np.load(wpath)
        cw = {k[len("coarse."):]: tw[k] for k in tw.files if k.startswith("coarse.")}
        fw = {k[len("fine."):]: tw[k] for k in tw.files if k.startswith("fine.")}
        coarse = nerfw.NeRFW("coarse", D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27)
        fine = nerfw.NeRFW("fine", D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27,
                           encode_appearance=True, encode_transient=True, in_channels_a=50, in_channels_t=20)
        load_into(coarse, cw)
        load_into(fine, fw)
        emb_a = torch.nn.Embedding(1000, 5)
        emb_t = torch.nn.Embedding(1000, 2)
        emb_a.weight.data.copy_(t(tw["embedding_a.weight"]))
        emb_t.weight.data.copy_(t(tw["embedding_t.weight"]))
        nets["trained"] = (coarse.eval(), fine.eval(), emb_a, emb_t)
        r15 = np.random.default_rng(1515)
        Ht, Wt, ft = 60, 80, 585.0 / 8          # the training cameras (tools/gpu_train_scene.py)
        with torch.no_grad():
            c2w = syn.orbit_pose(7, 16)[:3, :4]
            ro, rd = ray_utils.get_rays(Ht, Wt, ft, t(c2w))
            sel = r15.choice(Ht * Wt, 64, replace=False)
            ro, rd = ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel]
            # The reference's own intermediates of this call — the sorted z_vals (rendering.py:300-304), the coarse network's raw output
            # and compositing weights (:292-296) — are recorded by wrapping the module-level compositor for the duration of the call: the
            # staged parity test feeds exactly these fine samples to the HIP fine network, which separates arithmetic error (network +
            # compositing on the SAME samples) from the sampler's bin flips.
            taps, inner = {}, rendering.raw2outputs_NeRFW

            def tap(raw, z_vals, *a, **kw):
                out = inner(raw, z_vals, *a, **kw)
                typ = kw.get("typ", "coarse")
                taps[typ] = (raw.clone(), z_vals.clone(), out[3].clone())
                return out
            rendering.raw2outputs_NeRFW = tap
            try:
                rgb, disp, acc, extras = rendering.render(Ht, Wt, ft, chunk=32768, rays=torch.stack([ro, rd], 0), near=0., far=2.5,
                                                           img_idx=t(hist)[None], retraw=True, **kwargs_for("trained", 64, 128))
            finally:
                rendering.raw2outputs_NeRFW = inner
            assert torch.equal(taps["fine"][0], extras["raw"])
            save("g15_trained_render_rays", Nc=64, Ni=128, near=0., far=2.5, hist=hist, rays_o=ro, rays_d=rd, rgb=rgb, disp=disp, acc=acc,
                 raw=extras["raw"], z_vals=taps["fine"][1], weights=taps["fine"][2],
                 coarse_raw=taps["coarse"][0], coarse_z=taps["coarse"][1], coarse_weights=taps["coarse"][2])
            H, Wd, focal = 12, 16, 585.0 / 40   # the same field of view at a fifth of the training resolution
            c2w = syn.orbit_pose(11, 40)
            rgb, disp, acc, _ = rendering.render(H, Wd, focal, chunk=100, c2w=t(c2w)[:3, :4], near=0., far=2.5, img_idx=t(hist)[None],
                                                 **kwargs_for("trained", 64, 128))
            save("g15_trained_render_image", H=H, W=Wd, focal=focal, c2w=c2w, near=0., far=2.5, hist=hist, Nc=64, Ni=128,
                 rgb=rgb, disp=disp, acc=acc)


if __name__ == "__main__":
    main()

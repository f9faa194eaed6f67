"""SURVEY 8(f) N3, second half: render_path's per-frame back-end on the device (dfn_frame_post) against the reference's numpy
formulas (models/rendering.py:423-452, to8b of models/nerf.py:11), and BASELINE configs[2]'s workload — a 1000-pose render_path
batch at 640x480, 64+128 — at world size 1 (first / last frame against the oracle on 64 rays, every PNG written, the PNG tail
shorter than the render)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def to8b(x):   # models/nerf.py:11
    return (255 * np.clip(x, 0, 1)).astype(np.uint8)


@pytest.mark.parametrize("n,H,W", [(3, 12, 16), (2, 480, 640), (5, 7, 9)])
@pytest.mark.parametrize("single_gt", [False, True])
def test_frame_post_bit_exact_vs_numpy(n, H, W, single_gt):
    from dfnet_amd import engine as eng
    rng = np.random.default_rng(n * 1000 + H)
    rgb = rng.uniform(-0.2, 1.2, (n, H, W, 3)).astype(np.float32)
    rgb.reshape(-1)[:6] = [0.0, 1.0, 1.0 / 255, 254.999 / 255, -0.0, 0.5]   # exact bin edges
    disp = rng.uniform(1e-3, 7.0, (n, H, W)).astype(np.float32)
    gt = rng.uniform(0, 1, (H, W, 3) if single_gt else (n, H, W, 3)).astype(np.float32)
    dev = "cuda:0"
    out = eng.frame_post(torch.from_numpy(rgb).to(dev), torch.from_numpy(disp).to(dev), torch.from_numpy(gt).to(dev))
    assert np.array_equal(out["rgb8"].cpu().numpy(), to8b(rgb))
    assert np.array_equal(out["gt8"].cpu().numpy(), to8b(gt))
    for i in range(n):
        assert np.array_equal(out["disp8"][i].cpu().numpy(), to8b(disp[i] / np.max(disp[i])))   # rendering.py:450
        g = gt if single_gt else gt[i]
        ref = np.mean(np.square(rgb[i].astype(np.float64) - g.astype(np.float64)))
        ref32 = -10. * np.log10(np.mean(np.square(rgb[i] - g)))                                  # the reference's own fp32 PSNR
        got = float(out["mse"][i])
        assert abs(got - ref) <= 1e-6 * ref
        assert abs(-10. * np.log10(got) - ref32) < 1e-4
    assert np.array_equal(out["disp_max"].cpu().numpy(), disp.reshape(n, -1).max(1))
    # no ground truth: conversions only
    out2 = eng.frame_post(torch.from_numpy(rgb).to(dev), torch.from_numpy(disp).to(dev))
    assert out2["mse"] is None and out2["gt8"] is None and torch.equal(out2["rgb8"], out["rgb8"]) and torch.equal(out2["disp8"], out["disp8"])


def test_render_path_writes_reference_pngs_per_frame(tmp_path):
    """render_path with a save directory and per-frame ground truth: the PNGs hold exactly to8b of what it returns, PSNR as the
    reference prints it."""
    from PIL import Image
    from dfnet_amd import engine as eng, rendering, synthetic as syn
    from dfnet_amd.nerfw import HipQuery
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision="f16x3").load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=16, N_samples=8, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0.0, far=2.5)
    N, H, W = 19, 12, 16   # more than one back-end batch
    poses = torch.stack([torch.from_numpy(syn.orbit_pose(k, N)) for k in range(N)])
    hist = torch.from_numpy(syn.HIST_IDX).repeat(N, 1)
    gt = np.random.default_rng(0).uniform(0, 1, (N, H, W, 3)).astype(np.float32)
    rgbs, disps = rendering.render_path(None, poses, [H, W, 21.9], 32768, kw, gt_imgs=gt, savedir=str(tmp_path), img_ids=hist)
    assert rgbs.shape == (N, H, W, 3) and disps.shape == (N, H, W)
    assert sorted(os.listdir(tmp_path)) == sorted(f"{i:03d}{s}.png" for i in range(N) for s in ("", "_GT", "_disp"))
    for i in (0, 7, 16, 18):
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"{i:03d}.png")), to8b(rgbs[i]))
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"{i:03d}_GT.png")), to8b(gt[i]))
        assert np.array_equal(np.asarray(Image.open(tmp_path / f"{i:03d}_disp.png")), to8b(disps[i] / np.max(disps[i])))


def test_c3_thousand_pose_render_path_world1(tmp_path):
    """BASELINE configs[2]'s workload on one GPU: 1000 poses, 640x480, 64+128, split-f16."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_c3
    from dfnet_amd import synthetic as syn
    from oracle import nerfh_oracle as orc
    N = 1000
    rec, rgbs, disps = gpu_c3.run(N, "f16x3", str(tmp_path))
    assert rgbs.shape == (N, 480, 640, 3) and disps.shape == (N, 480, 640) and np.isfinite(rgbs).all()
    assert rec["png_files_rank0_dir"] == 3 * N
    assert rec["png_tail_s"] < rec["render_s"], rec          # the host back-end hides behind the render
    T = torch.from_numpy
    cw, fw, ea, et = syn.nerfh_weights(0)
    c, f = {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()}
    sel = torch.randperm(480 * 640, generator=torch.Generator().manual_seed(3))[:64]
    for k in (0, N - 1):
        ro, rd = orc.get_rays(480, 640, 585.0, T(syn.orbit_pose(k, N))[:3, :4])
        rows = orc.pack_ray_rows(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], 0.0, 2.5, syn.HIST_IDX)
        with torch.no_grad():
            ref = orc.render_rays(rows, c, f, T(ea), T(et), 64, 128)
        got = T(rgbs[k]).reshape(-1, 3)[sel]
        assert float((got - ref["rgb_map"]).abs().max() / ref["rgb_map"].abs().max()) < 2e-5
        gd = T(disps[k]).reshape(-1)[sel]
        assert float((gd - ref["disp_map"]).abs().max() / ref["disp_map"].abs().max()) < 2e-5
    print(rec)

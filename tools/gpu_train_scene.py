#!/usr/bin/env python3
"""Train NeRF-H natively (the fused HIP step, dfnet_amd.nerf_train) on a synthetic scene WITH REAL OCCUPANCY — three shaded spheres in
front of a checkered wall, ground truth by analytic ray casting — and save the weights as a fixture (numbers only):
tests/golden/make_golden.py then runs the REFERENCE on them (G15) and the GPU parity tests compare all three arithmetic modes against
those outputs.  SURVEY section 7 warns that random-init weights are contractive and trained checkpoints (sharp sigma) amplify error;
this is the checkpoint-like case.  The loop is /root/reference/script/run_nerf.py:32-80's (one image per step, N_rand rays without
replacement, NerfWLoss, Adam, exponential lr decay).
Usage (GPU box): python tools/gpu_train_scene.py [steps] [out.npz]     -> prints one JSON line"""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, nerf_train, synthetic as syn
from dfnet_amd.nerfw import NeRFW

T = torch.from_numpy
dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "trained_nerfh_weights.npz")
H, W, focal, near, far = 60, 80, 585.0 / 8, 0., 2.5
K_TRAIN, N_RAND, NC, NI = 40, 1536, 64, 64



def ground_truth(c2w):
    """Analytic image of the scene (dfnet_amd/synthetic.py: analytic_scene_image) at the training resolution."""
    return syn.analytic_scene_image(c2w, H, W, focal, far)


def main():
    torch.manual_seed(0)
    np.random.seed(0)
    cw, fw, ea, et = syn.nerfh_weights(0)
    coarse = NeRFW('coarse', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27)
    fine = NeRFW('fine', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True, encode_transient=True,
                 in_channels_a=50, in_channels_t=20)
    coarse.load_state_dict({k: T(v) for k, v in cw.items()})
    fine.load_state_dict({k: T(v) for k, v in fw.items()})
    emb_a, emb_t = torch.nn.Embedding(1000, 5), torch.nn.Embedding(1000, 2)
    emb_a.weight.data.copy_(T(ea))
    emb_t.weight.data.copy_(T(et))
    mods = [m.to(dev) for m in (coarse, fine, emb_a, emb_t)]
    E = eng.NerfHEngine(width=128, precision="f16x3").load_numpy(cw, fw, ea, et)
    tr = nerf_train.NerfHTrainer(E, *mods)
    opt = torch.optim.Adam(tr.params, lr=5e-4, betas=(0.9, 0.999))
    poses = [syn.orbit_pose(k, K_TRAIN)[:3, :4] for k in range(K_TRAIN)]
    imgs = [T(ground_truth(p)).to(dev) for p in poses]
    rays = [eng.raygen(H, W, focal, T(p).to(dev), want_viewdirs=False)[:2] for p in poses]
    hist = T(syn.HIST_IDX)[None].to(dev)
    t0 = time.time()
    psnr_log = []
    for step in range(steps):
        k = step % K_TRAIN
        sel = torch.from_numpy(np.random.choice(H * W, size=[N_RAND], replace=False)).to(dev)
        o, d = rays[k][0].reshape(-1, 3)[sel], rays[k][1].reshape(-1, 3)[sel]
        tgt = imgs[k].reshape(-1, 3)[sel]
        ld, psnr, _ = tr.train_step(o, d, hist, tgt, NC, NI, near, far, perturb=1., raw_noise_std=1.0)
        opt.step()
        for g in opt.param_groups:
            g['lr'] = 5e-4 * (0.1 ** (step / (steps * 1.5)))
        if step % 500 == 0 or step == steps - 1:
            psnr_log.append((step, round(float(psnr), 2)))
    torch.cuda.synchronize()
    train_s = time.time() - t0
    # held-out view with the test-time renderer
    sd = {k: p.detach().cpu().numpy() for k, p in zip(tr.names, tr.params)}
    cut = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    E.load_numpy(cut("coarse."), cut("fine."), sd["embedding_a.weight"], sd["embedding_t.weight"])
    test_pose = syn.orbit_pose(7, 16)[:3, :4]
    rgb, disp, acc = E.render_image(T(test_pose).to(dev), H, W, focal, hist[0], 64, 128, near, far, precision="f32")
    gt = T(ground_truth(test_pose)).to(dev)
    test_psnr = float(-10 * torch.log10(((rgb - gt) ** 2).mean()))
    flags = E.range_flags()
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    np.savez_compressed(out_path, **{k: v.astype(np.float32) for k, v in sd.items()},
                        meta=np.array([steps, H, W, focal, NC, NI, N_RAND, K_TRAIN, test_psnr], np.float64))
    absmax = max(float(np.abs(v).max()) for k, v in sd.items() if "embedding" not in k)
    print(json.dumps({"steps": steps, "train_s": round(train_s, 2), "ms_per_step": round(1e3 * train_s / steps, 3), "train_psnr": psnr_log,
                      "heldout_psnr_f32_64+128": round(test_psnr, 2), "acc_mean": round(float(acc.mean()), 3), "range_flags": flags,
                      "range_recoveries": tr.range_recoveries, "largest_weight": round(absmax, 3), "out": out_path}))


if __name__ == "__main__":
    main()

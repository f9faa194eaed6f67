#!/usr/bin/env python3
"""BASELINE configs[2] on the GPUs at hand: a render_path batch of N poses (default 1000 = the 7-Scenes heads test split's
length; synthetic orbit, random-weight NeRF-H) at 640x480, 64+128 samples, through dfnet_amd.rendering.render_path: frames
block-partitioned over the ranks, device back-end (dfn_frame_post), every rank writing its own PNGs, one gather at the end.

  python tools/gpu_c3.py [--frames 1000] [--precision f16x3] [--savedir DIR] [--size 640x480]
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/gpu_c3.py      (the configuration itself)

Prints one JSON line on rank 0 (frames/s, render / PNG-tail seconds, mean PSNR against a constant grey frame)."""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n_frames=1000, precision="f16x3", savedir=None, H=480, W=640, focal=585.0, Nc=64, Ni=128, keep=True):
    """Returns (record, rgbs, disps) on rank 0 ((record, None, None) elsewhere)."""
    from dfnet_amd import dist as ddist, engine as eng, rendering, synthetic as syn
    from dfnet_amd.nerfw import HipQuery
    rank, world, local = ddist.init_from_env()
    torch.cuda.set_device(local)
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision=precision).load_numpy(cw, fw, ea, et)
    kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=Ni, N_samples=Nc, use_viewdirs=True, white_bkgd=False,
              raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0.0, far=2.5)
    poses = torch.stack([torch.from_numpy(syn.orbit_pose(k, n_frames)) for k in range(n_frames)])
    hist = torch.from_numpy(syn.HIST_IDX).repeat(n_frames, 1)
    gt = np.full((H, W, 3), 0.5, np.float32)
    own_dir = savedir is None
    if own_dir:
        savedir = tempfile.mkdtemp(prefix="c3_png_")
    os.makedirs(savedir, exist_ok=True)
    t0 = time.time()
    with torch.no_grad():
        rgbs, disps = rendering.render_path(None, poses, [H, W, focal], 32768, kw, gt_imgs=gt, savedir=savedir, single_gt_img=True,
                                            img_ids=hist)
    wall = time.time() - t0
    rec = None
    if rank == 0:
        tm = rendering.render_path.last_timing
        n_png = len([f for f in os.listdir(savedir) if f.endswith(".png")]) if world == 1 else None
        rec = {"workload": f"render_path, {n_frames} poses, {W}x{H}, {Nc}+{Ni} samples, {world} GPU(s), precision {precision}",
               "frames": n_frames, "world": world, "wall_s": wall, "frames_per_s": n_frames / wall,
               "rays_per_s": n_frames * H * W / wall, "render_s": tm["render_s"], "post_launch_s": tm["post_launch_s"],
               "png_tail_s": tm["png_tail_s"], "png_files_rank0_dir": n_png,
               "mean_psnr_vs_grey": float(np.mean(-10. * np.log10(np.mean(np.square(rgbs - gt[None]), axis=(1, 2, 3)))))}
    if own_dir:
        import shutil
        shutil.rmtree(savedir, ignore_errors=True)
    if world > 1:
        torch.distributed.barrier()
    return rec, (rgbs if keep else None), (disps if keep else None)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--savedir", default=None)
    ap.add_argument("--size", default="640x480")
    a = ap.parse_args()
    Wd, Hd = (int(v) for v in a.size.split("x"))
    rec, _, _ = run(a.frames, a.precision, a.savedir, Hd, Wd, 585.0 * Wd / 640.0, keep=False)
    if rec is not None:
        print(json.dumps(rec), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()

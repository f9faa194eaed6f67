#!/usr/bin/env python3
"""bench.py — rendered rays/s of the NeRF-H hot path on MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch: the test-time render of ONE 640x480 frame
(307,200 rays, 64 coarse + 128 importance samples, NeRF-H D=8 W=128) — BASELINE.json
configs[1], synthetic random-weight scene of SURVEY.md §8(d), inputs (pose, histogram, weights)
resident in HBM before the timed region.  With N > 1 GPUs every rank renders its own block of
K frames of the render_path batch (weak scaling, no data-path collective) and the rendered
frames are gathered to rank 0 over RCCL inside the timed region.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the fine MLP, MFMA
bound): algorithmic FLOPs of its launches / their device time measured with HIP events on the
launch stream.  `cpu_baseline` times the CPU oracle (a torch-CPU port of the reference's path)
on a bounded sample of the same workload on this box's host cores (rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dfnet_amd import _lib, dist as ddist, engine as eng, synthetic as syn  # noqa: E402

H, W, FOCAL, NEAR, FAR = 480, 640, 585.0, 0.0, 2.5
NC, NI = 64, 128
MAC_COARSE, MAC_FINE = 130944, 182720  # algorithmic MAC per sample, SURVEY.md Appendix A
PEAK_TFLOPS = {"f16": 2500.0, "f32": 157.3, "f16x3": 2500.0 / 3}  # dense MFMA peaks, MI355X_MICROARCH.md (split-f16: 3 f16 MFMAs per product)


def pmc_traffic(kernel="nerfh_fine_kernel"):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary
    (profiles/*_pmc_summary.json: separate --pmc passes of this same command, tools/gpu_round.sh).
    FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream
    (MI355X_MICROARCH.md §HBM), so it is doubled.  None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")))
    if not files:
        return None
    d = json.load(open(files[-1]))
    for name, c in d.items():
        if kernel in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c and c.get("dispatches"):
            return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 / c["dispatches"]
    return None


def cpu_baseline(sample_rays, engine=None, device=None):
    """The oracle (torch CPU port of the reference path) on `sample_rays` rays of frame 0; with `engine`, the
    same rays are also rendered by the HIP path and compared (PSNR / max relative error vs the oracle)."""
    from oracle import nerfh_oracle as orc
    T = torch.from_numpy
    cw, fw, ea, et = syn.nerfh_weights(0)
    c = {k: T(v) for k, v in cw.items()}
    f = {k: T(v) for k, v in fw.items()}
    ro, rd = orc.get_rays(H, W, FOCAL, T(syn.orbit_pose(0, 8))[:3, :4])
    sel = torch.randperm(H * W, generator=torch.Generator().manual_seed(0))[:sample_rays]
    rows = orc.pack_ray_rows(ro.reshape(-1, 3)[sel], rd.reshape(-1, 3)[sel], NEAR, FAR, syn.HIST_IDX)
    with torch.no_grad():
        orc.render_rays(rows[:512], c, f, T(ea), T(et), NC, NI)  # warm-up
        t0 = time.perf_counter()
        ref = orc.render_rays(rows, c, f, T(ea), T(et), NC, NI)
        dt = time.perf_counter() - t0
    out = {"value": sample_rays / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": "port",
           "sample": f"{sample_rays} random rays of frame 0 at 64+128 samples, one chunk, {dt:.1f} s "
                     f"(oracle/nerfh_oracle.py, torch {torch.__version__} CPU fp32)"}
    if engine is not None:
        rgb, disp, acc, _ = engine.render_rays(rows[:, 0:3].to(device), rows[:, 3:6].to(device),
                                               torch.from_numpy(syn.HIST_IDX).to(device), NC, NI, NEAR, FAR)
        d = (rgb.cpu() - ref["rgb_map"]).double()
        out["parity_vs_oracle"] = {"psnr_db": float(-10 * torch.log10((d ** 2).mean().clamp_min(1e-30))),
                                   "rgb_max_rel": float(d.abs().max() / ref["rgb_map"].abs().max()),
                                   "disp_max_rel": float((disp.cpu() - ref["disp_map"]).abs().max() / ref["disp_map"].abs().max())}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--precision", default="f16", choices=["f16", "f32", "f16x3"])
    ap.add_argument("--cpu-sample", type=int, default=8192, help="rays in the CPU baseline sample (0 = skip)")
    args = ap.parse_args()

    rank, world, local = ddist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision=args.precision).load_numpy(cw, fw, ea, et)
    K, Wm = args.steps, args.warmup
    n_frames = K * world
    lo, _ = ddist.frame_block(n_frames, rank, world)
    poses = torch.stack([torch.from_numpy(syn.orbit_pose(lo + k, n_frames)) for k in range(K)]).to(dev)
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    rgbs = torch.empty(K, H, W, 3, device=dev)
    disps = torch.empty(K, H, W, device=dev)

    def step(k):
        E.render_image(poses[k % K], H, W, FOCAL, hist, NC, NI, NEAR, FAR,
                       out=(rgbs[k % K], disps[k % K], E._acc_scratch))

    E._acc_scratch = torch.empty(H, W, device=dev)
    for k in range(Wm):
        step(k)
    if world > 1:  # warm the collective too
        ddist.gather_frames(rgbs[:1], world)
    lib = _lib.load()
    torch.cuda.synchronize()
    ddist.barrier()
    lib.dfn_profile_enable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    if world > 1:
        all_rgb = ddist.gather_frames(rgbs, n_frames)
        all_disp = ddist.gather_frames(disps, n_frames)
    torch.cuda.synchronize()
    ddist.barrier()
    dt = time.perf_counter() - t0
    dt = ddist.max_over_ranks(dt, dev)

    import ctypes
    avg_ms, launches = ctypes.c_double(), ctypes.c_int()
    _lib.check(lib.dfn_profile_read(1, ctypes.byref(avg_ms), ctypes.byref(launches)), "dfn_profile_read")
    c_ms, c_l = ctypes.c_double(), ctypes.c_int()
    _lib.check(lib.dfn_profile_read(0, ctypes.byref(c_ms), ctypes.byref(c_l)), "dfn_profile_read")
    lib.dfn_profile_enable(0)

    if rank == 0:
        rays = H * W
        value = world * K * rays / dt
        fine_flops_per_launch = 2.0 * MAC_FINE * (NC + NI) * rays * K / max(launches.value, 1)
        achieved = fine_flops_per_launch / (avg_ms.value * 1e-3) / 1e12
        peak = PEAK_TFLOPS[args.precision]
        line = {
            "metric": "rendered rays/sec (64+128 samples, 640x480)",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": K, "warmup": Wm,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: synthetic random-weight NeRF-H (D=8, W=128), 640x480, "
                                   "64+128 samples, test-time render_image, 1 frame per step per GPU",
                       "rays_per_step_per_gpu": rays, "precision": {"f16": "f16 MFMA inputs / fp32 accumulate", "f32": "exact fp32 MFMA",
                                     "f16x3": "split-f16: hi/lo f16 operands, 3 f16 MFMAs per product, fp32-grade"}[args.precision],
                       "parallelism": f"frames sharded over {world} GPU(s), gather at end"},
            "roofline": {"bound": "mfma", "kernel": "nerfh_fine_kernel", "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": pmc_traffic() if args.precision == "f16" else None,
                         "traffic_note": "HBM bytes per launch, rocprofv3 PMC (2*FETCH_SIZE + WRITE_SIZE) from profiles/; "
                                         "algorithmic bytes per launch = rays * (768 z + 24 o,d + 512 ray-bias + 144 segment composites); "
                                         "raw (6912 B/ray) stays in registers since compositing is fused",
                         "launches_per_step": launches.value / K, "avg_launch_ms": avg_ms.value,
                         "algorithmic_flops_per_launch": fine_flops_per_launch,
                         "coarse_kernel_avg_launch_ms": c_ms.value,
                         "whole_path_mfma_frac": value / world * 2.0 * (MAC_COARSE * NC + MAC_FINE * (NC + NI)) / 1e12 / peak},
        }
        if world == 1 and args.cpu_sample > 0:
            line["cpu_baseline"] = cpu_baseline(args.cpu_sample, E, dev)
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

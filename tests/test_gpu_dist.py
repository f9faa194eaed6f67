"""The multi-GPU path on the one GPU a test box has: RCCL is brought up through torch.distributed's "nccl" backend
under the same launcher the driver uses, and DFN_FORCE_COLLECTIVES=1 makes dfnet_amd.dist run its gather / all-reduce / barrier in
the group of one rank — communicator initialisation, device buffers and frame order are exercised for real; only the number of
peers is not.  (1 -> 8 scaling itself is the driver's to measure.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, local = ddist.init_from_env(backend="nccl")
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl" and world == 1 and ddist.active()
dev = torch.device("cuda", local)
frames = torch.arange(5, dtype=torch.float32, device=dev)[:, None, None, None].expand(5, 6, 8, 3).contiguous() + 0.25
out = ddist.gather_frames(frames, 5)                      # a real RCCL gather (not the world-1 early-out)
assert out is not frames and out.shape == frames.shape and torch.equal(out, frames)
# the direct form of render_path / bench.py in the one-rank group (falls back to the padded gather: no peer to send to) + its record
outs, (v_rgb, v_err) = ddist.root_buffers([(6, 8, 3), ()], 5, dev)
v_rgb.copy_(frames); v_err.copy_(torch.arange(5, device=dev, dtype=torch.float32))
(d_rgb, d_err), extra = ddist.gather_frames_direct([v_rgb, v_err], 5, outs=outs, extra=torch.tensor([2.0], device=dev))
assert torch.equal(d_rgb, frames) and torch.equal(d_err, v_err) and extra.shape == (1, 1) and float(extra[0, 0]) == 2.0
assert ddist.all_gather_floats([1.0, 2.0], dev).tolist() == [[1.0, 2.0]]
p = torch.nn.Parameter(torch.ones(1 << 20, device=dev)); p.grad = torch.full_like(p, 3.0)
ddist.allreduce_gradients([p]); assert float(p.grad.mean()) == 3.0
ddist.allreduce_gradients([p], contributors=1); assert float(p.grad.mean()) == 3.0
assert ddist.max_over_ranks(1.5, dev) == 1.5
bn = torch.nn.BatchNorm2d(4).to(dev); bn.running_var.fill_(2.0); ddist.sync_buffers(bn); assert float(bn.running_var[0]) == 2.0
ddist.barrier(); torch.cuda.synchronize()
print("RCCL_WORLD1_OK", torch.cuda.nccl.version())
torch.distributed.destroy_process_group()
'''


def _launch(args, timeout=600):
    env = dict(os.environ, DFN_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                           "--master-port", "29655"] + args, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_rccl_collectives_forced_at_world1(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    r = _launch([str(script), ROOT])
    assert r.returncode == 0, r.stderr[-3000:]
    assert "RCCL_WORLD1_OK" in r.stdout


def test_bench_under_torchrun_with_rccl_gather(tmp_path):
    """bench.py exactly as the driver launches it for N > 1 (torch.distributed.run, nccl), here with one rank: the frames go
    through the RCCL gather inside the timed region."""
    r = _launch([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--no-extras"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["value"] > 1e5
    assert "forced" in line["config"]["collectives"] and line["dtype"] == "f16x3"
    # the per-rank record that will explain the N-GPU runs: render / gather seconds, package power and shader clock over the timed region
    ranks = line["ranks"]
    assert len(ranks["render_s"]["per_rank"]) == 1 and ranks["render_s"]["max"] > 0 and ranks["gather_s"]["max"] >= 0
    assert ranks["gathered_bytes"] == 0   # one rank: every frame is the root's own
    assert ranks["samples_per_rank"][0] > 0 and 100 < ranks["power_w"]["mean"] < 1600 and 300 < ranks["sclk_mhz"]["mean"] < 2600, ranks


def test_row_bands_equal_the_full_frame_bit_for_bit():
    """SURVEY 8(e)'s small-batch fallback on the device: what each rank of a world-3 / world-8 group would render for 1, 2 and 4 frames
    (dist.band_unit -> rendering.render_band), assembled the way gather_bands_direct places it, against render(c2w=...) of the whole
    frame on one rank: rgb and disparity bit-identical (rays are independent), so disp / max(disp) and the PSNR computed on the
    assembled frame are the single-GPU numbers.  The exchange itself is tests/test_host_logic.py::test_row_band_gather_gloo."""
    import torch
    from dfnet_amd import dist as ddist, engine as eng, nerfw, rendering, synthetic as syn
    dev = torch.device("cuda:0")
    E = eng.NerfHEngine().load_numpy(*syn.nerfh_weights(0))
    kw = dict(network_query_fn=nerfw.HipQuery(E, 65536), perturb=False, N_importance=32, N_samples=16, use_viewdirs=True,
              white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
    H, W, focal = 22, 32, 29.0
    hist = torch.from_numpy(syn.HIST_IDX).float().to(dev)
    for world, n_frames in ((3, 1), (3, 2), (8, 1), (8, 4), (8, 3)):
        poses = torch.stack([torch.from_numpy(syn.orbit_pose(k, 8)) for k in range(n_frames)]).to(dev)
        full = [rendering.render(H, W, focal, c2w=poses[k][:3, :4], img_idx=hist, **kw) for k in range(n_frames)]
        rgb = torch.full((n_frames, H, W, 3), float("nan"), device=dev)
        disp = torch.full((n_frames, H, W), float("nan"), device=dev)
        for rank in range(world):
            f, r0, r1 = ddist.band_unit(n_frames, H, rank, world)
            b_rgb, b_disp = rendering.render_band(H, W, focal, 32768, poses[f][:3, :4], hist, r0, r1, kw)
            assert b_rgb.shape == (r1 - r0, W, 3) and b_disp.shape == (r1 - r0, W)
            rgb[f, r0:r1], disp[f, r0:r1] = b_rgb, b_disp
        for k in range(n_frames):
            assert torch.equal(rgb[k], full[k][0]) and torch.equal(disp[k], full[k][1]), (world, n_frames, k)
    E.check_range()

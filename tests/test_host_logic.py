"""CPU tests of the host-side mirror: option tables / config files, dataset front-end on a synthetic
7-Scenes-layout tree, frame sharding and the world_size-2 gloo gather."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from dfnet_amd import datasets, dist as ddist, options, synthetic as syn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_parsers_defaults_match_reference_tables():
    n = options.nerf_parser().parse_args([])
    assert (n.netdepth, n.netwidth, n.N_samples, n.N_importance, n.chunk, n.netchunk) == (8, 128, 64, 64, 32768, 65536)
    assert (n.multires, n.multires_views, n.N_vocab, n.hist_bin, n.in_channels_a, n.in_channels_t) == (10, 4, 1000, 10, 50, 20)
    assert n.use_viewdirs is True and n.perturb == 1.0 and n.NeRFH is False and n.epochs == 600
    f = options.feature_parser().parse_args([])
    assert f.NeRFH is True and f.no_batching is True and f.epochs == 2000 and f.learning_rate == 1e-4
    assert f.combine_loss_w == [1, 1, 1] and f.patience == [200, 50] and f.tinyscale == 4.0
    d = options.dm_parser().parse_args([])
    assert d.no_grad_update is True and d.learning_rate == 1e-5 and d.feature_matching_lvl == [0, 1, 2]
    assert d.combine_loss_w == [0.5, 0.5] and d.basedir == "../logs/" and d.i_eval == 50
    assert not hasattr(d, "tinyimg") and not hasattr(n, "DFNet")


def test_config_file_then_cli_override(tmp_path):
    cfg = tmp_path / "c.txt"
    cfg.write_text("# comment\nexpname=nerfh\nbasedir = ../logs/heads\ndf=4\nNeRFH=True\nencode_hist=True\n"
                   "render_test=False\nlrate_decay=0.754  # trailing\n")
    a = options.nerf_parser().parse_args(["--config", str(cfg), "--df", "2", "--N_importance", "128"])
    assert a.expname == "nerfh" and a.basedir == "../logs/heads" and a.NeRFH and a.encode_hist
    assert a.df == 2.0 and a.render_test is False and a.lrate_decay == 0.754 and a.N_importance == 128
    assert a.config == str(cfg)
    cfg2 = tmp_path / "d.txt"
    cfg2.write_text("combine_loss_w = [0., 0., 1.]\nfeature_matching_lvl = [0]\nDFNet=True\n")
    d = options.dm_parser().parse_args(["--config", str(cfg2)])
    assert d.combine_loss_w == [0., 0., 1.] and d.feature_matching_lvl == [0] and d.DFNet


def make_scene(root, n_train=3, n_val=2, H=48, W=64, seed=0, poses=None):
    """A synthetic tree with the 7-Scenes layout under root/data/...  `poses(i)` overrides the 4x4 pose written for frame i."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    datadir = os.path.join(root, "data", "7Scenes", "heads")
    frames = os.path.join(root, "data", "deepslam_data", "7Scenes", "heads")
    os.makedirs(datadir)
    json.dump({"near": 0, "far": 2.5, "pose_scale": 1, "pose_scale2": 1, "move_all_cam_vec": [0.0, 0.0, 1.0]},
              open(os.path.join(datadir, "world_setup.json"), "w"))
    np.savetxt(os.path.join(datadir, "pose_avg_stats.txt"), np.eye(4)[:3])
    for name, seq, n in (("TrainSplit.txt", 2, n_train), ("TestSplit.txt", 1, n_val)):
        os.makedirs(os.path.join(frames, f"seq-{seq:02d}"))
        open(os.path.join(frames, name), "w").write(f"sequence{seq}\n")
        for i in range(n):
            img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(frames, f"seq-{seq:02d}", f"frame-{i:06d}.color.png"))
            np.savetxt(os.path.join(frames, f"seq-{seq:02d}", f"frame-{i:06d}.pose.txt"), syn.orbit_pose(i, 8) if poses is None else poses(i))
    return datadir


def test_seven_scenes_front_end(tmp_path):
    datadir = make_scene(str(tmp_path))
    args = options.nerf_parser().parse_args(["--datadir", datadir, "--dataset_type", "7Scenes", "--df", "2",
                                             "--load_pose_avg_stats", "--render_test"])
    train_dl, val_dl, hwf, i_split, bds, _, _ = datasets.load_7Scenes_dataloader_NeRF(args)
    assert hwf == [24, 32, 585. / 2] and list(bds) == [0, 2.5] and len(train_dl) == 3 and len(val_dl) == 2
    img, pose, hist = next(iter(train_dl))
    assert img.shape == (1, 3, 24, 32) and pose.shape == (1, 12) and hist.shape == (1, 10)
    assert 0 <= float(img.min()) and float(img.max()) <= 1 and abs(float(hist.sum()) - 100) <= 5
    assert bool((hist == hist.round()).all())
    # pose convention: identity average pose -> flip y,z columns/rows as written, then z += 1
    p = pose.reshape(3, 4).numpy()
    src = syn.orbit_pose(0, 8)
    flip = np.diag([1., -1., -1., 1.])
    want = (flip @ (src @ flip))[:3]
    want[:, 3] += [0, 0, 1]
    np.testing.assert_allclose(p, want, atol=1e-6)


def test_recentre_roundtrip():
    rng = np.random.default_rng(0)
    poses = np.stack([syn.orbit_pose(k, 8)[:3] for k in range(6)]).astype(np.float64)
    cen, avg = datasets.recentre_poses(poses)
    a4 = np.eye(4); a4[:3] = avg
    back = (a4 @ np.concatenate([cen, np.tile([[[0, 0, 0, 1.]]], (6, 1, 1))], 1))[:, :3]
    np.testing.assert_allclose(back, poses, atol=1e-12)
    np.testing.assert_allclose(avg[:, :3].T @ avg[:, :3], np.eye(3), atol=1e-12)


def test_frame_block_partition():
    for n, w in ((8, 8), (1000, 8), (7, 8), (10, 3), (5, 2), (0, 4)):
        blocks = [ddist.frame_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
        sizes = [hi - lo for lo, hi in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_band_units_cover_every_row_once():
    """SURVEY 8(e)'s small-batch fallback (fewer frames than ranks): ranks are block-partitioned over the frames, a frame's rows over
    its ranks — every row of every frame belongs to exactly one rank, no rank idles while H >= its frame's rank count, band heights
    within a frame differ by at most one row."""
    for n, w, H in ((1, 8, 480), (4, 8, 120), (3, 8, 480), (1, 2, 7), (2, 3, 5), (1, 3, 2), (7, 8, 480)):
        assert ddist.band_mode(n, w)
        seen = np.zeros((n, H), int)
        per_frame = {}
        for r in range(w):
            f, r0, r1 = ddist.band_unit(n, H, r, w)
            seen[f, r0:r1] += 1
            per_frame.setdefault(f, []).append(r1 - r0)
        assert (seen == 1).all(), (n, w, H)
        assert sorted(per_frame) == list(range(n))
        for f, hs in per_frame.items():
            assert len(hs) in (w // n, w // n + 1) and max(hs) - min(hs) <= 1
    assert not ddist.band_mode(8, 8) and not ddist.band_mode(9, 8) and not ddist.band_mode(0, 8) and not ddist.band_mode(1, 1)


_BAND_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, _ = ddist.init_from_env(backend="gloo")
n_frames, H, W = int(sys.argv[2]), int(sys.argv[3]), 5
assert ddist.active() and ddist.band_mode(n_frames, world)
def frame(f):      # what a single rank would render: every element a function of (frame, row, column, channel)
    y, x, c = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(3), indexing="ij")
    return (1000. * f + 10. * y + x + 0.25 * c).float()
f, r0, r1 = ddist.band_unit(n_frames, H, rank, world)
rgb = frame(f)[r0:r1].contiguous()
disp = frame(f)[r0:r1, :, 0].contiguous() + 0.5
full_rgb, full_disp = ddist.gather_bands_direct([rgb, disp], n_frames, H)
flags = ddist.all_gather_flags(rank, torch.device("cpu"))
assert flags == list(range(world))
if rank == 0:
    want = torch.stack([frame(k) for k in range(n_frames)])
    assert full_rgb.shape == (n_frames, H, W, 3) and torch.equal(full_rgb, want)            # bit-identical to the one-rank result
    assert full_disp.shape == (n_frames, H, W) and torch.equal(full_disp, want[..., 0] + 0.5)
    print("BANDS_OK", n_frames, H)
else:
    assert full_rgb is None and full_disp is None
ddist.barrier()
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("world,n_frames,H", [(3, 1, 11), (3, 2, 11), (2, 1, 4), (3, 1, 2)])
def test_row_band_gather_gloo(tmp_path, world, n_frames, H):
    """render_path's row-band fallback over gloo: world 3 with 1 and 2 frames (3 bands of one frame; 2 + 1 ranks on two frames),
    world 2 with one frame, and more ranks than rows (an empty band) — rank 0's assembled tensors equal the single-rank frames bit
    for bit, bands received in place, the flag all-gather behind it."""
    script = tmp_path / "b.py"
    script.write_text(_BAND_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29670 + 4 * world + n_frames + H % 3), str(script), ROOT,
                        str(n_frames), str(H)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"BANDS_OK {n_frames} {H}" in r.stdout


_GLOO_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, _ = ddist.init_from_env(backend="gloo")
n_frames = int(sys.argv[2])
lo, hi = ddist.frame_block(n_frames, rank, world)
n_loc = hi - lo
local = torch.stack([torch.full((3, 4, 3), float(i)) for i in range(lo, hi)]) if n_loc else torch.zeros(0, 3, 4, 3)
disp = torch.stack([torch.full((3, 4), 100.0 + i) for i in range(lo, hi)]) if n_loc else torch.zeros(0, 3, 4)
err = torch.tensor([1000.0 + i for i in range(lo, hi)])
out = ddist.gather_frames(local, n_frames)            # the padded dist.gather (what a forced one-rank group runs)
t = ddist.max_over_ranks(float(rank + 1), torch.device("cpu"))
assert t == float(world)
if rank == 0:
    assert out.shape == (n_frames, 3, 4, 3), out.shape
    assert all(float(out[i, 0, 0, 0]) == i and float(out[i].min()) == i for i in range(n_frames))
    print("GATHER_OK", n_frames)
else:
    assert out is None
# the direct form (render_path / bench.py): rank 0 owns the FINAL tensors, renders into its block's views and receives every peer's
# block in place: one grouped send / receive batch, no padding, no packing, plus a small per-rank record in the same batch.
# Blocks may be uneven and a rank may own NO frame (n_frames < world): it posts no frame operation, only its record.
outs, (v_rgb, v_disp, v_err) = ddist.root_buffers([(3, 4, 3), (3, 4), ()], n_frames, torch.device("cpu"))
assert v_rgb.shape[0] == n_loc
v_rgb.copy_(local); v_disp.copy_(disp); v_err.copy_(err)
(d_rgb, d_disp, d_err), extra = ddist.gather_frames_direct([v_rgb, v_disp, v_err], n_frames, outs=outs, extra=torch.tensor([float(rank) + .25]))
if rank == 0:
    assert d_rgb.data_ptr() == outs[0].data_ptr() and (n_loc == 0 or v_rgb.data_ptr() == outs[0][lo:hi].data_ptr())   # received / rendered in place
    assert torch.equal(d_rgb, out)
    assert all(float(d_disp[i].min()) == 100.0 + i == float(d_disp[i].max()) and float(d_err[i]) == 1000.0 + i for i in range(n_frames))
    assert extra.shape == (world, 1) and [float(x) for x in extra[:, 0]] == [r + .25 for r in range(world)]
    assert ddist.gathered_bytes([v_rgb, v_disp, v_err], n_frames) == (n_frames - (hi - lo)) * (36 + 12 + 1) * 4
    print("DIRECT_OK", n_frames)
else:
    assert d_rgb is None and d_disp is None and d_err is None and extra is None and outs is None
# range-guard words: every rank learns every rank's bits (bit 1 on rank 0, bit 2 on the last rank: they must OR, not max)
flags = ddist.all_gather_flags(1 if rank == 0 else (2 if rank == world - 1 else 0), torch.device("cpu"))
want = [1] + [0] * (world - 2) + [2] if world > 1 else [1]
assert flags == want, (rank, flags)
if rank == 0:
    print("FLAGS_OK", world)
ddist.barrier()
torch.distributed.destroy_process_group()
'''


def _run_gather_worker(tmp_path, world, n_frames, port):
    script = tmp_path / "w.py"
    script.write_text(_GLOO_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script), ROOT, str(n_frames)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert f"GATHER_OK {n_frames}" in r.stdout and f"DIRECT_OK {n_frames}" in r.stdout and f"FLAGS_OK {world}" in r.stdout


@pytest.mark.parametrize("n_frames", [5, 8])
def test_gather_frames_gloo_world2(tmp_path, n_frames):
    _run_gather_worker(tmp_path, 2, n_frames, 29611 + n_frames)


@pytest.mark.parametrize("n_frames", [7, 2])
def test_gather_frames_gloo_world3_uneven_and_empty_rank(tmp_path, n_frames):
    """World 3 with an uneven block split (7 = 3 + 2 + 2) and with fewer frames than ranks (2: the last rank owns nothing and
    takes part in the gather with its per-rank record only): the in-place receive of render_path's end gather."""
    _run_gather_worker(tmp_path, 3, n_frames, 29631 + n_frames)


def test_pose_error_metrics_match_scipy():
    """matrix_to_quaternion / pose_errors (feature/misc.py:49-116 without pytorch3d) against scipy's Rotation."""
    from scipy.spatial.transform import Rotation
    from dfnet_amd.feature_misc import matrix_to_quaternion, pose_errors
    rng = np.random.default_rng(0)
    R1 = Rotation.random(64, random_state=1)
    R2 = Rotation.from_rotvec(rng.normal(0, 0.2, (64, 3))) * R1
    q = matrix_to_quaternion(torch.from_numpy(R1.as_matrix())).numpy()
    ref = R1.as_quat()[:, [3, 0, 1, 2]]  # scipy: (x, y, z, w)
    assert np.allclose(np.abs((q * ref).sum(-1)), 1.0, atol=1e-6)  # same rotation up to sign
    a = np.concatenate([R1.as_matrix(), rng.normal(0, 1, (64, 3, 1))], -1)
    b = np.concatenate([R2.as_matrix(), a[:, :, 3:] + rng.normal(0, 0.1, (64, 3, 1))], -1)
    ex, eq = pose_errors(torch.from_numpy(b), torch.from_numpy(a))
    want = np.degrees((R2 * R1.inv()).magnitude())
    assert np.allclose(eq.numpy(), want, atol=2e-2)  # float32 acos near 1
    assert np.allclose(ex.numpy(), np.linalg.norm(a[:, :, 3] - b[:, :, 3], axis=-1), atol=1e-6)
    # a half-turn about each axis exercises every branch of the conversion
    for axis in np.eye(3):
        Rh = Rotation.from_rotvec(np.pi * axis).as_matrix()
        qh = matrix_to_quaternion(torch.from_numpy(Rh)).numpy()
        assert abs(qh[0]) < 1e-6 and np.allclose(np.abs(qh[1:]), axis, atol=1e-6)


_GLOO_GRAD_WORKER = r'''
import os, sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, _ = ddist.init_from_env(backend="gloo")
params = [torch.nn.Parameter(torch.zeros(3, 4)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2))]
params[0].grad = torch.full((3, 4), float(rank + 1))
params[1].grad = torch.arange(5.) * (rank + 1)
# params[2] has no gradient on any rank (e.g. the adaptation layers of the pose regressor): skipped
ddist.allreduce_gradients(params)
mean = sum(range(1, world + 1)) / world
assert torch.allclose(params[0].grad, torch.full((3, 4), mean)) and torch.allclose(params[1].grad, torch.arange(5.) * mean)
assert params[2].grad is None
if rank == 0:
    print("ALLREDUCE_OK")
ddist.barrier()
torch.distributed.destroy_process_group()
'''


def test_gradient_allreduce_gloo_world2(tmp_path):
    """Data-parallel DFNet_dm: one flat all-reduce averages the regressor's gradients over ranks."""
    script = tmp_path / "g.py"
    script.write_text(_GLOO_GRAD_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29641", str(script), ROOT],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ALLREDUCE_OK" in r.stdout


_GLOO_ROUNDS_WORKER = r'''
import sys, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, _ = ddist.init_from_env(backend="gloo")
# 5 "images" on 2 ranks (the NeRF-H training loop of script/run_nerf.py): every rank must take ceil(5 / 2) = 3 steps,
# the last one with a single contributor; BatchNorm buffers are averaged before a checkpoint
p = torch.nn.Parameter(torch.zeros(3))
steps, seen = 0, []
done = False
for epoch in range(2):
    for i, (mine, end, contrib) in enumerate(ddist.data_parallel_rounds(5, rank, world)):
        if mine:
            p.grad = torch.full((3,), float(10 * epoch + i)); done = True
        if not end:
            continue
        if not done:
            p.grad = torch.zeros(3) if p.grad is None else p.grad.zero_()
        done = False
        ddist.allreduce_gradients([p], contrib)
        seen.append(float(p.grad[0])); steps += 1
want = [0.5, 2.5, 4.0, 10.5, 12.5, 14.0]
assert steps == 6 and seen == want, (rank, steps, seen)
bn = torch.nn.BatchNorm1d(2)
bn.running_mean.fill_(float(rank)); bn.num_batches_tracked.fill_(7 + rank)
ddist.sync_buffers(bn)
assert torch.allclose(bn.running_mean, torch.full((2,), 0.5)) and int(bn.num_batches_tracked) == 7
if rank == 0: print("ROUNDS_OK")
ddist.barrier()
torch.distributed.destroy_process_group()
'''


def test_data_parallel_rounds_odd_image_count_gloo_world2(tmp_path):
    """NeRF-H training over ranks with an image count that the world size does not divide: same number of steps on every rank,
    the partial round averaged over its contributors; BatchNorm buffers synchronised before rank 0 checkpoints."""
    from dfnet_amd import dist as ddist
    assert [list(x) for x in ddist.data_parallel_rounds(3, 1, 2)] == [[False, False, 2], [True, True, 2], [False, True, 1]]
    assert [x[1] for x in ddist.data_parallel_rounds(4, 0, 1)] == [True] * 4
    script = tmp_path / "r.py"
    script.write_text(_GLOO_ROUNDS_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29643", str(script), ROOT],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "ROUNDS_OK" in r.stdout


def test_triplet_losses_and_view_perturbation(tmp_path):
    """Training-side host helpers of run_feature.py: the triplet losses against nn.TripletMarginLoss on explicitly built
    triplets (feature/misc.py:355-435), random-view pose perturbation (misc.py:437-483) and EarlyStopping."""
    from dfnet_amd import feature_misc as fm
    from dfnet_amd.callbacks import EarlyStopping
    gen = torch.Generator().manual_seed(0)
    f1, f2 = torch.randn(3, 4, 8, 5, 6, generator=gen), torch.randn(3, 4, 8, 5, 6, generator=gen)
    crit = torch.nn.TripletMarginLoss(margin=0.7, reduction='mean')
    a_neg, neg = torch.roll(f1, 1, 1), torch.roll(f2, 1, 1)
    assert torch.allclose(fm.triplet_loss(f1, f2, 0.7), crit(f1, f2, neg))
    mse = torch.nn.functional.mse_loss
    cases = [(mse(f1, neg), (f1, f2, neg)), (mse(f2, a_neg), (f2, f1, a_neg)), (mse(f1, a_neg), (f1, f2, a_neg)), (mse(f2, neg), (f2, f1, neg))]
    want = crit(*min(cases, key=lambda c: float(c[0]))[1])
    assert torch.allclose(fm.triplet_loss_hard_negative_mining_plus(f1, f2, 0.7), want)
    want2 = crit(*min(cases[:2], key=lambda c: float(c[0]))[1])
    assert torch.allclose(fm.triplet_loss_hard_negative_mining(f1, f2, 0.7), want2)
    # a positive much closer than the negative: zero loss; identical streams: the margin
    assert float(fm.triplet_loss(f1, f1 + 1e-4, 0.1)) == 0.0
    np.random.seed(3)
    pose = syn.orbit_pose(2, 8)[:3]
    new = fm.perturb_single_render_pose(pose, 0.2, 10.0)
    assert new.shape == (1, 3, 4)
    R = new[0, :, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    assert np.abs(new[0, :, 3] - pose[:, 3]).max() <= 0.2   # rotation leaves the camera where it was; +-x on top
    ang = np.degrees(np.arccos(np.clip((np.trace(R @ pose[:, :3].T) - 1) / 2, -1, 1)))
    assert 0 < ang <= 10 * np.sqrt(3) + 1e-6
    np.testing.assert_allclose(fm.perturb_rotation(pose, 0, 90, 0)[:, 3], [pose[0, 3], -pose[2, 3], pose[1, 3]], atol=1e-6)

    class A:
        val_on_psnr, basedir, model_name = False, str(tmp_path), "m"
    es = EarlyStopping(A, patience=2)
    net = torch.nn.Linear(2, 2)
    for ep, v in enumerate([1.0, 0.5, 0.6, 0.7]):
        es(v, net, epoch=ep, save_multiple=True)
    assert es.early_stop and es.val_loss_min == 0.5
    assert sorted(os.listdir(tmp_path / "m")) == ["checkpoint-0000-1.0000.pt", "checkpoint-0001-0.5000.pt"]


_GLOO_BATCH_WORKER = r'''
import importlib.util, os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from dfnet_amd import dist as ddist
rank, world, _ = ddist.init_from_env(backend="gloo")
spec = importlib.util.spec_from_file_location("run_feature", os.path.join(sys.argv[1], "script", "run_feature.py"))
rf = importlib.util.module_from_spec(spec); spec.loader.exec_module(rf)
np.random.seed(0)
mine = [list(map(int, b)) for b in rf._batches(23, 4)]      # 5 full batches -> 2 per rank, one dropped
out = [None] * world
torch.distributed.all_gather_object(out, mine)
if rank == 0:
    assert all(len(o) == 2 for o in out), out
    flat = [i for o in out for b in o for i in b]
    assert len(flat) == len(set(flat)) == 16 and all(len(b) == 4 for o in out for b in o)
    print("BATCHES_OK")
ddist.barrier()
torch.distributed.destroy_process_group()
'''


def test_feature_training_batches_gloo_world2(tmp_path):
    """Data-parallel DFNet training: every rank draws the same permutation and takes every world-th full batch."""
    script = tmp_path / "b.py"
    script.write_text(_GLOO_BATCH_WORKER)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29655", str(script), ROOT],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "BATCHES_OK" in r.stdout


def make_cambridge_scene(root, scene="ShopFacade", n_train=44, n_val=3, H=48, W=86, seed=0):
    """A synthetic tree with the Cambridge layout of the reference: <scene>/{train,test}/{rgb,poses}/, world_setup.json."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    datadir = os.path.join(root, "data", "Cambridge", scene)
    for split, n in (("train", n_train), ("test", n_val)):
        os.makedirs(os.path.join(datadir, split, "rgb"))
        os.makedirs(os.path.join(datadir, split, "poses"))
        for i in range(n):
            img = (rng.random((H, W, 3)) * 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(datadir, split, "rgb", f"{i:05d}-frame.png"))
            np.savetxt(os.path.join(datadir, split, "poses", f"{i:05d}-pose.txt"), syn.orbit_pose(i, 50))
    json.dump({"near": 0, "far": 10, "pose_scale": 0.3, "pose_scale2": 0.5, "move_all_cam_vec": [0.0, 0.0, 1.0]},
              open(os.path.join(datadir, "world_setup.json"), "w"))
    np.savetxt(os.path.join(datadir, "pose_avg_stats.txt"), np.eye(4)[:3])
    return datadir


def test_cambridge_front_end(tmp_path):
    """Cambridge Landmarks layout (cambridge_scenes.py:139-215, load_Cambridge.py fix_coord): file pairing, ShopFacade's two
    dropped training frames, trainskip, hwf from the image size and COLMAP's focal, and the axis correction written out."""
    datadir = make_cambridge_scene(str(tmp_path))
    args = options.nerf_parser().parse_args(["--datadir", datadir, "--dataset_type", "Cambridge", "--df", "2", "--trainskip", "2",
                                             "--render_test"])
    train_dl, val_dl, hwf, i_split, bds, _, _ = datasets.load_Cambridge_dataloader_NeRF(args)
    assert hwf == [24, 43, 744. / 2] and list(bds) == [0, 10]
    assert len(train_dl) == 21 and len(val_dl) == 3            # 44 - 2 dropped = 42 frames, every second one
    kept = [i for i in range(44) if i not in (35, 42)][::2]
    ds = train_dl.dataset
    assert [int(os.path.basename(f)[:5]) for f in ds.files] == kept
    img, pose, hist = ds[1] if not ds.device_prep else (None, torch.tensor(ds.poses[1], dtype=torch.float32), None)
    # the pose of kept frame 1, from its source matrix: centre on the average pose, half turn about x, R <- -R, R <- R diag(-1,1,1), rescale
    src = np.stack([syn.orbit_pose(i, 50)[:3] for i in kept + [0, 1, 2]]).astype(np.float64)
    cen, _ = datasets.recentre_poses(src)
    p4 = np.eye(4); p4[:3] = cen[1]
    rx = np.diag([1., -1., -1., 1.])
    want = (rx @ p4)[:3]
    want[:, :3] = -want[:, :3] @ np.diag([-1., 1., 1.])
    want[:, 3] = (want[:, 3] * 0.3 + [0, 0, 1.0]) * 0.5
    np.testing.assert_allclose(pose.reshape(3, 4).numpy(), want, atol=1e-6)
    R = want[:, :3]
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-6)
    f = options.feature_parser().parse_args(["--datadir", datadir, "--dataset_type", "Cambridge", "--df", "2", "--pose_only", "1"])
    tr, va, te, hwf2, _, near, far = datasets.load_Cambridge_dataloader(f)
    assert hwf2 == hwf and (near, far) == (0.0, 10.0) and tr.dataset.pose_scale == 0.3 and len(te) == 3


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_bench_entry_point_two_ranks_gloo(launcher):
    """`python bench.py --gpus 2` must start its own two ranks (the form the driver uses on an 8-GPU node), and must also
    run under an external torch.distributed.run.  --cpu-dry swaps the HIP render for a frame fill so the launch,
    frame sharding, gather, barrier / max-over-ranks timing and the JSON line are exercised here over gloo."""
    import json
    bench = os.path.join(ROOT, "bench.py")
    tail = ["--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-dry", "--backend", "gloo"]
    if launcher == "self":
        cmd = [sys.executable, bench] + tail
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29541", bench] + tail
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["frames_gathered_in_order"] is True and rec["value"] > 0
    # what will explain the 8-GPU number: per-rank render / gather seconds, clocks and power, bytes gathered; frames received in place
    ranks = rec["ranks"]
    assert rec["received_in_place"] is True and ranks["gathered_bytes"] == 3 * 6 * 8 * 4 * 4
    for k in ("render_s", "gather_s", "power_w", "sclk_mhz"):
        assert len(ranks[k]["per_rank"]) == 2
    assert ranks["render_s"]["min"] <= ranks["render_s"]["mean"] <= ranks["render_s"]["max"]
    assert ranks["power_w"]["per_rank"] == [None, None]   # no device to sample in the dry run


@pytest.mark.parametrize("tag", ["a", "b"])
def test_nerfw_loss_module_matches_reference_golden(tag):
    """dfnet_amd.losses.NerfWLoss (the host-side mirror of models/losses.py:19-57) on the REFERENCE's training-mode render
    outputs (G12) reproduces the REFERENCE's loss terms (G13), and ColorLoss / loss_dict keep the reference's surface."""
    from dfnet_amd import losses
    gd = os.path.join(ROOT, "tests", "golden")
    g12, g13 = np.load(os.path.join(gd, f"g12_render_train_{tag}.npz")), np.load(os.path.join(gd, f"g13_train_step_{tag}.npz"))
    T = torch.from_numpy
    inputs = {'rgb_fine': T(g12["rgb"]), 'rgb_coarse': T(g12["rgb0"]), 'beta': T(g12["beta"]), 'transient_sigmas': T(g12["transient_sigmas"])}
    out = losses.loss_dict['nerfw'](coef=1)(inputs, T(g13["target"]))
    assert sorted(out) == ['b_l', 'c_l', 'f_l', 's_l']
    for k, v in out.items():
        assert abs(float(v) - float(g13["loss_" + k])) <= 1e-6 * abs(float(g13["loss_" + k])) + 1e-9, k
    assert abs(float(sum(out.values())) - float(g13["loss"])) <= 1e-6 * abs(float(g13["loss"]))
    two = losses.loss_dict['nerfw'](coef=2)(inputs, T(g13["target"]))
    assert abs(float(two['f_l']) - 2 * float(out['f_l'])) < 1e-6
    no_beta = losses.NerfWLoss()({k: v for k, v in inputs.items() if k in ('rgb_fine', 'rgb_coarse')}, T(g13["target"]))
    assert sorted(no_beta) == ['c_l', 'f_l']
    col = losses.loss_dict['color'](coef=1)(inputs, T(g13["target"]))
    ref = torch.nn.functional.mse_loss(inputs['rgb_coarse'], T(g13["target"])) + torch.nn.functional.mse_loss(inputs['rgb_fine'], T(g13["target"]))
    assert abs(float(col) - float(ref)) < 1e-7


def test_render_path_png_job_numbers_by_global_frame_index(tmp_path):
    """The PNG job of one back-end batch (dfnet_amd.rendering._write_frames, run on a pool thread by every rank for its own frame
    block): file names carry the GLOBAL frame index (rendering.py:438-452), one ground-truth image may serve every frame."""
    from PIL import Image
    from dfnet_amd import rendering

    class Ev:
        def __init__(self): self.waited = False
        def synchronize(self): self.waited = True

    rng = np.random.default_rng(0)
    rgb8 = torch.from_numpy(rng.integers(0, 256, (3, 5, 7, 3), dtype=np.uint8))
    disp8 = torch.from_numpy(rng.integers(0, 256, (3, 5, 7), dtype=np.uint8))
    gt_one = torch.from_numpy(rng.integers(0, 256, (5, 7, 3), dtype=np.uint8))
    ev = Ev()
    rendering._write_frames(ev, [rgb8, disp8, gt_one], str(tmp_path), 125)
    assert ev.waited
    assert sorted(os.listdir(tmp_path)) == sorted(f"{i:03d}{s}.png" for i in (125, 126, 127) for s in ("", "_GT", "_disp"))
    assert np.array_equal(np.asarray(Image.open(tmp_path / "126.png")), rgb8[1].numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "127_disp.png")), disp8[2].numpy())
    assert np.array_equal(np.asarray(Image.open(tmp_path / "125_GT.png")), gt_one.numpy())
    # the options table carries this implementation's additions with the reference-precision defaults
    from dfnet_amd import options
    ns = options.nerf_parser().parse_known_args([])[0]
    assert ns.precision == "f16x3" and ns.coarse_precision == "same"   # the shipped default: every stage fp32-grade


def test_fused_training_tables_selfcheck():
    """The fused NeRF-H training step (csrc/nerfh_fused_api.hip) ties the master parameter tensors to the chain kernels' staging units
    and to the accumulators of the weight-gradient stream through host-built tables: every parameter element must be packed exactly
    once per pass and every gradient element written exactly once.  Host-only (no device work)."""
    import ctypes
    from dfnet_amd import _lib
    lib = _lib.load()

    class Desc(ctypes.Structure):
        _fields_ = [(k, ctypes.c_int) for k in "depth width multires multires_views hist_bin dim_a dim_t n_vocab".split()]

    for hist_bin, dim_a, dim_t in ((10, 5, 2), (8, 3, 1), (1, 1, 1)):
        d = Desc(8, 128, 10, 4, hist_bin, dim_a, dim_t, 1000)
        assert lib.dfn_nerfh_train_tables_selfcheck(ctypes.byref(d)) == 0, lib.dfn_last_error().decode()
    d = Desc(8, 64, 10, 4, 10, 5, 2, 1000)
    assert lib.dfn_nerfh_train_tables_selfcheck(ctypes.byref(d)) != 0   # netwidth 128 only


def test_roofline_table_n1_section_agrees_with_the_bench_line():
    """The generated roofline table (tools/roofline_table.py -> profiles/rNN_roofline.md) is the source of every fraction quoted in
    DESIGN.md.  Its round-5 N1 section divided per-dispatch PMC averages by the dispatch count again (8-9x too small).  Held here: for
    the newest round whose profiles carry both the N1 kernel stats and PMC, the matrix instructions issued per step x 32 768 FLOP over
    the bench line's own step time must agree with the bench line's issued-MFMA model within 15 %, and the per-kernel rows must be
    the magnitudes the counters hold (the fine forward chain issues ~9.9 M MFMAs per 1 536-ray step, not 1.1 M)."""
    import glob, importlib.util, json
    spec = importlib.util.spec_from_file_location("roofline_table", os.path.join(ROOT, "tools", "roofline_table.py"))
    rt = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(rt)
    tags = sorted({os.path.basename(f)[:3] for f in glob.glob(os.path.join(ROOT, "profiles", "r*_train_step_pmc.json"))
                   if os.path.exists(f.replace("_train_step_pmc.json", "_train_step_kernel_stats.csv"))
                   and os.path.exists(f.replace("_train_step_pmc.json", "_bench.json"))})
    assert tags
    tag = tags[-1]
    rows, steps, pmc = rt.n1_section(tag)
    n1 = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))["secondary"]["nerfh_train_step_n1"]
    m, frac_kernel_time = rt.step_fraction(rows, steps, pmc, per_dispatch=True)
    _, frac_step = rt.step_fraction(rows, steps, pmc, per_dispatch=True, step_ms=n1["step_ms"])
    R, NC, NI = 1536, 64, 128
    mac_f = R * (NC + NI) * 182720
    mac_c = R * NC * (130944 + 128 * 128 + 64 * (128 + 27) + 64 * 3)
    model = n1.get("f16_mfma_issued_frac_of_nominal", 2.0 * (9 * mac_c + 7 * mac_f) / (n1["step_ms"] * 1e-3) / 2.5e15)
    assert abs(frac_step / model - 1) < 0.15, (tag, frac_step, model)
    assert 0.8 * n1["f16_mfma_frac_of_nominal"] * 7.5 / 9 < frac_step <= 1.05 * n1["f16_mfma_frac_of_nominal"]
    fwd = next(v for k, v in pmc.items() if "train_fwd_chain_kernel<true" in k)
    assert 9e6 < rt.mfma_per_call(fwd, per_dispatch=True) < 11e6
    text = open(os.path.join(ROOT, "profiles", f"{tag}_roofline.md")).read()
    line = next(l for l in text.splitlines() if "train_fwd_chain_kernel<true" in l)
    assert 0.25 < float(line.split("|")[7]) < 0.5, line      # "of nominal" column of the fine forward chain (0.35), not 0.039

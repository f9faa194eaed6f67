"""BASELINE configs[0] plumbing on the GPU: `run_nerf.py --config config_nerfh.txt --render_test`
on a synthetic 7-Scenes-layout tree (160x120 after df=4, 32+64 samples, random-init weights, no
checkpoint), checked frame by frame against the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import nerfh_oracle as orc
from tests.test_host_logic import make_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_nerf_render_test_cli(tmp_path):
    from PIL import Image
    from dfnet_amd import datasets, options
    from dfnet_amd.nerfw import NeRFW
    datadir = make_scene(str(tmp_path), n_train=2, n_val=2, H=480, W=640)
    basedir = str(tmp_path / "logs")
    cli = ["--config", os.path.join(ROOT, "script", "config_nerfh.txt"), "--render_test", "--datadir", datadir,
           "--basedir", basedir, "--N_samples", "32", "--N_importance", "64", "--testskip", "1", "--trainskip", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_nerf.py")] + cli, cwd=os.path.join(ROOT, "script"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Mean PSNR of this run is:" in r.stdout and "Not ndc!" in r.stdout
    out = os.path.join(basedir, "nerfh")
    assert os.path.exists(os.path.join(out, "args.txt")) and os.path.exists(os.path.join(out, "config.txt"))
    # the same weights the CLI drew (global seed 0, embeddings first, each NeRFW reseeds to 0)
    torch.manual_seed(0)
    ea, et = torch.nn.Embedding(1000, 5), torch.nn.Embedding(1000, 2)
    coarse = NeRFW('coarse', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27)
    fine = NeRFW('fine', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27, encode_appearance=True,
                 encode_transient=True, in_channels_a=50, in_channels_t=20)
    c = {k: v.detach() for k, v in coarse.state_dict().items()}
    f = {k: v.detach() for k, v in fine.state_dict().items()}
    args = options.nerf_parser().parse_args(cli)
    train_dl, val_dl, hwf, _, bds, _, _ = datasets.load_7Scenes_dataloader_NeRF(args)
    assert hwf == [120, 160, 585. / 4]
    for split, dl in (("train", train_dl), ("val", val_dl)):
        d = os.path.join(out, f"evaluate_{split}_test_000000")
        for i, (img, pose, hist) in enumerate(dl):
            for suffix in ("", "_GT", "_disp"):
                assert os.path.exists(os.path.join(d, f"{i:03d}{suffix}.png"))
            if i > 0:
                continue
            c2w = torch.eye(4)
            c2w[:3, :4] = pose.reshape(3, 4)
            with torch.no_grad():
                rgb, disp, acc = orc.render(120, 160, 585. / 4, 32768, c, f, ea.weight.detach(), et.weight.detach(),
                                            32, 64, float(bds[0]), float(bds[1]), hist[0].cpu().numpy(), c2w=c2w)
            want = (255 * np.clip(rgb.numpy(), 0, 1)).astype(np.uint8)
            got = np.asarray(Image.open(os.path.join(d, "000.png")))
            assert got.shape == want.shape == (120, 160, 3)
            assert int(np.abs(got.astype(int) - want.astype(int)).max()) <= 1  # to8b truncation of a 1e-3 match
            gt = np.asarray(Image.open(os.path.join(d, "000_GT.png")))
            assert int(np.abs(gt.astype(int) - (255 * img[0].permute(1, 2, 0).cpu().numpy()).astype(np.uint8).astype(int)).max()) == 0


def test_run_nerf_reloads_reference_checkpoint(tmp_path):
    """create_nerf's reload branch (models/nerfw.py:452-472): a `000100.tar` in the reference's format (run_nerf.py:150-158
    keys) with seeded NON-default weights is found under basedir/expname, `start` becomes its global_step (output folder
    evaluate_*_000100) and frame 0 is the oracle's render with the checkpoint's weights — not the random initialisation."""
    from PIL import Image
    from dfnet_amd import datasets, options, synthetic as syn
    datadir = make_scene(str(tmp_path), n_train=1, n_val=1, H=240, W=320)
    basedir = str(tmp_path / "logs")
    os.makedirs(os.path.join(basedir, "nerfh"))
    cw, fw, ea, et = syn.nerfh_weights(5)
    T = torch.from_numpy
    torch.save({'global_step': 100,
                'network_fn_state_dict': {k: T(v) for k, v in cw.items()},
                'network_fine_state_dict': {k: T(v) for k, v in fw.items()},
                'embedding_a_state_dict': {'weight': T(ea)}, 'embedding_t_state_dict': {'weight': T(et)},
                'optimizer_state_dict': {}}, os.path.join(basedir, "nerfh", "000100.tar"))
    cli = ["--config", os.path.join(ROOT, "script", "config_nerfh.txt"), "--render_test", "--datadir", datadir,
           "--basedir", basedir, "--N_samples", "16", "--N_importance", "32", "--testskip", "1", "--trainskip", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_nerf.py")] + cli, cwd=os.path.join(ROOT, "script"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Reloading from" in r.stdout and "000100.tar" in r.stdout
    d = os.path.join(basedir, "nerfh", "evaluate_train_test_000100")
    assert os.path.isdir(d), os.listdir(os.path.join(basedir, "nerfh"))
    args = options.nerf_parser().parse_args(cli)
    train_dl, _, hwf, _, bds, _, _ = datasets.load_7Scenes_dataloader_NeRF(args)
    img, pose, hist = next(iter(train_dl))
    c2w = torch.eye(4)
    c2w[:3, :4] = pose.reshape(3, 4)
    with torch.no_grad():
        rgb, _, _ = orc.render(hwf[0], hwf[1], hwf[2], 32768, {k: T(v) for k, v in cw.items()}, {k: T(v) for k, v in fw.items()},
                               T(ea), T(et), 16, 32, float(bds[0]), float(bds[1]), hist[0].cpu().numpy(), c2w=c2w)
    want = (255 * np.clip(rgb.numpy(), 0, 1)).astype(np.uint8)
    got = np.asarray(Image.open(os.path.join(d, "000.png")))
    assert got.shape == want.shape
    assert int(np.abs(got.astype(int) - want.astype(int)).max()) <= 1


def test_run_nerf_render_test_cli_on_trained_checkpoint(tmp_path):
    """`run_nerf.py --config config_nerfh.txt --render_test` exactly as shipped (no precision flags; 64 + 128 samples) on the TRAINED fixture
    (tests/golden/trained_nerfh_weights.npz re-saved as a reference-format `020000.tar`, models/nerfw.py:452-472): the pose files are
    written so that, after the loader's axis flip and shift (dataset_loaders/seven_scenes.py), NeRF-H sees the cameras it was trained
    on.  Every written frame against the oracle's fp32 render with the same weights: 8-bit levels equal up to the to8b truncation
    (<= 1 level) at every pixel but surface-grazing ones, whose count and worst step are printed and bounded."""
    from PIL import Image
    from dfnet_amd import datasets, options, synthetic as syn
    flip = np.diag([1., -1., -1., 1.])

    def pose_file(i):   # loader: c2w = flip @ (file @ flip), then z += 1  ->  file = flip @ (want - shift) @ flip
        want = syn.orbit_pose(3 + 4 * i, 16).astype(np.float64)
        want[2, 3] -= 1.0
        return flip @ want @ flip
    datadir = make_scene(str(tmp_path), n_train=2, n_val=1, H=240, W=320, poses=pose_file)
    basedir = str(tmp_path / "logs")
    os.makedirs(os.path.join(basedir, "nerfh"))
    tw = np.load(os.path.join(ROOT, "tests", "golden", "trained_nerfh_weights.npz"))
    T = torch.from_numpy
    cw = {k[len("coarse."):]: T(tw[k]) for k in tw.files if k.startswith("coarse.")}
    fw = {k[len("fine."):]: T(tw[k]) for k in tw.files if k.startswith("fine.")}
    ea, et = T(tw["embedding_a.weight"]), T(tw["embedding_t.weight"])
    torch.save({'global_step': 20000, 'network_fn_state_dict': cw, 'network_fine_state_dict': fw,
                'embedding_a_state_dict': {'weight': ea}, 'embedding_t_state_dict': {'weight': et}, 'optimizer_state_dict': {}},
               os.path.join(basedir, "nerfh", "020000.tar"))
    cli = ["--config", os.path.join(ROOT, "script", "config_nerfh.txt"), "--render_test", "--datadir", datadir,
           "--basedir", basedir, "--testskip", "1", "--trainskip", "1", "--N_importance", "128"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_nerf.py")] + cli, cwd=os.path.join(ROOT, "script"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Reloading from" in r.stdout and "020000.tar" in r.stdout
    args = options.nerf_parser().parse_args(cli)
    assert args.precision == "f16x3" and args.coarse_precision == "same" and (args.N_samples, args.N_importance) == (64, 128)
    train_dl, val_dl, hwf, _, bds, _, _ = datasets.load_7Scenes_dataloader_NeRF(args)
    assert hwf == [60, 80, 585. / 4]
    seen = 0
    for split, dl in (("train", train_dl), ("val", val_dl)):
        d = os.path.join(basedir, "nerfh", f"evaluate_{split}_test_020000")
        for i, (img, pose, hist) in enumerate(dl):
            c2w = torch.eye(4)
            c2w[:3, :4] = pose.reshape(3, 4)
            if split == "train":
                np.testing.assert_allclose(c2w.numpy(), syn.orbit_pose(3 + 4 * i, 16), atol=1e-6)
            with torch.no_grad():
                rgb, disp, acc = orc.render(60, 80, 585. / 4, 32768, cw, fw, ea, et, 64, 128, float(bds[0]), float(bds[1]),
                                            hist[0].cpu().numpy(), c2w=c2w)
            assert float(acc.mean()) > 0.95, "the trained scene must be in view (an occupied render, not fog)"
            want = (255 * np.clip(rgb.numpy(), 0, 1)).astype(np.uint8)
            got = np.asarray(Image.open(os.path.join(d, f"{i:03d}.png")))
            assert got.shape == want.shape == (60, 80, 3)
            step = np.abs(got.astype(int) - want.astype(int))
            off = int((step > 1).sum())
            print(f"trained checkpoint through the CLI, {split} frame {i}: {off} of {step.size} 8-bit values differ by more than one level "
                  f"(worst {int(step.max())}), mean |step| {float(step.mean()):.3f}")
            assert off <= 3 and int(step.max()) <= 8      # a grazing pixel moves with fp32 round-off in the reference itself
            seen += 1
    assert seen == 3


def test_run_nerf_training_cli(tmp_path):
    """run_nerf.py WITHOUT --render_test: the NeRF-H optimisation loop (run_nerf.py:32-80,127-240) for three epochs on the
    synthetic tree — every step on the HIP training kernels — then checkpoints in the reference's format, a validation
    render with the re-packed test-time engine, and a --render_test run that reloads the newest checkpoint."""
    datadir = make_scene(str(tmp_path), n_train=3, n_val=1, H=240, W=320)
    basedir = str(tmp_path / "logs")
    cli = ["--config", os.path.join(ROOT, "script", "config_nerfh.txt"), "--datadir", datadir, "--basedir", basedir,
           "--N_samples", "16", "--N_importance", "32", "--testskip", "1", "--trainskip", "1", "--N_rand", "512", "--epochs", "2",
           "--i_weights", "1", "--i_testset", "2", "--i_print", "1", "--lrate", "1e-3"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_nerf.py")] + cli, cwd=os.path.join(ROOT, "script"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("[TRAIN] Iter")]
    assert len(lines) == 3 and all("nan" not in l for l in lines), r.stdout[-2000:]
    losses = [float(l.split("Loss: ")[1].split()[0]) for l in lines]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses   # 9 Adam steps at lr 1e-3 on 3 images
    out = os.path.join(basedir, "nerfh")
    for k in (1, 2):
        ck = torch.load(os.path.join(out, f"{k:06d}.tar"), map_location="cpu")
        assert sorted(ck) == sorted(['global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'embedding_a_state_dict',
                                     'embedding_t_state_dict', 'optimizer_state_dict'])
        assert "xyz_encoding_5.0.weight" in ck['network_fn_state_dict'] and "transient_beta.0.bias" in ck['network_fine_state_dict']
    assert os.path.exists(os.path.join(out, "testset_000002", "000.png")) and os.path.exists(os.path.join(out, "trainset_000002", "000_GT.png"))
    # the trained weights differ from the initialisation, and the validation render used them (engine re-packed after training)
    from dfnet_amd.nerfw import NeRFW
    torch.manual_seed(0)
    torch.nn.Embedding(1000, 5), torch.nn.Embedding(1000, 2)
    init = NeRFW('coarse', D=8, W=128, skips=[4], in_channels_xyz=63, in_channels_dir=27).state_dict()
    ck = torch.load(os.path.join(out, "000002.tar"), map_location="cpu")
    assert float((ck['network_fn_state_dict']["xyz_encoding_1.0.weight"] - init["xyz_encoding_1.0.weight"]).abs().max()) > 1e-4
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_nerf.py")] + cli + ["--render_test"],
                        cwd=os.path.join(ROOT, "script"), capture_output=True, text=True, timeout=900)
    assert r2.returncode == 0, r2.stderr[-3000:]
    assert "000002.tar" in r2.stdout and os.path.isdir(os.path.join(out, "evaluate_val_test_000002"))


def test_run_feature_render_feature_only_cli(tmp_path):
    """run_feature.py --render_feature_only on the synthetic tree: NeRF-H quarter-res renders + bicubic x4,
    siamese DFNet forward, feature PNGs written."""
    datadir = make_scene(str(tmp_path), n_train=2, n_val=2, H=128, W=160)
    basedir = str(tmp_path / "logs")
    cli = ["--config", os.path.join(ROOT, "script", "config_dfnet.txt"), "--render_feature_only", "--datadir", datadir,
           "--basedir", basedir, "--N_samples", "16", "--N_importance", "32", "--df", "2", "--testskip", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_feature.py")] + cli, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "render features done" in r.stdout
    from PIL import Image
    for sub in ("target", "rgb"):
        for i in range(2):
            p = tmp_path / "tmp" / "nerfh" / sub / f"{i:04d}.png"
            assert p.exists()
            a = np.asarray(Image.open(p))
            assert a.shape == (64, 80) and a.max() == 255 and a.min() == 0


def test_run_feature_eval_cli(tmp_path):
    """run_feature.py --eval: pose regression over the test split on the HIP path + the reference's error report."""
    datadir = make_scene(str(tmp_path), n_train=2, n_val=3, H=128, W=160)
    cli = ["--config", os.path.join(ROOT, "script", "config_dfnet.txt"), "--eval", "--datadir", datadir,
           "--basedir", str(tmp_path / "logs"), "--N_samples", "16", "--N_importance", "32", "--df", "2", "--testskip", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_feature.py")] + cli, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith(("Median error", "Mean error"))]
    assert len(lines) == 2 and all("degrees" in l for l in lines)
    med = float(lines[0].split("error ")[1].split("m and")[0])
    assert np.isfinite(med) and med >= 0


def test_train_dm_cli(tmp_path):
    """train.py (DFNet_dm) for one epoch on the synthetic tree: every step's forward, backward and update run, the
    loss is reported and a checkpoint with the reference's state_dict keys is written."""
    datadir = make_scene(str(tmp_path), n_train=2, n_val=2, H=128, W=160)
    basedir = str(tmp_path / "logs")
    cli = ["--config", os.path.join(ROOT, "script", "config_dfnetdm.txt"), "--datadir", datadir, "--basedir", basedir,
           "--N_samples", "16", "--N_importance", "32", "--df", "2", "--trainskip", "1", "--testskip", "1",
           "--learning_rate", "1e-6", "--i_eval", "1"]
    import torch
    from dfnet_amd import synthetic as syn
    pre = str(tmp_path / "dfnet_pretrained.pt")   # the reference requires a pretrained DFNet (train.py:108)
    ck0 = {k: torch.from_numpy(v) for k, v in syn.dfnet_weights(3).items()}
    ck0.update({f"adaptation_layers.adapt_layer_{i}.3.num_batches_tracked": torch.tensor(0) for i in range(3)})
    torch.save(ck0, pre)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "train.py")] + cli, cwd=str(tmp_path),
                       env=dict(os.environ, DFNET_DM_EPOCHS="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode != 0 and "pretrain_model_path is required" in (r.stdout + r.stderr)
    cli += ["--pretrain_model_path", pre]
    env = dict(os.environ, DFNET_DM_EPOCHS="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "train.py")] + cli, cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("At epoch")]
    assert len(line) == 1 and "train loss" in line[0] and "val loss" in line[0] and "val psnr" in line[0] and "nan" not in line[0]
    assert any("Median error" in l for l in r.stdout.splitlines())      # get_error_in_q every i_eval epochs (:469-471)
    import glob
    import torch
    cks = glob.glob(os.path.join(basedir, "dfnet_dm", "checkpoint-0000-*.pt"))   # EarlyStopping's name: checkpoint-<epoch>-<val loss>.pt
    assert len(cks) == 1, os.listdir(os.path.join(basedir, "dfnet_dm"))
    ck = torch.load(cks[0], map_location="cpu")
    assert "encoder.0.weight" in ck and "fc_pose.bias" in ck and "adaptation_layers.adapt_layer_0.3.running_var" in ck


@pytest.mark.parametrize("extra", [[], ["--freezeBN"]], ids=["bn_train", "freezeBN"])
def test_run_feature_training_cli(tmp_path, extra):
    """run_feature.py without --eval: DFNet itself trained for two epochs on the synthetic tree with the reference's
    recipe (triplet loss, random view synthesis, BatchNorm in train() mode or --freezeBN): NeRF-H renders, siamese HIP
    forward / backward of every parameter, Adam, validation, checkpoints."""
    datadir = make_scene(str(tmp_path), n_train=4, n_val=2, H=128, W=160)
    basedir = str(tmp_path / "logs")
    cli = ["--config", os.path.join(ROOT, "script", "config_dfnet.txt"), "--datadir", datadir, "--basedir", basedir,
           "--N_samples", "16", "--N_importance", "32", "--df", "2", "--trainskip", "1", "--testskip", "1",
           "--featurenet_batch_size", "2", "--learning_rate", "1e-5", "--i_eval", "1", "--rvs_refresh_rate", "2"] + extra
    env = dict(os.environ, DFNET_FEATURE_EPOCHS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_feature.py")] + cli, cwd=str(tmp_path), env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("At epoch")]
    assert len(lines) == 2 and all("nan" not in l for l in lines), r.stdout[-2000:]
    assert r.stdout.count("renders RVS...") == 1 and r.stdout.count("Median error") == 2
    import glob
    import torch
    cks = sorted(glob.glob(os.path.join(basedir, "dfnet", "checkpoint-*.pt")))
    assert cks, os.listdir(os.path.join(basedir, "dfnet"))
    ck = torch.load(cks[0], map_location="cpu")
    assert "encoder.0.weight" in ck and "adaptation_layers.adapt_layer_2.3.running_mean" in ck
    moved = float((ck["adaptation_layers.adapt_layer_0.3.running_mean"]).abs().max()) > 0
    assert moved == (not extra)   # train() mode moves the running statistics, --freezeBN leaves them


def test_run_feature_eval_cambridge_cli(tmp_path):
    """The Cambridge Landmarks front-end (the reference's default config_dfnet.txt scene) through run_feature.py --eval."""
    from tests.test_host_logic import make_cambridge_scene
    datadir = make_cambridge_scene(str(tmp_path), scene="KingsCollege", n_train=3, n_val=3, H=128, W=228)
    cli = ["--config", os.path.join(ROOT, "script", "config_dfnet.txt"), "--eval", "--dataset_type", "Cambridge", "--datadir", datadir,
           "--basedir", str(tmp_path / "logs"), "--N_samples", "16", "--N_importance", "32", "--df", "2", "--testskip", "1"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "script", "run_feature.py")] + cli, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith(("Median error", "Mean error"))]
    assert len(lines) == 2 and all("degrees" in l for l in lines)

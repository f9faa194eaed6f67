"""Dataset front-end producing the hot path's inputs: per frame `(img [1,3,H,W] in [0,1],
pose [1,12], hist [1,hist_bin])`, plus `hwf` and `[near, far]`.

File I/O and PNG decoding are host work.  The per-frame ARITHMETIC — INTER_AREA downscale, luma,
histogram, percentages — runs on the GPU when one is present (`dfn_frame_prep`: the decoded 8-bit
frame goes to HBM once, 3 bytes per source pixel, and the items come out as device tensors); without a GPU (the CPU
test suite) the same arithmetic is the numpy / torch restatement below.  Restated compactly so the
drop-in CLIs run: the 7-Scenes on-disk layout (`TrainSplit.txt`, `seq-XX/frame-XXXXXX.color.png`,
`.pose.txt`; /root/reference/dataset_loaders/seven_scenes.py:185-354), the pose re-centring /
axis flip / scene rescale of load_7Scenes.py:279-344 (`fix_coord`, including its `M·([R|T]·M)`
product as written) and the 10-bin luma histogram index of seven_scenes.py:346-352.

Cambridge Landmarks (`dataset_type=Cambridge`, the scene of the reference's default configs): the layout of
cambridge_scenes.py:139-215 (`<datadir>/{train,test}/{rgb,poses}/`, focal 744 px) and the axis correction of
load_Cambridge.py's fix_coord.

Path convention: `--datadir ../data/7Scenes/<scene>` holds world_setup.json and
pose_avg_stats.txt; frames live under `<datadir>/../../deepslam_data/7Scenes/<scene>` (the
reference hard-codes `../data/deepslam_data/7Scenes`, which is the same place for the standard tree).
"""
import json
import os
import os.path as osp

import numpy as np
import torch


def _load_png(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.


def _area_downscale(img, H, W):
    """cv2.INTER_AREA: exact box mean for integer factors, PIL BOX otherwise."""
    h, w = img.shape[:2]
    if h % H == 0 and w % W == 0:
        return img.reshape(H, h // H, W, w // W, 3).mean((1, 3), dtype=np.float32)
    from PIL import Image
    chans = [np.asarray(Image.fromarray(img[..., c]).resize((W, H), Image.BOX)) for c in range(3)]
    return np.stack(chans, -1).astype(np.float32)


def luma_histogram(img_chw, bins=10):
    """Histogram index vector fed to the NeRF-H embeddings: Y = .299R+.587G+.114B, torch.histc over
    [0,1], converted to integer percentages (seven_scenes.py:346-352)."""
    y = 0.299 * img_chw[0] + 0.587 * img_chw[1] + 0.114 * img_chw[2]
    h = torch.histc(y, bins=bins, min=0., max=1.)
    return torch.round(h / h.sum() * 100)


def recentre_poses(poses, pose_avg=None):
    """[N,3,4] camera-to-world -> centred by the inverse average pose (load_7Scenes.py:143-197)."""
    if pose_avg is None:
        centre = poses[..., 3].mean(0)
        z = poses[..., 2].mean(0)
        z = z / np.linalg.norm(z)
        x = np.cross(poses[..., 1].mean(0), z)
        x = x / np.linalg.norm(x)
        pose_avg = np.stack([x, np.cross(z, x), z, centre], 1)
    avg = np.eye(4)
    avg[:3] = pose_avg
    homo = np.concatenate([poses, np.tile(np.array([[[0, 0, 0, 1.]]]), (len(poses), 1, 1))], 1)
    return (np.linalg.inv(avg) @ homo)[:, :3], pose_avg


def to_nerf_frame(poses, setup, dataset="7Scenes"):
    """Axis correction to the LLFF convention and scene rescale.  7-Scenes (load_7Scenes.py:311-338): M ([R|T] M) with
    M = diag(1, -1, -1, 1), as written there.  Cambridge (load_Cambridge.py: fix_coord): a half turn about x applied to
    the whole pose, then R <- -R, then R <- R diag(-1, 1, 1)."""
    homo = np.concatenate([poses, np.tile(np.array([[[0, 0, 0, 1.]]]), (len(poses), 1, 1))], 1)
    if dataset == "Cambridge":
        half_turn_x = np.array([[1, 0, 0, 0], [0, np.cos(np.pi), -np.sin(np.pi), 0], [0, np.sin(np.pi), np.cos(np.pi), 0], [0, 0, 0, 1.]])
        out = half_turn_x @ homo
        out[:, :3, :3] = -out[:, :3, :3]
        out[:, :3, :3] = out[:, :3, :3] @ np.diag([-1., 1., 1.])
        out = out[:, :3, :4]
    else:
        flip = np.diag([1., -1., -1., 1.])
        out = (flip @ (homo @ flip))[:, :3, :4]
    out[:, :3, 3] *= setup["pose_scale"]
    if list(setup["move_all_cam_vec"]) != [0., 0., 0.]:
        out[:, :3, 3] += np.asarray(setup["move_all_cam_vec"])
    if setup["pose_scale2"] != 1.0:
        out[:, :3, 3] *= setup["pose_scale2"]
    return out


class _Frames(torch.utils.data.Dataset):
    """Items are (img [3,H,W] in [0,1], pose [12], hist [bins]); subclasses fill files / poses / H / W / focal."""

    def _finish(self, df, focal, hist_bin, device_prep):
        h, w = _load_png(self.files[0]).shape[:2]
        self.df = df
        self.H, self.W, self.focal = int(h // df), int(w // df), focal / df
        self.hist_bin = hist_bin
        # frames are prepared where they are consumed: on the GPU whenever there is one
        self.device_prep = torch.cuda.is_available() if device_prep is None else bool(device_prep)

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        if self.device_prep:
            from PIL import Image
            from .engine import frame_prep
            raw = torch.from_numpy(np.asarray(Image.open(self.files[i]).convert("RGB"), dtype=np.uint8).copy())
            img, hist = frame_prep(raw.to(torch.device("cuda", torch.cuda.current_device())), self.H, self.W, self.hist_bin)
            return img, torch.tensor(self.poses[i], dtype=torch.float32), hist
        img = _load_png(self.files[i])
        if self.df != 1.:
            img = _area_downscale(img, self.H, self.W)
        img = torch.from_numpy(np.ascontiguousarray(img.transpose(2, 0, 1)))
        return img, torch.tensor(self.poses[i], dtype=torch.float32), luma_histogram(img, self.hist_bin)


class CambridgeFrames(_Frames):
    """One split of a Cambridge Landmarks scene as the reference lays it out (cambridge_scenes.py:139-215):
    <datadir>/{train,test}/rgb/* and .../poses/* (one 4x4 text matrix per frame), matched by sorted file name; every
    `skip`-th frame; focal 744 px at full resolution (COLMAP, :115); ShopFacade's two abnormal training frames (indices
    42 and 35) dropped as the reference does."""

    def __init__(self, datadir, train, skip=1, df=1., focal=744., hist_bin=10, device_prep=None):
        root = osp.join(datadir, 'train' if train else 'test')
        rgb = sorted(osp.join(root, 'rgb', f) for f in os.listdir(osp.join(root, 'rgb')))
        pos = sorted(osp.join(root, 'poses', f) for f in os.listdir(osp.join(root, 'poses')))
        if osp.basename(osp.normpath(datadir)) == 'ShopFacade' and train:
            for k in (42, 35):
                del rgb[k], pos[k]
        if len(rgb) != len(pos):
            raise Exception('RGB file count does not match pose file count!')
        self.gt_idx = np.arange(len(rgb))[::max(int(skip), 1)]
        self.files = [rgb[i] for i in self.gt_idx]
        self.poses = np.asarray([np.loadtxt(pos[i])[:3, :4].reshape(12) for i in self.gt_idx], dtype=np.float64).reshape(-1, 12)
        self._finish(df, focal, hist_bin, device_prep)


class SevenScenesFrames(_Frames):
    """One split of a 7-Scenes scene; items are (img [3,H,W], pose [12], hist [bins])."""

    def __init__(self, frames_root, train, skip=1, df=1., focal=585., hist_bin=10, device_prep=None):
        split = osp.join(frames_root, 'TrainSplit.txt' if train else 'TestSplit.txt')
        with open(split) as fh:
            seqs = [int(l.split('sequence')[-1]) for l in fh if l.strip() and not l.startswith('#')]
        self.files, poses = [], []
        for seq in seqs:
            d = osp.join(frames_root, 'seq-{:02d}'.format(seq))
            ids = sorted(int(n[6:12]) for n in os.listdir(d) if 'pose' in n)[::max(int(skip), 1)]
            for i in ids:
                self.files.append(osp.join(d, 'frame-{:06d}.color.png'.format(i)))
                poses.append(np.loadtxt(osp.join(d, 'frame-{:06d}.pose.txt'.format(i))).flatten()[:12])
        self.poses = np.asarray(poses, dtype=np.float64).reshape(-1, 12)
        self._finish(df, focal, hist_bin, device_prep)


def _setup_of(datadir):
    with open(osp.join(datadir, 'world_setup.json')) as fh:
        return json.load(fh)


def load_Cambridge_dataloader_NeRF(args):
    """(train_dl, val_dl, hwf, i_split, bounds, render_poses, render_img) as load_Cambridge.py:420-476."""
    datadir = osp.normpath(args.datadir)
    setup = _setup_of(datadir)
    kw = dict(df=args.df, hist_bin=args.hist_bin)
    train_set = CambridgeFrames(datadir, True, args.trainskip, **kw)
    val_set = CambridgeFrames(datadir, False, args.testskip, **kw)
    n_train = len(train_set)
    allp = np.concatenate([train_set.poses, val_set.poses]).reshape(-1, 3, 4)
    avg = np.loadtxt(osp.join(datadir, 'pose_avg_stats.txt')) if args.load_pose_avg_stats else None
    allp, _ = recentre_poses(allp, avg)
    allp = to_nerf_frame(allp, setup, "Cambridge").reshape(-1, 12)
    train_set.poses, val_set.poses = allp[:n_train], allp[n_train:]
    shuffle = not (args.render_video_train or args.render_test)
    train_dl = torch.utils.data.DataLoader(train_set, batch_size=1, shuffle=shuffle)
    val_dl = torch.utils.data.DataLoader(val_set, batch_size=1, shuffle=False)
    hwf = [train_set.H, train_set.W, train_set.focal]
    idx = [train_set.gt_idx, val_set.gt_idx, val_set.gt_idx]
    return train_dl, val_dl, hwf, idx, np.array([setup["near"], setup["far"]]), None, None


def load_Cambridge_dataloader(args):
    """(train_dl, val_dl, test_dl, hwf, i_split, near, far) for the pose-regression / feature CLIs
    (load_Cambridge.py:349-418): axes corrected, NOT rescaled (fix_coord_supp applies the scene rescale later)."""
    if not args.pose_only:
        raise Exception('load_Cambridge_dataloader() currently only support PoseNet Training, not NeRF training')
    datadir = osp.normpath(args.datadir)
    setup = _setup_of(datadir)
    kw = dict(df=args.df, hist_bin=args.hist_bin)
    train_set = CambridgeFrames(datadir, not args.finetune_unlabel, args.trainskip, **kw)
    val_set = CambridgeFrames(datadir, False, args.testskip, **kw)
    n_train = len(train_set)
    allp = np.concatenate([train_set.poses, val_set.poses]).reshape(-1, 3, 4)
    avg = np.loadtxt(osp.join(datadir, 'pose_avg_stats.txt')) if args.load_pose_avg_stats else None
    allp, _ = recentre_poses(allp, avg)
    unit = dict(pose_scale=1, pose_scale2=1.0, move_all_cam_vec=[0., 0., 0.])
    allp = to_nerf_frame(allp, unit, "Cambridge").reshape(-1, 12)
    train_set.poses, val_set.poses = allp[:n_train], allp[n_train:]
    for ds in (train_set, val_set):
        ds.pose_scale, ds.pose_scale2, ds.move_all_cam_vec = setup["pose_scale"], setup["pose_scale2"], setup["move_all_cam_vec"]
        ds.near, ds.far = setup["near"], setup["far"]
    train_dl = torch.utils.data.DataLoader(train_set, batch_size=args.batch_size, shuffle=not args.eval)
    val_dl = torch.utils.data.DataLoader(val_set, batch_size=args.val_batch_size, shuffle=False)
    test_dl = torch.utils.data.DataLoader(val_set, batch_size=1, shuffle=False)
    hwf = [train_set.H, train_set.W, train_set.focal]
    idx = [train_set.gt_idx, val_set.gt_idx, val_set.gt_idx]
    return train_dl, val_dl, test_dl, hwf, idx, float(min(setup["near"], setup["far"])), float(max(setup["near"], setup["far"]))


def load_7Scenes_dataloader_NeRF(args):
    """(train_dl, val_dl, hwf, i_split, bounds, render_poses, render_img) as load_7Scenes.py:497-555."""
    datadir = osp.normpath(args.datadir)
    scene = osp.basename(datadir)
    dataset = osp.basename(osp.dirname(datadir))
    frames_root = osp.join(osp.dirname(osp.dirname(datadir)), 'deepslam_data', dataset, scene)
    with open(osp.join(datadir, 'world_setup.json')) as fh:
        setup = json.load(fh)
    kw = dict(df=args.df, hist_bin=args.hist_bin)
    train_set = SevenScenesFrames(frames_root, True, args.trainskip, **kw)
    val_set = SevenScenesFrames(frames_root, False, args.testskip, **kw)
    n_train = len(train_set)
    allp = np.concatenate([train_set.poses, val_set.poses]).reshape(-1, 3, 4)
    avg = np.loadtxt(osp.join(datadir, 'pose_avg_stats.txt')) if args.load_pose_avg_stats else None
    allp, _ = recentre_poses(allp, avg)
    allp = to_nerf_frame(allp, setup).reshape(-1, 12)
    train_set.poses, val_set.poses = allp[:n_train], allp[n_train:]
    shuffle = not (args.render_video_train or args.render_test)
    train_dl = torch.utils.data.DataLoader(train_set, batch_size=1, shuffle=shuffle)
    val_dl = torch.utils.data.DataLoader(val_set, batch_size=1, shuffle=False)
    hwf = [train_set.H, train_set.W, train_set.focal]
    idx = np.arange(n_train), np.arange(len(val_set)), np.arange(len(val_set))
    return train_dl, val_dl, hwf, list(idx), np.array([setup["near"], setup["far"]]), None, None


def load_7Scenes_dataloader(args):
    """(train_dl, val_dl, test_dl, hwf, i_split, near, far) for the pose-regression / feature CLIs
    (load_7Scenes.py:422-495): poses re-centred and axis-flipped but NOT rescaled (the scene rescale is applied
    later by fix_coord_supp); the datasets carry pose_scale / pose_scale2 / move_all_cam_vec."""
    if not args.pose_only:
        raise Exception('load_7Scenes_dataloader() currently only support PoseNet Training, not NeRF training')
    datadir = osp.normpath(args.datadir)
    scene, dataset = osp.basename(datadir), osp.basename(osp.dirname(datadir))
    frames_root = osp.join(osp.dirname(osp.dirname(datadir)), 'deepslam_data', dataset, scene)
    with open(osp.join(datadir, 'world_setup.json')) as fh:
        setup = json.load(fh)
    kw = dict(df=args.df, hist_bin=args.hist_bin)
    train_set = SevenScenesFrames(frames_root, not args.finetune_unlabel, args.trainskip, **kw)
    val_set = SevenScenesFrames(frames_root, False, args.testskip, **kw)
    n_train = len(train_set)
    allp = np.concatenate([train_set.poses, val_set.poses]).reshape(-1, 3, 4)
    avg = np.loadtxt(osp.join(datadir, 'pose_avg_stats.txt')) if args.load_pose_avg_stats else None
    allp, _ = recentre_poses(allp, avg)
    unit = dict(pose_scale=1, pose_scale2=1.0, move_all_cam_vec=[0., 0., 0.])
    allp = to_nerf_frame(allp, unit).reshape(-1, 12)
    train_set.poses, val_set.poses = allp[:n_train], allp[n_train:]
    for ds in (train_set, val_set):
        ds.pose_scale, ds.pose_scale2, ds.move_all_cam_vec = setup["pose_scale"], setup["pose_scale2"], setup["move_all_cam_vec"]
        ds.near, ds.far = setup["near"], setup["far"]
    train_dl = torch.utils.data.DataLoader(train_set, batch_size=args.batch_size, shuffle=True)
    val_dl = torch.utils.data.DataLoader(val_set, batch_size=args.val_batch_size, shuffle=False)
    test_dl = torch.utils.data.DataLoader(val_set, batch_size=1, shuffle=False)
    hwf = [train_set.H, train_set.W, train_set.focal]
    idx = [np.arange(n_train), np.arange(len(val_set)), np.arange(len(val_set))]
    return train_dl, val_dl, test_dl, hwf, idx, float(min(setup["near"], setup["far"])), float(max(setup["near"], setup["far"]))

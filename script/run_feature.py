#!/usr/bin/env python3
"""Drop-in for the reference's script/run_feature.py on the MI355X hot path: NeRF-H renders of the
dataset poses (render_nerfw_imgs) fed to the DFNet feature extractor.

    python run_feature.py --config config_dfnet.txt --render_feature_only

Native here: everything `--render_feature_only` executes (run_feature.py:313-346) — render every test
frame with NeRF-H (quarter resolution + bicubic x4 with --tinyimg), run the siamese DFNet forward on
[target, render] and save one feature channel of each stream as PNG under ./tmp/<expname>/{target,rgb}/.
`--eval` prints the median / mean pose error of the regressor over the test split (HIP forward + the quaternion
error of feature/misc.py:49-131).  Without either flag DFNet itself is trained (run_feature.py:100-422): NeRF-H
renders of the training poses, then per epoch Adam steps on pose loss + feature loss (MSE or the triplet loss with
in-triplet hard negatives) [+ the pose loss on randomly synthesised views, --random_view_synthesis], validation,
ReduceLROnPlateau, early stopping with checkpoints.  Every forward and every parameter gradient of DFNet — encoder,
adaptation layers, BatchNorm on batch statistics or frozen (--freezeBN), pose head — runs on the HIP path
(dfn_dfnet_forward_train / dfn_dfnet_backward_all_params); the losses on the feature stacks and the optimizer are
torch tensor ops.  DFNET_FEATURE_EPOCHS caps the epoch count (default: --epochs + 1 like the reference).
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd.datasets import load_7Scenes_dataloader, load_Cambridge_dataloader  # noqa: E402
from dfnet_amd.dfnet import DFNet, DFNet_s  # noqa: E402
from dfnet_amd import dist as ddist  # noqa: E402
from dfnet_amd import optim  # noqa: E402
from dfnet_amd.callbacks import EarlyStopping  # noqa: E402
from dfnet_amd.feature_misc import (PoseLoss, freeze_bn_layer, freeze_bn_layer_train, get_error_in_q,  # noqa: E402
                                    perturb_single_render_pose, render_nerfw_imgs, render_virtual_imgs,
                                    triplet_loss_hard_negative_mining_plus)
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import feature_parser  # noqa: E402
from dfnet_amd.rendering import _write_png  # noqa: E402


def _save_channel(t, path):
    """One feature channel [H,W] -> 8-bit PNG, min-max normalised (the reference renders it through a
    matplotlib colour map; the values are the same, the palette is not reproduced)."""
    a = t.detach().float().cpu().numpy()
    a = (a - a.min()) / max(float(a.max() - a.min()), 1e-12)
    _write_png(path, (a * 255 + 0.5).clip(0, 255).astype(np.uint8))


def _batches(dset_size, batch_size):
    """The reference's batching (run_feature.py:108-123): a random permutation cut into FULL batches; a trailing
    partial batch is dropped."""
    select_inds = np.random.choice(dset_size, size=[dset_size], replace=False)
    starts = list(range(0, dset_size - batch_size + 1, batch_size))
    rank, world = ddist.rank_world()
    if world > 1:   # data parallel: batches dealt round-robin to the ranks (same seed on every rank), equal step counts
        starts = starts[:len(starts) // world * world][rank::world]
    for i_batch in starts:
        yield select_inds[i_batch:i_batch + batch_size]


def _step(feat_model, optimizer, loss):
    """backward, gradient all-reduce, update.  Returns the loss as a 0-dim DEVICE tensor: the reference's `loss.item()` per step
    (run_feature.py:160, :226) makes the host wait for the device and then enqueue the next step's ~300 launches with the GPU idle
    (1.5 ms of a 14.5 ms step); the epoch functions read the losses once, at the end (_mean_loss)."""
    loss.backward()
    ddist.allreduce_gradients(list(feat_model.parameters()))   # no-op on one GPU; one flat RCCL all-reduce otherwise
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach().reshape(1)


def _mean_loss(losses):
    """np.mean over the reference's per-step floats, from the 0-dim device tensors of an epoch (one device wait)."""
    if not losses:
        return float("nan")
    return float(np.mean(torch.cat(losses).cpu().numpy().astype(np.float64)))


def _siamese_forward(args, feat_model, target_in, rgb_in, H, W):
    features, predict_pose = feat_model(torch.cat([target_in, rgb_in]), True, upsampleH=H, upsampleW=W)
    return features[0], features[1], predict_pose   # [L,B,128,H,W] target / render stacks, [2B,12]


def _feature_loss(args, features_rgb, features_target, FeatureLoss):
    if args.tripletloss:
        return triplet_loss_hard_negative_mining_plus(features_rgb, features_target, margin=args.triplet_margin)
    return FeatureLoss(features_rgb, features_target)


def train_on_batch(args, targets, rgbs, poses, feat_model, dset_size, FeatureLoss, optimizer, hwf, device):
    """One epoch over the rendered training set (run_feature.py:100-164)."""
    feat_model.train()
    H, W = int(hwf[0]), int(hwf[1])
    if args.freezeBN:
        feat_model = freeze_bn_layer_train(feat_model)
    losses = []
    batch_size = args.featurenet_batch_size
    for i_inds in _batches(dset_size, batch_size):
        target_in = targets[i_inds].permute(0, 3, 1, 2).to(device)
        rgb_in = rgbs[i_inds].permute(0, 3, 1, 2).to(device)
        pose = poses[i_inds].reshape(batch_size, 12).to(device)
        pose = torch.cat([pose, pose])
        features_target, features_rgb, predict_pose = _siamese_forward(args, feat_model, target_in, rgb_in, H, W)
        if args.poselossonly:
            loss = PoseLoss(args, predict_pose, pose, device)
        elif args.featurelossonly:
            loss = FeatureLoss(features_rgb, features_target)
        else:
            loss = PoseLoss(args, predict_pose, pose, device) + _feature_loss(args, features_rgb, features_target, FeatureLoss)
        losses.append(_step(feat_model, optimizer, loss))
    return _mean_loss(losses)


def train_on_batch_with_random_view_synthesis(args, targets, rgbs, poses, virtue_view, poses_perturb, feat_model, dset_size,
                                              FeatureLoss, optimizer, hwf, device):
    """One epoch with random view synthesis (run_feature.py:166-230): the siamese step on [target, render] plus the pose
    loss of the regressor on a NeRF-H render at a perturbed pose."""
    feat_model.train()
    H, W = int(hwf[0]), int(hwf[1])
    if args.freezeBN:
        feat_model = freeze_bn_layer_train(feat_model)
    losses = []
    batch_size = args.featurenet_batch_size
    for i_inds in _batches(dset_size, batch_size):
        target_in = targets[i_inds].permute(0, 3, 1, 2).to(device)
        rgb_in = rgbs[i_inds].permute(0, 3, 1, 2).to(device)
        pose = poses[i_inds].reshape(batch_size, 12).to(device)
        rgb_perturb = virtue_view[i_inds].permute(0, 3, 1, 2).to(device)
        pose_perturb = poses_perturb[i_inds].reshape(batch_size, 12).to(device)
        pose = torch.cat([pose, pose])
        if getattr(feat_model, "pyramid_features", False):
            # the synthesised views' pose forward (the reference's second call, run_feature.py:219: `feat_model(rgb_perturb, False)`) rides in
            # the siamese forward's encoder pass: one batch of 3 x batch_size frames, the leading 2 x batch_size being the siamese pair —
            # the encoder has no batch-coupled layer, BatchNorm and the feature loss see the pair only, so every number is the two-pass one
            features, pose_all = feat_model(torch.cat([target_in, rgb_in, rgb_perturb]), True, upsampleH=H, upsampleW=W,
                                            feature_images=2 * batch_size)
            features_target, features_rgb = features[0], features[1]
            predict_pose, virtue_pose = pose_all[:2 * batch_size], pose_all[2 * batch_size:]
        else:
            features_target, features_rgb, predict_pose = _siamese_forward(args, feat_model, target_in, rgb_in, H, W)
            virtue_pose = None
        loss_pose = PoseLoss(args, predict_pose, pose, device)
        loss_f = _feature_loss(args, features_rgb, features_target, FeatureLoss)
        if virtue_pose is None:
            _, virtue_pose = feat_model(rgb_perturb, False)
        loss_pose_perturb = PoseLoss(args, virtue_pose, pose_perturb, device)
        loss = args.combine_loss_w[0] * loss_pose + args.combine_loss_w[1] * loss_f + args.combine_loss_w[2] * loss_pose_perturb
        losses.append(_step(feat_model, optimizer, loss))
    return _mean_loss(losses)


def _synthesise_views(args, poses, img_idxs, hwf, device, render_kwargs_test, world_setup_dict):
    """Perturbed poses clipped to the training poses' bounding box + d_max, and their NeRF-H renders
    (run_feature.py:358-380)."""
    dset_size = poses.shape[0]
    b_min = [float(poses[:, j, 3].min()) - args.d_max for j in range(3)]
    b_max = [float(poses[:, j, 3].max()) + args.d_max for j in range(3)]
    poses_perturb = poses.clone().numpy()
    for i in range(dset_size):
        poses_perturb[i] = perturb_single_render_pose(poses_perturb[i], args.rvs_trans, args.rvs_rotation)
        for j in range(3):
            poses_perturb[i, j, 3] = min(max(poses_perturb[i, j, 3], b_min[j]), b_max[j])
    poses_perturb = torch.Tensor(poses_perturb).to(device)
    print("renders RVS...")
    virtue_view = render_virtual_imgs(args, poses_perturb, img_idxs, hwf, device, render_kwargs_test, world_setup_dict)
    return virtue_view, poses_perturb


def train_feature(args, train_dl, val_dl, test_dl, hwf, i_split, near, far):
    device = torch.device("cuda", torch.cuda.current_device())
    feat_model = DFNet_s() if args.DFNet_s else DFNet()
    if args.pretrain_model_path != '':
        print("load posenet from ", args.pretrain_model_path)
        feat_model.load_state_dict(torch.load(args.pretrain_model_path, map_location="cpu"))
    feat_model.eval()
    H, W, focal = hwf
    H, W = int(H), int(W)
    hwf = [H, W, focal]
    os.makedirs(os.path.join(args.basedir, args.expname), exist_ok=True)
    with open(os.path.join(args.basedir, args.expname, 'args.txt'), 'w') as fh:
        for arg in sorted(vars(args)):
            fh.write('{} = {}\n'.format(arg, getattr(args, arg)))
    _, render_kwargs_test, start, _, _ = create_nerf(args)
    render_kwargs_test.update({'near': near, 'far': far})
    world_setup_dict = {k: getattr(train_dl.dataset, k) for k in ('pose_scale', 'pose_scale2', 'move_all_cam_vec')}
    if args.render_feature_only:
        targets, rgbs, poses, img_idxs = render_nerfw_imgs(args, test_dl, hwf, device, render_kwargs_test, world_setup_dict)
        out_t = os.path.join('.', 'tmp', args.expname, 'target')
        out_r = os.path.join('.', 'tmp', args.expname, 'rgb')
        os.makedirs(out_t, exist_ok=True)
        os.makedirs(out_r, exist_ok=True)
        save_i = 2  # feature channel saved, out of 128 (run_feature.py:327)
        with torch.no_grad():
            for i in range(poses.shape[0]):
                x = torch.cat([targets[i:i + 1].permute(0, 3, 1, 2), rgbs[i:i + 1].permute(0, 3, 1, 2)]).to(device)
                features, _ = feat_model(x, True, upsampleH=H, upsampleW=W)
                features_target, features_rgb = features[0], features[1]   # each [L, 1, 128, H, W]
                _save_channel(features_target[0, 0, save_i], os.path.join(out_t, '%04d.png' % i))
                _save_channel(features_rgb[0, 0, save_i], os.path.join(out_r, '%04d.png' % i))
        print("render features done")
        return
    if args.eval:  # run_feature.py:306-311: pose error of the regressor over the test split
        get_error_in_q(args, test_dl, feat_model, len(val_dl.dataset), device, batch_size=1)
        return
    # ---- training (run_feature.py:232-422)
    if args.freezeBN:
        feat_model = freeze_bn_layer(feat_model)
    feat_model.to(device)
    # --tripletloss: the siamese training forward keeps its feature stacks as a low-resolution pyramid and the triplet loss is taken
    # from there (dfnet.FeaturePyramid, csrc/dfnet_triplet_pyr.hip): the same loss and gradients without the two [3, B, 128, H, W]
    # stacks, their gradients, the upsample and its adjoint.  Any loss that needs real stacks (MSE FeatureLoss) keeps the tensors.
    feat_model.pyramid_features = bool(args.tripletloss) and not args.featurelossonly and os.environ.get("DFNET_PYRAMID_TRIPLET", "1") != "0"
    optimizer = optim.Adam(feat_model.parameters(), lr=args.learning_rate)   # torch.optim.Adam, one launch per step (dfnet_amd/optim.py)
    scheduler = torch.optim.lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.95, patience=args.patience[1])
    early_stopping = EarlyStopping(args, patience=args.patience[0], verbose=False)
    loss_func = torch.nn.MSELoss(reduction='mean')
    targets, rgbs, poses, img_idxs = render_nerfw_imgs(args, train_dl, hwf, device, render_kwargs_test, world_setup_dict)
    dset_size = len(train_dl.dataset)
    n_epoch = int(os.environ.get("DFNET_FEATURE_EPOCHS", args.epochs + 1))
    virtue_view = poses_perturb = None
    for epoch in range(n_epoch):
        if epoch:
            feat_model.recommit()   # fresh split-f16 weight scales for the trained weights (dfnet.py: recommit)
        if args.random_view_synthesis:
            if epoch % args.rvs_refresh_rate == 0:
                virtue_view, poses_perturb = _synthesise_views(args, poses, img_idxs, hwf, device, render_kwargs_test,
                                                               world_setup_dict)
            train_loss = train_on_batch_with_random_view_synthesis(args, targets, rgbs, poses, virtue_view, poses_perturb,
                                                                   feat_model, dset_size, loss_func, optimizer, hwf, device)
        else:
            train_loss = train_on_batch(args, targets, rgbs, poses, feat_model, dset_size, loss_func, optimizer, hwf, device)
        feat_model.eval()
        val_losses = []
        with torch.no_grad():
            for data, pose, _ in val_dl:
                _, predict = feat_model(data.to(device))
                val_losses.append(loss_func(predict, pose.to(device)).item())
        val_loss = float(np.mean(val_losses))
        scheduler.step(val_loss)
        print('At epoch {0:6d} : train loss: {1:.4f}, val loss: {2:.4f}'.format(epoch, train_loss, val_loss))
        early_stopping(val_loss, feat_model, epoch=epoch, save_multiple=(not args.no_save_multiple), save_all=args.save_all_ckpt)
        if early_stopping.early_stop:
            print("Early stopping")
            break
        if args.featurelossonly:
            continue
        if epoch % args.i_eval == 0:
            get_error_in_q(args, test_dl, feat_model, len(test_dl.dataset), device, batch_size=1)


def main(argv=None):
    np.random.seed(0)
    torch.manual_seed(0)
    args = feature_parser().parse_args(argv)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    ddist.init_from_env()   # torchrun: one process per GPU, gradients averaged over RCCL; single process otherwise
    if args.dataset_type not in ('7Scenes', 'Cambridge'):
        raise NotImplementedError(f"dataset_type={args.dataset_type}: the 7Scenes and Cambridge front-ends are built")
    load_dataloader = load_7Scenes_dataloader if args.dataset_type == '7Scenes' else load_Cambridge_dataloader
    train_dl, val_dl, test_dl, hwf, i_split, near, far = load_dataloader(args)
    train_feature(args, train_dl, val_dl, test_dl, hwf, i_split, near, far)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Drop-in surface of the reference's script/train.py (DFNet_dm: direct feature matching) on the MI355X path.

    python train.py --config config_dfnetdm.txt --eval

Native: the whole step — DFNet pose regression, NeRF-H render at the predicted pose (quarter resolution + bicubic
x4), siamese DFNet features, cosine feature-matching loss — and its backward: HIP gradient kernels for the feature
extractor's input, the bicubic resize, the render (down to the pose) and the pose regressor's own conv / fc
weights; Adam (torch.optim over the module's parameters) applies the update.  `--eval` prints the median / mean
pose error over the test split (as the reference) and the mean losses / PSNR over the validation split.
Training follows the reference's loop (train epoch, validation pass, EarlyStopping with its checkpoint naming, pose error every
i_eval epochs); the TensorBoard writer is not mirrored.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd import dist as ddist  # noqa: E402
from dfnet_amd import optim  # noqa: E402
from dfnet_amd.datasets import load_7Scenes_dataloader, load_Cambridge_dataloader  # noqa: E402
from dfnet_amd.dfnet import DFNet, DFNet_s  # noqa: E402
from dfnet_amd.callbacks import EarlyStopping  # noqa: E402
from dfnet_amd.direct_feature_matching import matching_step_forward, train_feature_matching  # noqa: E402
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import dm_parser  # noqa: E402


def main(argv=None):
    np.random.seed(0)
    torch.manual_seed(0)
    args = dm_parser().parse_args(argv)
    rank, world, local = ddist.init_from_env()   # torchrun: one process per GPU, gradients averaged over RCCL
    torch.cuda.set_device(local)
    device = torch.device("cuda", torch.cuda.current_device())
    if args.dataset_type not in ('7Scenes', 'Cambridge'):
        raise NotImplementedError(f"dataset_type={args.dataset_type}: the 7Scenes and Cambridge front-ends are built")
    load_dataloader = load_7Scenes_dataloader if args.dataset_type == '7Scenes' else load_Cambridge_dataloader
    args.pose_only = 1  # the reference passes the PoseNet loader (dm/prepare_data.py)
    train_dl, val_dl, test_dl, hwf, i_split, near, far = load_dataloader(args)
    Net = DFNet_s if args.DFNet_s else DFNet
    model, feat_model = Net().eval(), Net().eval()
    # train.py:108-121 of the reference: a pretrained DFNet is REQUIRED for the pose estimator; the feature extractor uses
    # --pretrain_featurenet_path, or the same DFNet when that is empty.  (Random features would train silently.)
    if not args.pretrain_model_path:
        raise SystemExit("train.py: --pretrain_model_path is required (a DFNet checkpoint from run_feature.py)")
    model.load_state_dict(torch.load(args.pretrain_model_path, map_location="cpu"))
    if args.pretrain_featurenet_path:
        feat_model.load_state_dict(torch.load(args.pretrain_featurenet_path, map_location="cpu"))
    else:
        print('Use the same DFNet for Feature Extraction and Pose Regression')
        feat_model.load_state_dict(torch.load(args.pretrain_model_path, map_location="cpu"))
    if not args.eval:
        # train.py:122-136 of the reference: Adam over the pose regressor, the EarlyStopping callback, train_feature_matching
        # (direct_feature_matching.py:412-471: train epoch, validation pass, early stopping / checkpoint-<epoch>-<val>.pt, pose
        # error every i_eval epochs); every gradient on the HIP path, the optimizer step by torch
        model.to(device)
        optimizer = optim.Adam(model.parameters(), lr=args.learning_rate)   # torch.optim.Adam, one launch per step (dfnet_amd/optim.py)
        early_stopping = EarlyStopping(args, patience=args.patience[0], verbose=False)
        n_epoch = int(os.environ.get("DFNET_DM_EPOCHS", 2001))   # the reference hard-codes 2001 (:437) and relies on early stopping
        train_feature_matching(args, model, feat_model, optimizer, i_split, hwf, near, far, device, early_stopping, train_dl=train_dl,
                               val_dl=val_dl, test_dl=test_dl, n_epoch=n_epoch)
        return
    _, render_kwargs_test, start, _, _ = create_nerf(args)
    render_kwargs_test.update({'near': near, 'far': far})
    setup = {k: getattr(train_dl.dataset, k) for k in ('pose_scale', 'pose_scale2', 'move_all_cam_vec')}
    # train.py:138-157: `--eval` = pose error of the DFNet_dm regressor over the test split ...
    from dfnet_amd.feature_misc import get_error_in_q
    print(len(test_dl.dataset))
    get_error_in_q(args, test_dl, model, len(test_dl.dataset), device, batch_size=1)
    # ... and, beyond the reference, the feature-matching losses of the forward step over the validation split
    stats = []
    for data, pose, img_idx in val_dl:
        out = matching_step_forward(args, data, model, feat_model, pose, img_idx, hwf, True, device, setup,
                                    **render_kwargs_test)
        stats.append([float(out[k]) for k in ("loss", "feat_loss", "photo_loss", "pose_loss", "psnr")])
    m = np.mean(stats, 0)
    print('DFNet_dm forward over {} val batches: loss {:.6f} feat {:.6f} photo {:.6f} pose {:.6f} psnr {:.3f}'.format(
        len(stats), *m))


if __name__ == '__main__':
    main()

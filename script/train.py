#!/usr/bin/env python3
"""Drop-in surface of the reference's script/train.py (DFNet_dm: direct feature matching) on the MI355X path.

    python train.py --config config_dfnetdm.txt --eval

Native: the forward half of every step — DFNet pose regression, NeRF-H render at the predicted pose (quarter
resolution + bicubic x4), siamese DFNet features, cosine feature-matching loss — evaluated over the validation
split (`--eval` prints the mean losses / PSNR).  The update itself (loss.backward(), Adam on the pose network)
needs gradient kernels for the render and the conv stack and is not built: without --eval this script stops
with a clear message.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd.datasets import load_7Scenes_dataloader  # noqa: E402
from dfnet_amd.dfnet import DFNet, DFNet_s  # noqa: E402
from dfnet_amd.direct_feature_matching import matching_step_forward  # noqa: E402
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import dm_parser  # noqa: E402


def main(argv=None):
    np.random.seed(0)
    torch.manual_seed(0)
    args = dm_parser().parse_args(argv)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    device = torch.device("cuda", torch.cuda.current_device())
    if args.dataset_type != '7Scenes':
        raise NotImplementedError(f"dataset_type={args.dataset_type}: only the 7Scenes front-end is built")
    args.pose_only = 1  # the reference passes the PoseNet loader (dm/prepare_data.py)
    train_dl, val_dl, test_dl, hwf, i_split, near, far = load_7Scenes_dataloader(args)
    Net = DFNet_s if args.DFNet_s else DFNet
    model, feat_model = Net().eval(), Net().eval()
    if args.pretrain_model_path:
        model.load_state_dict(torch.load(args.pretrain_model_path, map_location="cpu"))
    if args.pretrain_featurenet_path:
        feat_model.load_state_dict(torch.load(args.pretrain_featurenet_path, map_location="cpu"))
    _, render_kwargs_test, start, _, _ = create_nerf(args)
    render_kwargs_test.update({'near': near, 'far': far})
    setup = {k: getattr(train_dl.dataset, k) for k in ('pose_scale', 'pose_scale2', 'move_all_cam_vec')}
    if not args.eval:
        raise NotImplementedError("DFNet_dm optimisation needs backward kernels (render + conv stack) that are not "
                                  "built yet; run with --eval for the forward feature-matching losses")
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

#!/usr/bin/env python3
"""Drop-in for the reference's script/run_feature.py on the MI355X hot path: NeRF-H renders of the
dataset poses (render_nerfw_imgs) fed to the DFNet feature extractor.

    python run_feature.py --config config_dfnet.txt --render_feature_only

Native here: everything `--render_feature_only` executes (run_feature.py:313-346) — render every test
frame with NeRF-H (quarter resolution + bicubic x4 with --tinyimg), run the siamese DFNet forward on
[target, render] and save one feature channel of each stream as PNG under ./tmp/<expname>/{target,rgb}/.
`--eval` prints the median / mean pose error of the regressor over the test split (HIP forward + the quaternion
error of feature/misc.py:49-131).  The optimisation loop (run_feature.py:349-422: Adam, triplet loss, random view
synthesis) needs weight gradients and stops with a clear message.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd.datasets import load_7Scenes_dataloader  # noqa: E402
from dfnet_amd.dfnet import DFNet, DFNet_s  # noqa: E402
from dfnet_amd.feature_misc import render_nerfw_imgs  # noqa: E402
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import feature_parser  # noqa: E402
from dfnet_amd.rendering import _write_png  # noqa: E402


def _save_channel(t, path):
    """One feature channel [H,W] -> 8-bit PNG, min-max normalised (the reference renders it through a
    matplotlib colour map; the values are the same, the palette is not reproduced)."""
    a = t.detach().float().cpu().numpy()
    a = (a - a.min()) / max(float(a.max() - a.min()), 1e-12)
    _write_png(path, (a * 255 + 0.5).clip(0, 255).astype(np.uint8))


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
        from dfnet_amd.feature_misc import get_error_in_q
        get_error_in_q(args, test_dl, feat_model, len(val_dl.dataset), device, batch_size=1)
        return
    raise NotImplementedError("DFNet optimisation (triplet loss, random view synthesis: weight gradients of the conv "
                              "stack) is not built; use --eval / --render_feature_only, or train with the reference "
                              "and load the checkpoint via --pretrain_model_path")


def main(argv=None):
    np.random.seed(0)
    torch.manual_seed(0)
    args = feature_parser().parse_args(argv)
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    if args.dataset_type != '7Scenes':
        raise NotImplementedError(f"dataset_type={args.dataset_type}: only the 7Scenes front-end is built")
    train_dl, val_dl, test_dl, hwf, i_split, near, far = load_7Scenes_dataloader(args)
    train_feature(args, train_dl, val_dl, test_dl, hwf, i_split, near, far)


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Drop-in for the reference's script/run_nerf.py on the MI355X render path.

    python run_nerf.py --config config_nerfh.txt --render_test

Same flags and config files (dfnet_amd/options.py), same output tree
(`<basedir>/<expname>/evaluate_{train,val}_test_<step>/NNN.png`, `args.txt`, `config.txt`).
Single GPU: run as is.  Multi-GPU (frames of render_path sharded over ranks, one RCCL gather):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_nerf.py ...
The optimisation loop of the reference (run_nerf.py:32-80,127-240) is not part of the hot path and
is not implemented: without --render_test this script stops with a clear message.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd import dist as ddist  # noqa: E402
from dfnet_amd.datasets import load_7Scenes_dataloader_NeRF, load_Cambridge_dataloader_NeRF  # noqa: E402
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import config_parser  # noqa: E402
from dfnet_amd.rendering import render_test  # noqa: E402


def train_nerf(args, train_dl, val_dl, hwf, i_split, near, far, render_poses=None, render_img=None):
    basedir, expname = args.basedir, args.expname
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    os.makedirs(os.path.join(basedir, expname), exist_ok=True)
    if rank == 0:
        with open(os.path.join(basedir, expname, 'args.txt'), 'w') as fh:
            for arg in sorted(vars(args)):
                fh.write('{} = {}\n'.format(arg, getattr(args, arg)))
        if args.config is not None:
            with open(os.path.join(basedir, expname, 'config.txt'), 'w') as fh:
                fh.write(open(args.config, 'r').read())
    render_kwargs_train, render_kwargs_test, start, grad_vars, optimizer = create_nerf(args)
    bds = {'near': near, 'far': far}
    render_kwargs_train.update(bds)
    render_kwargs_test.update(bds)
    if args.render_test:
        print('TRAIN views are', i_split[0])
        print('VAL views are', i_split[1])
        render_test(args, train_dl, val_dl, hwf, start, render_kwargs_test)
        return
    raise NotImplementedError(
        "NeRF-H optimisation (run_nerf.py without --render_test) is outside the render hot path and not "
        "implemented here; train with the reference and render with this tool (checkpoints load unchanged).")


def main(argv=None):
    np.random.seed(0)
    torch.manual_seed(0)
    args = config_parser().parse_args(argv)
    rank, world, local = ddist.init_from_env()
    torch.cuda.set_device(local)
    if args.dataset_type not in ('7Scenes', 'Cambridge'):
        raise NotImplementedError(f"dataset_type={args.dataset_type}: the 7Scenes and Cambridge front-ends are built")
    loader = load_7Scenes_dataloader_NeRF if args.dataset_type == '7Scenes' else load_Cambridge_dataloader_NeRF
    train_dl, val_dl, hwf, i_split, bds, render_poses, render_img = loader(args)
    near, far = float(bds[0]), float(bds[1])
    print('NEAR FAR', near, far)
    train_nerf(args, train_dl, val_dl, hwf, i_split, near, far, render_poses, render_img)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()

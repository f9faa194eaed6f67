#!/usr/bin/env python3
"""Drop-in for the reference's script/run_nerf.py on the MI355X path.

    python run_nerf.py --config config_nerfh.txt [--render_test]

Same flags and config files (dfnet_amd/options.py), same output tree (`<basedir>/<expname>/{:06d}.tar` checkpoints in the
reference's format, `evaluate_{train,val}_test_<step>/NNN.png`, `trainset_/testset_<epoch>` renders, `args.txt`, `config.txt`).
Without --render_test it trains NeRF-H as run_nerf.py:32-80,127-240 does: per training image N_rand random rays ->
render(**render_kwargs_train) -> NerfWLoss -> backward -> Adam -> exponential lr decay, every network product and every
gradient on the HIP training kernels (dfnet_amd/nerf_train.py); validation renders use the packed test-time engine.
Single GPU: run as is.  Multi-GPU (one process per GPU; training images of an epoch dealt round-robin to the ranks with
one flat gradient all-reduce per step over RCCL, frames of render_path sharded with one gather):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 run_nerf.py ...
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from dfnet_amd import dist as ddist  # noqa: E402
from dfnet_amd.datasets import load_7Scenes_dataloader_NeRF, load_Cambridge_dataloader_NeRF  # noqa: E402
from dfnet_amd.losses import loss_dict  # noqa: E402
from dfnet_amd.nerfw import create_nerf  # noqa: E402
from dfnet_amd.options import config_parser  # noqa: E402
from dfnet_amd.ray_utils import get_rays  # noqa: E402
from dfnet_amd.rendering import render_path, render_test  # noqa: E402


def train_on_epoch_nerfw(args, train_dl, H, W, focal, N_rand, optimizer, loss_func, global_step, render_kwargs_train):
    """run_nerf.py:32-80.  One step per training image: N_rand rays without replacement (np.random.choice, as the reference),
    the fused HIP step (forward, NerfWLoss, every gradient), Adam, lr decay.  Returns the last (loss, psnr)."""
    device = torch.device("cuda", torch.cuda.current_device())
    trainer = render_kwargs_train['network_query_fn'].trainer
    rank, world = ddist.rank_world()
    loss = psnr = None
    # Images are dealt to the ranks in rounds of `world`; every rank takes ONE optimisation step per round, also in the last,
    # partial round of an epoch (ranks without an image contribute zero gradients): dist.data_parallel_rounds
    plan = ddist.data_parallel_rounds(len(train_dl), rank, world)
    stepped_in_round = False
    for (target, pose, img_idx), (mine, end_of_round, contributors) in zip(train_dl, plan):
        select_inds = np.random.choice(H * W, size=[N_rand], replace=False) if N_rand is not None else np.arange(H * W)   # every rank draws: generators stay in step
        if mine:
            target = target[0].permute(1, 2, 0).to(device)
            pose = pose.reshape(3, 4).to(device)
            rays_o, rays_d = get_rays(H, W, focal, pose)
            sel = torch.from_numpy(select_inds).to(device)
            rays_o, rays_d = rays_o.reshape(-1, 3)[sel], rays_d.reshape(-1, 3)[sel]
            target_s = target.reshape(-1, 3)[sel]
            loss_d, psnr, _ = trainer.train_step(rays_o, rays_d, img_idx.to(device), target_s, args.N_samples, args.N_importance,
                                                 render_kwargs_train['near'], render_kwargs_train['far'], perturb=float(args.perturb),
                                                 raw_noise_std=float(args.raw_noise_std), coef=loss_func.coef, lambda_u=loss_func.lambda_u)
            loss = sum(loss_d.values())
            stepped_in_round = True
        if not end_of_round:
            continue
        if not stepped_in_round:   # no image for this rank in the (partial) round: zero gradients into the all-reduce
            for q in trainer.params:
                q.grad = torch.zeros_like(q) if q.grad is None else q.grad.zero_()
        stepped_in_round = False
        ddist.allreduce_gradients(trainer.params, contributors)
        optimizer.step()
        render_kwargs_train['network_query_fn'].stale = True
        # NOTE: IMPORTANT!  update learning rate (run_nerf.py:70-76)
        decay_rate = 0.1
        decay_steps = args.lrate_decay * 1000
        new_lrate = args.lrate * (decay_rate ** (global_step / decay_steps))
        for param_group in optimizer.param_groups:
            param_group['lr'] = new_lrate
    trainer.flush_range_check()   # range-flag reads still in flight (NerfHTrainer.range_check = "skip"): act on them before a checkpoint / render
    return loss, psnr


def _holdout(dl, device, skip=1):
    imgs, poses, idxs = [], [], []
    for batch_idx, (img, pose, img_idx) in enumerate(dl):
        if batch_idx % skip != 0:
            continue
        imgs.append(img.permute(0, 2, 3, 1))
        p = torch.zeros(1, 4, 4)
        p[0, :3, :4] = pose.reshape(3, 4)[:3, :4]
        p[0, 3, 3] = 1.
        poses.append(p)
        idxs.append(img_idx)
    return torch.cat(imgs, 0).cpu().numpy(), torch.cat(poses, 0).to(device), torch.cat(idxs, 0).to(device)


def train_nerf(args, train_dl, val_dl, hwf, i_split, near, far, render_poses=None, render_img=None):
    H, W, focal = hwf
    H, W = int(H), int(W)
    hwf = [H, W, focal]
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
    global_step = start
    bds = {'near': near, 'far': far}
    render_kwargs_train.update(bds)
    render_kwargs_test.update(bds)
    if args.render_test:
        print('TRAIN views are', i_split[0])
        print('VAL views are', i_split[1])
        render_test(args, train_dl, val_dl, hwf, start, render_kwargs_test)
        return
    if optimizer is None:
        raise ValueError("--no_grad_update without --render_test: nothing to do")
    device = torch.device("cuda", torch.cuda.current_device())
    N_rand = args.N_rand
    N_epoch = args.epochs + 1
    print('Begin')
    print('TRAIN views are', i_split[0])
    print('VAL views are', i_split[1])
    loss_func = loss_dict['nerfw'](coef=1)
    for i in range(start, N_epoch):
        time0 = time.time()
        loss, psnr = train_on_epoch_nerfw(args, train_dl, H, W, focal, N_rand, optimizer, loss_func, global_step, render_kwargs_train)
        dt = time.time() - time0
        if i % args.i_weights == 0 and i != 0 and rank == 0:
            path = os.path.join(basedir, expname, '{:06d}.tar'.format(i))
            torch.save({
                'global_step': global_step,
                'network_fn_state_dict': render_kwargs_train['network_fn'].state_dict(),
                'network_fine_state_dict': render_kwargs_train['network_fine'].state_dict(),
                'embedding_a_state_dict': render_kwargs_train['embedding_a'].state_dict(),
                'embedding_t_state_dict': render_kwargs_train['embedding_t'].state_dict(),
                'optimizer_state_dict': optimizer.state_dict(),
            }, path)
            print('Saved checkpoints at', path)
        if i % args.i_testset == 0 and i > 0:
            with torch.no_grad():
                for tag, dl, skip in (('trainset', train_dl, 10), ('testset', val_dl, 1)):
                    savedir = os.path.join(basedir, expname, '{}_{:06d}'.format(tag, i))
                    os.makedirs(savedir, exist_ok=True)
                    images, poses, index = _holdout(dl, device, skip)
                    print(tag, 'poses shape', poses.shape)
                    render_path(args, poses, hwf, args.chunk, render_kwargs_test, gt_imgs=images, savedir=savedir, img_ids=index)
                    print('Saved', tag)
        if i % args.i_print == 0 and rank == 0 and loss is not None:
            print(f"[TRAIN] Iter: {i} Loss: {loss.item()}  PSNR: {psnr.item()}  ({dt:.2f} s / epoch)")
        global_step += 1


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

"""Mirror of the render helpers DFNet training calls: /root/reference/script/feature/misc.py:203-289
(`render_nerfw_imgs`, `render_virtual_imgs`), dm/direct_pose_model.py:147-167 (`fix_coord_supp`) and
the cosine feature loss of feature/direct_feature_matching.py:114-136."""
import torch

from . import engine as _engine
from .rendering import render


class _BicubicFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, H, W):
        ctx.shape = img.shape[:2]
        return _engine.upsample_bicubic(img.detach(), H, W)

    @staticmethod
    def backward(ctx, g):
        return _engine.upsample_bicubic_backward(g.contiguous(), *ctx.shape), None, None


def upsample_bicubic(img, H, W):
    """nn.Upsample(size=(H, W), mode='bicubic') of an [h,w,C] image; differentiable (HIP adjoint kernel)."""
    if torch.is_grad_enabled() and img.requires_grad:
        return _BicubicFn.apply(img, int(H), int(W))
    return _engine.upsample_bicubic(img, int(H), int(W))


def fix_coord_supp(args, pose, world_setup_dict, device=None):
    """t <- ((t * pose_scale) + move_all_cam_vec) * pose_scale2 on [N,3,4] poses, in place."""
    move = torch.tensor(world_setup_dict['move_all_cam_vec'], dtype=pose.dtype, device=pose.device)
    pose[:, :3, 3] *= world_setup_dict['pose_scale']
    pose[:, :3, 3] += move
    pose[:, :3, 3] *= world_setup_dict['pose_scale2']
    return pose


def _render_one(args, pose_nerf, img_idx, hwf, render_kwargs_test):
    H, W, focal = hwf
    dev = torch.device("cuda", torch.cuda.current_device())
    c2w = pose_nerf[0, :3, :4].to(dev)
    if args.tinyimg:
        h, w = int(H // args.tinyscale), int(W // args.tinyscale)
        rgb, _, _, _ = render(h, w, focal / args.tinyscale, chunk=args.chunk, c2w=c2w, img_idx=img_idx, **render_kwargs_test)
        return upsample_bicubic(rgb, int(H), int(W))  # nn.Upsample(size=(H, W), mode='bicubic')
    rgb, _, _, _ = render(int(H), int(W), focal, chunk=args.chunk, c2w=c2w, img_idx=img_idx, **render_kwargs_test)
    return rgb


def render_nerfw_imgs(args, dl, hwf, device, render_kwargs_test, world_setup_dict):
    """Render every frame of `dl` at its ground-truth pose -> CPU tensors (targets [N,H,W,3],
    rgbs [N,H,W,3], poses [N,3,4], img_idxs [N,1,bins]) like misc.py:203-247.  Frames are kept in HBM and
    copied to the host once at the end."""
    targets, rgbs, poses, idxs = [], [], [], []
    with torch.no_grad():
        for batch_idx, (target, pose, img_idx) in enumerate(dl):
            if batch_idx % 10 == 0:
                print("renders {}/total {}".format(batch_idx, len(dl.dataset)))
            pose = pose.reshape(3, 4)
            pose_nerf = fix_coord_supp(args, pose.clone()[None, ...], world_setup_dict)
            rgbs.append(_render_one(args, pose_nerf, img_idx.to(device), hwf, render_kwargs_test))
            targets.append(target[0].permute(1, 2, 0))
            poses.append(pose)
            idxs.append(img_idx)
    return (torch.stack(targets).detach().cpu(), torch.stack(rgbs).detach().cpu(), torch.stack(poses).detach().cpu(),
            torch.stack(idxs).detach().cpu())


def render_virtual_imgs(args, pose_perturb, img_idxs, hwf, device, render_kwargs_test, world_setup_dict):
    """Render at perturbed poses (random view synthesis), misc.py:249-289 -> rgbs [N,H,W,3] on the CPU."""
    out = []
    with torch.no_grad():
        for k in range(pose_perturb.shape[0]):
            pose_nerf = fix_coord_supp(args, pose_perturb[k].clone()[None, ...].cpu(), world_setup_dict)
            out.append(_render_one(args, pose_nerf, img_idxs[k].to(device), hwf, render_kwargs_test))
    return torch.stack(out).detach().cpu()


def feature_loss(feature_rgb, feature_target, per_channel=False):
    """1 - mean cosine similarity between rendered and target features [C,H,W]
    (direct_feature_matching.py:114-136: per channel over the H*W axis, or per pixel over channels)."""
    C = feature_rgb.shape[0]
    fr, ft = feature_rgb.reshape(C, -1), feature_target.reshape(C, -1)
    cos = torch.nn.CosineSimilarity(dim=0 if per_channel else 1, eps=1e-6)
    return 1 - cos(fr, ft).mean()

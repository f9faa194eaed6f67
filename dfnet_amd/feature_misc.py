"""Mirror of the render helpers DFNet training calls: /root/reference/script/feature/misc.py:203-289
(`render_nerfw_imgs`, `render_virtual_imgs`), dm/direct_pose_model.py:147-167 (`fix_coord_supp`), the cosine
feature loss of feature/direct_feature_matching.py:114-136 and the pose-error evaluation of
feature/misc.py:49-131 (`compute_error_in_q`, `get_error_in_q`; pytorch3d's matrix_to_quaternion restated)."""
import math

import numpy as np
import torch

from . import _lib, engine as _engine
from ._lib import check, current_stream, ptr
from .rendering import render


class _BicubicFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, H, W):
        ctx.shape = img.shape[:2]
        return _engine.upsample_bicubic(img.detach(), H, W)

    @staticmethod
    def backward(ctx, g):
        return _engine.upsample_bicubic_backward(g.contiguous(), *ctx.shape), None, None


class _BicubicBatchFn(torch.autograd.Function):
    """[B,h,w,C] -> [B,H,W,C]: the frames of a mini-batch in one launch each way (dfn_upsample_bicubic_frames)."""

    @staticmethod
    def forward(ctx, imgs, H, W, nchw=False):
        imgs = imgs.detach().contiguous()
        ctx.shape = imgs.shape[1:3]
        ctx.nchw = bool(nchw)
        return _engine.upsample_bicubic_frames(imgs, H, W, nchw=ctx.nchw)

    @staticmethod
    def backward(ctx, g):
        return _engine.upsample_bicubic_frames_backward(g.contiguous(), *ctx.shape, nchw=ctx.nchw), None, None, None


def upsample_bicubic(img, H, W, nchw=False):
    """nn.Upsample(size=(H, W), mode='bicubic') of an [h,w,C] image or a batch [B,h,w,C]; differentiable (HIP adjoint kernel).
    nchw (batches only): the result as a contiguous [B,C,H,W] tensor — `.permute(0, 3, 1, 2)` folded into the kernel's store."""
    if img.dim() == 4:
        return _BicubicBatchFn.apply(img, int(H), int(W), bool(nchw))
    if torch.is_grad_enabled() and img.requires_grad:
        return _BicubicFn.apply(img, int(H), int(W))
    return _engine.upsample_bicubic(img, int(H), int(W))


_MOVE_CACHE = {}


def fix_coord_supp(args, pose, world_setup_dict, device=None):
    """t <- ((t * pose_scale) + move_all_cam_vec) * pose_scale2 on [N,3,4] poses (misc.py fix_coord_supp).  In place like the
    reference, except on a tensor autograd tracks: there the same three operations are applied out of place and the result is
    assembled with one concatenation (three in-place slice updates cost six CopySlices copies per step in the DFNet_dm loop)."""
    vec = world_setup_dict['move_all_cam_vec']
    key = (tuple(float(v) for v in vec), pose.dtype, pose.device)
    move = _MOVE_CACHE.get(key)
    if move is None:
        move = _MOVE_CACHE[key] = torch.tensor(vec, dtype=pose.dtype, device=pose.device)
    if pose.requires_grad or pose.grad_fn is not None:
        t = ((pose[:, :3, 3] * world_setup_dict['pose_scale']) + move) * world_setup_dict['pose_scale2']
        return torch.cat([pose[:, :3, :3], t[..., None]], -1) if pose.shape[1] == 3 else \
            torch.cat([torch.cat([pose[:, :3, :3], t[..., None]], -1), pose[:, 3:]], 1)
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


class _FeatureCosineFn(torch.autograd.Function):
    """mean_b feature_loss(f_r[b], f_t[b]) over the selected pyramid levels of two feature stacks [L,B,C,H,W] as one fused HIP
    forward / backward (dfn_feature_cosine_*): no index_select / permute copies, no gradient for the target stack."""

    @staticmethod
    def forward(ctx, fr, ft, levels, lazy_grad=False):
        import ctypes
        from ._lib import check, current_stream, load, ptr
        lib = load()
        L, B, C, H, W = fr.shape
        lv = (ctypes.c_int * len(levels))(*levels)
        nbytes = lib.dfn_feature_cosine_state_bytes(len(levels), B, C)
        state = torch.empty(nbytes, dtype=torch.uint8, device=fr.device)
        loss = torch.empty(1, device=fr.device)
        check(lib.dfn_feature_cosine_forward(ctypes.c_void_p(fr.data_ptr()), fr.stride(0), ctypes.c_void_p(ft.data_ptr()), ft.stride(0), lv,
                                             len(levels), B, C, H * W, ptr(loss), ctypes.c_void_p(state.data_ptr()), nbytes, current_stream()),
              "dfn_feature_cosine_forward")
        ctx.save_for_backward(fr, ft, state)
        ctx.levels = tuple(levels)
        ctx.lazy_grad = bool(lazy_grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from ._lib import check, current_stream, load, ptr
        fr, ft, state = ctx.saved_tensors
        levels = ctx.levels
        L, B, C, H, W = fr.shape
        G = torch.empty(L, B, C, H, W, device=fr.device)
        from . import engine as _eng
        if ctx.lazy_grad and _eng.POISON_UNREAD:
            G.fill_(float("nan"))   # debug (DFN_DEBUG_POISON_UNREAD): the planes no one promised to read, see engine.POISON_UNREAD
        if not ctx.lazy_grad:       # levels the loss does not read carry no gradient (lazy_grad: the consumer was told which levels
            for l in range(L):      # to read — the feature extractor's grad_levels hint — and the unread planes stay unwritten)
                if l not in levels:
                    G[l].zero_()
        gl = g.detach().reshape(1).to(torch.float32).contiguous()
        lv = (ctypes.c_int * len(levels))(*levels)
        check(load().dfn_feature_cosine_backward(ctypes.c_void_p(fr.data_ptr()), fr.stride(0), ctypes.c_void_p(ft.data_ptr()), ft.stride(0),
                                                 lv, len(levels), B, C, H * W, ptr(gl), ctypes.c_void_p(state.data_ptr()),
                                                 ctypes.c_void_p(G.data_ptr()), G.stride(0), current_stream()),
              "dfn_feature_cosine_backward")
        return G, None, None, None


class _DmLossFn(torch.autograd.Function):
    """The rest of the DFNet_dm loss block (direct_feature_matching.py:359-370) as one HIP forward / backward pair
    (dfn_dm_loss_*): photo_loss = mean((rgb - target)^2), pose_loss = mse_loss(pose_, pose), loss = w[0] pose_loss + w[1] photo_loss
    + w[2] feat_loss.  Returns (loss, photo_loss, pose_loss); only `loss` carries gradient (to rgb, pose_ and feat_loss)."""

    @staticmethod
    def forward(ctx, rgb, target, pose_, pose_gt, feat_l, w_pose, w_photo, w_feat):
        import ctypes
        from ._lib import check, current_stream, load, ptr
        lib = load()
        rgb, target = rgb.detach().contiguous(), target.detach().contiguous()
        pose_, pose_gt = pose_.detach().contiguous(), pose_gt.detach().contiguous()
        feat = feat_l.detach().reshape(1).contiguous()
        out = torch.empty(4, device=rgb.device)
        scratch = torch.empty(lib.dfn_dm_loss_scratch_bytes(), dtype=torch.uint8, device=rgb.device)
        check(lib.dfn_dm_loss_forward(ptr(rgb), ptr(target), rgb.numel(), ptr(pose_), ptr(pose_gt), pose_.numel(), ptr(feat), float(w_pose),
                                      float(w_photo), float(w_feat), ptr(out), ctypes.c_void_p(scratch.data_ptr()), current_stream()),
              "dfn_dm_loss_forward")
        ctx.save_for_backward(rgb, target, pose_, pose_gt)
        ctx.w = (float(w_pose), float(w_photo), float(w_feat))
        ctx.pose_shape = pose_.shape
        photo, pl = out[1], out[2]
        ctx.mark_non_differentiable(photo, pl)
        return out[0], photo, pl

    @staticmethod
    def backward(ctx, g, _gp, _gq):
        from ._lib import check, current_stream, load, ptr
        rgb, target, pose_, pose_gt = ctx.saved_tensors
        g_rgb, g_pose = torch.empty_like(rgb), torch.empty_like(pose_)
        g_feat = torch.empty(1, device=rgb.device)
        gl = g.detach().reshape(1).to(torch.float32).contiguous()
        check(load().dfn_dm_loss_backward(ptr(rgb), ptr(target), rgb.numel(), ptr(pose_), ptr(pose_gt), pose_.numel(), *ctx.w, ptr(gl),
                                          ptr(g_rgb), ptr(g_pose), ptr(g_feat), current_stream()), "dfn_dm_loss_backward")
        return g_rgb, None, g_pose, None, g_feat.reshape(()), None, None, None


def dm_combined_loss(rgb, target, pose_, pose_gt, feat_l, w):
    """(loss, photo_loss, pose_loss) of the combine_loss branch (:359-368) — fused on the GPU, the reference's torch expression elsewhere."""
    if rgb.is_cuda and rgb.dtype == torch.float32 and rgb.shape == target.shape and pose_.shape == pose_gt.shape:
        return _DmLossFn.apply(rgb, target, pose_, pose_gt, feat_l, w[0], w[1], w[2])
    photo_l = torch.mean((rgb - target) ** 2)
    pose_l = torch.nn.functional.mse_loss(pose_, pose_gt)
    return w[0] * pose_l + w[1] * photo_l + w[2] * feat_l, photo_l, pose_l


def feature_loss_batch(fr, ft, levels, per_channel=False, lazy_grad=False):
    """The feature term of the DFNet_dm step (direct_feature_matching.py:352-358): the selected levels of the rendered and target
    stacks [L,B,C,H,W] -> [B, l*C, H, W] (preprocess_features_for_loss), feature_loss per image, mean over the batch.  On the
    GPU (per_channel False, the default) this is one fused kernel pair; per_channel=True and CPU tensors take the reference's own
    composition of torch ops."""
    levels = [int(l) for l in levels]
    if (not per_channel and fr.shape == ft.shape and len(set(levels)) == len(levels) <= 8 and _stack_layout(fr) is not None
            and _stack_layout(ft) is not None and not ft.requires_grad):
        return _FeatureCosineFn.apply(fr, ft, tuple(levels), bool(lazy_grad))
    idx = torch.tensor(levels, device=fr.device)
    def prep(f):
        f = torch.index_select(f, 0, idx).permute(1, 0, 2, 3, 4)
        return f.reshape(f.shape[0], -1, f.shape[3], f.shape[4])
    f_r, f_t = prep(fr), prep(ft)
    return torch.stack([feature_loss(f_r[b], f_t[b], per_channel=per_channel) for b in range(f_r.shape[0])]).mean()


def PoseLoss(args, pose_, pose, device):
    """MSE between predicted and ground-truth [B,12] poses (misc.py:321-325)."""
    return torch.nn.functional.mse_loss(pose_.to(device), pose)


def _stack_layout(t):
    """(level_stride in floats) if t is an fp32 CUDA stack [L,B,C,H,W] whose levels are dense [B,C,H,W] blocks — a
    contiguous tensor or one half of a siamese [L,2B,C,H,W] tensor — else None."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 5):
        return None
    L, B, C, H, W = t.shape
    if tuple(t.stride()[1:]) != (C * H * W, H * W, W, 1) or t.stride(0) < B * C * H * W:
        return None
    return t.stride(0)


class _TripletFn(torch.autograd.Function):
    """The three triplet losses of misc.py:355-435 as one fused HIP forward / backward (dfn_triplet_loss_*)."""

    @staticmethod
    def forward(ctx, f1, f2, margin, mining):
        import ctypes
        from ._lib import check, current_stream, load, ptr
        lib = load()
        L, B, C, H, W = f1.shape
        ls1, ls2 = _stack_layout(f1), _stack_layout(f2)
        nbytes = lib.dfn_triplet_loss_state_bytes(L, B, C * H)
        state = torch.empty(nbytes, dtype=torch.uint8, device=f1.device)
        loss = torch.empty(1, device=f1.device)
        check(lib.dfn_triplet_loss_forward(ctypes.c_void_p(f1.data_ptr()), ls1, ctypes.c_void_p(f2.data_ptr()), ls2, L, B, C * H, W,
                                           float(margin), int(mining), ptr(loss), ctypes.c_void_p(state.data_ptr()), nbytes,
                                           current_stream()), "dfn_triplet_loss_forward")
        ctx.save_for_backward(f1, f2, state)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        import ctypes
        from ._lib import check, current_stream, load, ptr
        f1, f2, state = ctx.saved_tensors
        lib = load()
        L, B, C, H, W = f1.shape
        slab = C * H * W
        ls1, ls2 = f1.stride(0), f2.stride(0)
        # the two halves of one siamese tensor: gradients go into the halves of ONE tensor as well, so the consumer
        # (DFNet's backward) gets d L/d features in its own layout without a concatenation
        d = (f1.data_ptr() - f2.data_ptr()) // 4
        if ls1 == ls2 == 2 * B * slab and abs(d) == B * slab:
            G = torch.empty(L, 2 * B, C, H, W, device=f1.device)
            g1, g2 = (G[:, B:], G[:, :B]) if d > 0 else (G[:, :B], G[:, B:])
        else:
            g1, g2 = torch.empty(L, B, C, H, W, device=f1.device), torch.empty(L, B, C, H, W, device=f1.device)
        gl = g.detach().reshape(1).to(torch.float32).contiguous()
        check(lib.dfn_triplet_loss_backward(ctypes.c_void_p(f1.data_ptr()), ls1, ctypes.c_void_p(f2.data_ptr()), ls2, L, B, C * H, W,
                                            ptr(gl), ctypes.c_void_p(state.data_ptr()), ctypes.c_void_p(g1.data_ptr()), g1.stride(0),
                                            ctypes.c_void_p(g2.data_ptr()), g2.stride(0), current_stream()),
              "dfn_triplet_loss_backward")
        return g1, g2, None, None


class _PyramidTripletFn(torch.autograd.Function):
    """The triplet losses on a pair of dfnet.FeaturePyramid (the two streams of one siamese training forward): the loss of the
    enlarged stacks from the low-resolution levels (dfn_dfnet_triplet_pyramid_forward).  The tokens carry d L / d loss back to
    DFNet's backward, which takes the gradient of the loss from the device state left here."""

    @staticmethod
    def forward(ctx, tok1, tok2, st, f1_half, margin, mining):
        if st.triplet is not None:
            raise RuntimeError("one triplet loss per siamese forward: its state is what DFNet's backward differentiates")
        loss, state = st.engine.triplet_pyramid_forward(st.tape, st.B, st.H, st.W, st.upH, st.upW, f1_half, margin, mining,
                                                        feature_images=st.feature_images)
        st.triplet = (state, f1_half)
        return loss

    @staticmethod
    def backward(ctx, g):
        g = g.detach().reshape(()).to(torch.float32)
        return g, g, None, None, None, None


def _fused_triplet(f1, f2, margin, mining):
    """The HIP path when both stacks live on the GPU in a layout the kernels address in place; None otherwise (CPU
    tensors — the torch composition below is then the reference's own code path, not a fallback of a GPU op)."""
    from .dfnet import FeaturePyramid
    if isinstance(f1, FeaturePyramid) or isinstance(f2, FeaturePyramid):
        if not (isinstance(f1, FeaturePyramid) and isinstance(f2, FeaturePyramid)) or f1.state is not f2.state or f1.half == f2.half:
            raise ValueError("the pyramid triplet loss takes the two streams of ONE siamese training forward")
        return _PyramidTripletFn.apply(f1.token, f2.token, f1.state, f1.half, float(margin), mining)
    if f1.shape == f2.shape and _stack_layout(f1) is not None and _stack_layout(f2) is not None:
        return _TripletFn.apply(f1, f2, float(margin), mining)
    if f1.is_cuda:
        raise ValueError("triplet loss on the GPU needs fp32 stacks [L,B,C,H,W] with dense levels, got "
                         f"{tuple(f1.shape)} {f1.stride()} / {tuple(f2.shape)} {f2.stride()}")
    return None


def _triplet(anchor, positive, negative, margin):
    # nn.TripletMarginLoss(margin, p=2, reduction='mean'): pairwise L2 distance over the LAST axis (eps 1e-6)
    return torch.nn.functional.triplet_margin_loss(anchor, positive, negative, margin=margin, p=2, reduction='mean')


def triplet_loss(f1, f2, margin=1.):
    """Naive triplet loss on feature stacks [lvl,B,C,H,W]: negative = the next image of the batch (misc.py:355-369)."""
    fused = _fused_triplet(f1, f2, margin, 0)
    if fused is not None:
        return fused
    return _triplet(f1, f2, torch.roll(f2, shifts=1, dims=1), margin)


def triplet_loss_hard_negative_mining(f1, f2, margin=1.):
    """In-triplet hard negative with anchor swap, two cases (misc.py:371-397)."""
    fused = _fused_triplet(f1, f2, margin, 1)
    if fused is not None:
        return fused
    a_neg, neg = torch.roll(f1, shifts=1, dims=1), torch.roll(f2, shifts=1, dims=1)
    with torch.no_grad():
        case1 = torch.nn.functional.mse_loss(f1, neg)
        case2 = torch.nn.functional.mse_loss(f2, a_neg)
    return _triplet(f1, f2, neg, margin) if case1 < case2 else _triplet(f2, f1, a_neg, margin)


def triplet_loss_hard_negative_mining_plus(f1, f2, margin=1.):
    """In-triplet hard negative, four cases: the closest of (anchor, negative), (positive, anchor_negative),
    (anchor, anchor_negative), (positive, negative) decides which pair anchors the loss (misc.py:399-435)."""
    fused = _fused_triplet(f1, f2, margin, 2)
    if fused is not None:
        return fused
    anchor, positive = f1, f2
    a_neg, neg = torch.roll(f1, shifts=1, dims=1), torch.roll(f2, shifts=1, dims=1)
    mse = torch.nn.functional.mse_loss
    with torch.no_grad():
        case = int(torch.argmin(torch.stack([mse(anchor, neg), mse(positive, a_neg), mse(anchor, a_neg), mse(positive, neg)])))
    if case == 0:
        return _triplet(anchor, positive, neg, margin)
    if case == 1:
        return _triplet(positive, anchor, a_neg, margin)
    if case == 2:
        return _triplet(anchor, positive, a_neg, margin)
    return _triplet(positive, anchor, neg, margin)


def perturb_rotation(c2w, theta, phi, psi=0):
    """Rotate a [3,4] camera-to-world about x (phi), then y (theta), then z (psi), degrees (misc.py:28-47, 437-446;
    the y rotation has the reference's sign convention: [[c,0,-s],[0,1,0],[s,0,c]])."""
    import numpy as np
    ph, th, ps = (np.deg2rad(float(v)) for v in (phi, theta, psi))
    rx = np.array([[1, 0, 0, 0], [0, np.cos(ph), -np.sin(ph), 0], [0, np.sin(ph), np.cos(ph), 0], [0, 0, 0, 1]], dtype=np.float64)
    ry = np.array([[np.cos(th), 0, -np.sin(th), 0], [0, 1, 0, 0], [np.sin(th), 0, np.cos(th), 0], [0, 0, 0, 1]], dtype=np.float64)
    rz = np.array([[np.cos(ps), -np.sin(ps), 0, 0], [np.sin(ps), np.cos(ps), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=np.float64)
    m = np.concatenate([np.asarray(c2w, dtype=np.float64), [[0, 0, 0, 1]]], 0)
    return (rz @ (ry @ (rx @ m)))[:3, :4]


def perturb_single_render_pose(poses, x, angle):
    """Random view synthesis: one [3,4] pose -> [1,3,4], rotation uniform in +-angle degrees per axis, camera position
    = the ORIGINAL position + uniform(+-x) per axis (misc.py:448-483; the rotation does not move the camera)."""
    import numpy as np
    c2w = np.asarray(poses, dtype=np.float64)
    loc = c2w[:, 3].copy()
    theta, phi, psi = np.random.uniform(-angle, angle, 3)
    new = perturb_rotation(c2w, theta, phi, psi)
    new[:, 3] = loc + np.random.uniform(-x, x, 3)
    return new[None]


class _PoseOrthoFn(torch.autograd.Function):
    """pose [B,3,4] -> [U V^T | t] of the rotation block's SVD, as the orthogonal polar factor (csrc/pose_polar.hip)."""

    @staticmethod
    def forward(ctx, pose):
        p = pose.detach().contiguous().float()
        out = torch.empty_like(p)
        check(_lib.load().dfn_pose_orthogonalize(ptr(p), p.shape[0], ptr(out), current_stream()), "dfn_pose_orthogonalize")
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        g = g.contiguous().float()
        gin = torch.empty_like(p)
        check(_lib.load().dfn_pose_orthogonalize_backward(ptr(p), ptr(g), p.shape[0], ptr(gin), current_stream()),
              "dfn_pose_orthogonalize_backward")
        return gin


def orthogonalize_pose(pose):
    """The reference's `svd_reg` (direct_feature_matching.py:85-92, misc.py:68-72): pose[:, :3, :3] <- U V^T of its SVD, for a
    pose batch [B,3,4].  GPU tensors: one closed-form kernel per direction (no rocSOLVER / BLAS call, differentiable); CPU
    tensors (the evaluation loop's host-side copies): torch.svd, as the reference."""
    if pose.is_cuda:
        return _PoseOrthoFn.apply(pose.reshape(-1, 3, 4))
    u, s, v = torch.svd(pose[:, :3, :3])
    return torch.cat([torch.matmul(u, v.transpose(-2, -1)), pose[:, :3, 3:]], -1)


def freeze_bn_layer(model):
    """--freezeBN, part 1 (utils/utils.py:18-28): BatchNorm weight / bias stop requiring grad."""
    print("Freezing BatchNorm Layers...")
    for module in model.modules():
        if isinstance(module, torch.nn.BatchNorm2d):
            module.weight.requires_grad_(False)
            module.bias.requires_grad_(False)
    return model


def freeze_bn_layer_train(model):
    """--freezeBN, part 2 (utils/utils.py:30-39): BatchNorm modules back to eval() after model.train()."""
    for module in model.modules():
        if isinstance(module, torch.nn.BatchNorm2d):
            module.eval()
    return model


def matrix_to_quaternion(R):
    """[...,3,3] rotation matrices -> unit quaternions [...,4], real part first (pytorch3d.transforms convention;
    the error below only uses |q1.q2|, so the sign / branch choice is immaterial).  Shepperd's method: pick the
    largest of the four squared components for a stable division."""
    R = torch.as_tensor(R, dtype=torch.float64)
    m00, m11, m22 = R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]
    q_abs = torch.sqrt(torch.clamp(torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22,
                                                1 - m00 + m11 - m22, 1 - m00 - m11 + m22], -1), min=0.))
    cand = torch.stack([
        torch.stack([q_abs[..., 0] ** 2, R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]], -1),
        torch.stack([R[..., 2, 1] - R[..., 1, 2], q_abs[..., 1] ** 2, R[..., 1, 0] + R[..., 0, 1], R[..., 0, 2] + R[..., 2, 0]], -1),
        torch.stack([R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] + R[..., 0, 1], q_abs[..., 2] ** 2, R[..., 1, 2] + R[..., 2, 1]], -1),
        torch.stack([R[..., 1, 0] - R[..., 0, 1], R[..., 2, 0] + R[..., 0, 2], R[..., 2, 1] + R[..., 1, 2], q_abs[..., 3] ** 2], -1),
    ], -2)
    cand = cand / (2.0 * q_abs[..., None].clamp(min=0.1))
    best = q_abs.argmax(-1)
    q = torch.gather(cand, -2, best[..., None, None].expand(*best.shape, 1, 4))[..., 0, :]
    return (q / q.norm(dim=-1, keepdim=True)).float()


def pose_errors(pred, gt):
    """(translation error, rotation error in degrees) between [N,3,4] poses: ||t - t'|| and 2 acos|q.q'|."""
    pred, gt = torch.as_tensor(pred, dtype=torch.float32).reshape(-1, 3, 4), torch.as_tensor(gt, dtype=torch.float32).reshape(-1, 3, 4)
    q1, q2 = matrix_to_quaternion(gt[:, :3, :3]), matrix_to_quaternion(pred[:, :3, :3])
    d = torch.clamp((q1 * q2).sum(-1).abs(), -1., 1.)
    return (gt[:, :3, 3] - pred[:, :3, 3]).norm(dim=-1), 2 * torch.acos(d) * 180 / math.pi


def compute_error_in_q(args, dl, model, device, results, batch_size=1):
    """Per-frame [translation error (m), rotation error (deg)] of the pose regressor over `dl` (misc.py:49-116):
    prediction orthogonalised by SVD, errors from quaternions."""
    pred_x, gt_x, thetas = [], [], []
    i = 0
    for batch in dl:
        data, pose = batch[0], batch[1]
        with torch.no_grad():
            _, predict = model(data.to(device))
            predict = orthogonalize_pose(predict.reshape(-1, 3, 4)).cpu()   # on the device, before the host copy
        pose = torch.as_tensor(pose, dtype=torch.float32).reshape(-1, 3, 4)
        ex, eq = pose_errors(predict, pose)
        for k in range(predict.shape[0]):
            results[i, :] = [float(ex[k]), float(eq[k])]
            pred_x.append(predict[k, :3, 3].numpy())
            gt_x.append(pose[k, :3, 3].numpy())
            thetas.append(float(eq[k]))
            i += 1
    return results, {"pose": np.array(pred_x), "pose_gt": np.array(gt_x), "theta": np.array(thetas)}


def get_error_in_q(args, dl, model, sample_size, device, batch_size=1):
    """Median / mean pose error of the regressor over `dl` (misc.py:118-131), printed like the reference."""
    model.eval()
    results = np.zeros((sample_size, 2))
    results, vis_info = compute_error_in_q(args, dl, model, device, results, batch_size)
    median_result, mean_result = np.median(results, axis=0), np.mean(results, axis=0)
    print('Median error {}m and {} degrees.'.format(median_result[0], median_result[1]))
    print('Mean error {}m and {} degrees.'.format(mean_result[0], mean_result[1]))
    return median_result, mean_result

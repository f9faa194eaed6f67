"""Forward pass of the DFNet_dm step, mirroring /root/reference/script/feature/direct_feature_matching.py:
`preprocess_features_for_loss` (:41-50), `inference_pose_regression` (:63-93) and the forward half of
`train_on_batch` (:322-376): pose regression -> (SVD orthogonalisation) -> scene rescale -> NeRF-H render at
quarter resolution -> bicubic x4 -> siamese DFNet features -> cosine feature loss (+ photometric / pose terms).

Every arithmetic-heavy stage runs in the HIP library (DFNet forward, render, bicubic); what remains here are
the few-element reductions of the loss.  `matching_step_forward` returns the losses only; `matching_step_grad`
also runs loss.backward() down to the PREDICTED POSE: the loss reductions by torch autograd (a few element-wise
ops on device tensors), everything below them by the HIP gradient kernels (DFNet input gradient, bicubic adjoint,
render gradient).  `train_on_batch` / `train_on_epoch` are the reference's step itself (:322-410): the pose
regressor is tracked too, loss.backward() continues through SVD / reshape (torch) into the HIP weight-gradient
kernels of its conv stack and fc_pose (dfn_dfnet_backward_params), and optimizer.step() (a torch optimizer over
the module's parameters) updates it; the module then re-packs the changed weights into MFMA fragments on the
device (dfn_dfnet_refresh_pose_params_device, ~1 ms; bit-identical to a host commit).
The reference renders only pose 0 of the batch (:342, i.e. batch size 1 in effect); this implementation renders
every pose of the batch."""
import torch

from . import dist as ddist
from .feature_misc import dm_combined_loss, feature_loss, feature_loss_batch, fix_coord_supp, orthogonalize_pose, upsample_bicubic
from .rendering import render, render_frames

PRUNE_FEATURE_LEVELS = True   # _losses: compute only the pyramid levels the feature loss reads (False: all three, like the reference)
OVERLAP_TARGET_FEATURES = True   # train_on_batch: the target frames' features (independent of the predicted pose) on a side stream
_SIDE_STREAMS = {}


def preprocess_features_for_loss(feature):
    """[L,B,C,H,W] -> [B, L*C, H, W] (:41-50)."""
    feature = feature.permute(1, 0, 2, 3, 4)
    B, L, C, H, W = feature.size()
    return feature.reshape((B, L * C, H, W))


def inference_pose_regression(args, data, device, model, retFeature=False, isSingleStream=True, return_pose=True):
    """DFNet forward wrapper (:63-93): returns (features, pose [B,3,4]) or (features, None)."""
    inputs = data.to(device)
    _, _, H, W = data.size()
    features, predict_pose = model(inputs, return_feature=retFeature, isSingleStream=isSingleStream,
                                   return_pose=return_pose, upsampleH=H, upsampleW=W)
    if not return_pose:
        return features, predict_pose
    pose = predict_pose.reshape(inputs.shape[0], 3, 4)
    if getattr(args, "svd_reg", False):
        # R <- U V^T (:85-88) = the orthogonal polar factor of R: one closed-form kernel forward, one backward (csrc/pose_polar.hip)
        # instead of rocSOLVER's SVD, two BLAS products and the slice / cat copies around them
        pose = orthogonalize_pose(pose)
    return features, pose


def matching_step_forward(args, data, model, feat_model, pose, img_idx, hwf, half_res, device, world_setup_dict,
                          **render_kwargs_test):
    """Losses of one DFNet_dm step without the update.  data [B,3,H,W] in [0,1], pose [B,12|3x4] ground truth,
    img_idx [B,bins].  Returns dict(loss, pose_loss, photo_loss, feat_loss, psnr, rgb [B,3,H,W], pose_pred)."""
    H, W, focal = hwf
    H, W = int(H), int(W)
    data = data.to(device)
    B = data.shape[0]
    with torch.no_grad():
        _, pose_ = inference_pose_regression(args, data, device, model, retFeature=False)
        pose_nerf = fix_coord_supp(args, pose_.clone(), world_setup_dict, device=device)
        img_idx = torch.as_tensor(img_idx, dtype=torch.float32, device=device).reshape(B, -1)
        rgb = _render_batch(H, W, focal, pose_nerf, img_idx, half_res, render_kwargs_test)
        feats, _ = inference_pose_regression(args, torch.cat([data, rgb]), device, feat_model, retFeature=True,
                                             isSingleStream=False, return_pose=False)
        feat_l = feature_loss_batch(feats[1], feats[0], args.feature_matching_lvl, per_channel=args.per_channel)
        photo_l = torch.mean((rgb - data) ** 2)
        pose_l = torch.nn.functional.mse_loss(pose_.reshape(B, 12), torch.as_tensor(pose, device=device).reshape(B, 12).float())
        if getattr(args, "combine_loss", False):
            w = args.combine_loss_w
            loss = w[0] * pose_l + w[1] * photo_l + w[2] * feat_l
        else:
            loss = feat_l
        psnr = -10. * torch.log10(photo_l)
    return dict(loss=loss, pose_loss=pose_l, photo_loss=photo_l, feat_loss=feat_l, psnr=psnr, rgb=rgb, pose_pred=pose_)


def matching_step_grad(args, data, model, feat_model, pose, img_idx, hwf, half_res, device, world_setup_dict,
                       **render_kwargs_test):
    """One DFNet_dm step up to and including d loss / d predicted pose (train_on_batch, :322-370, without the
    regressor's own backward / optimizer.step()).  Returns the dict of matching_step_forward plus
    `grad_pose` [B,3,4]: the gradient that the reference's loss.backward() hands to the pose regressor's output."""
    H, W, focal = hwf
    H, W = int(H), int(W)
    data = data.to(device)
    B = data.shape[0]
    with torch.no_grad():
        _, pose_ = inference_pose_regression(args, data, device, model, retFeature=False)
    pose_ = pose_.detach().requires_grad_(True)
    with torch.enable_grad():
        pose_nerf = fix_coord_supp(args, pose_, world_setup_dict, device=device)   # tracked: applied out of place
        img_idx = torch.as_tensor(img_idx, dtype=torch.float32, device=device).reshape(B, -1)
        rgb = _render_batch(H, W, focal, pose_nerf, img_idx, half_res, render_kwargs_test)
        loss, photo_l, feat_l, pose_l = _losses(args, data, rgb, pose_, pose, feat_model, device, parts=True)
        loss.backward()
    with torch.no_grad():
        psnr = -10. * torch.log10(photo_l)
    return dict(loss=loss.detach(), pose_loss=pose_l.detach(), photo_loss=photo_l.detach(), feat_loss=feat_l.detach(),
                psnr=psnr, rgb=rgb.detach(), pose_pred=pose_.detach(), grad_pose=pose_.grad)



_PINNED = {}


def _to_device_async(t, device, dtype=torch.float32):
    """A small host tensor (ground-truth poses, histogram vectors) onto `device` WITHOUT stalling the host: a pageable host-to-device copy
    waits for everything queued on the stream before it, which in the middle of a step leaves the GPU idle while the host catches up
    (two such copies cost 0.15 ms of the 13.6 ms DFNet_dm step).  Staged through a small ring of page-locked buffers instead."""
    t = torch.as_tensor(t)
    if t.is_cuda or not torch.device(device).type == "cuda":
        return t.to(device=device, dtype=dtype)
    t = t.to(dtype)
    key = (tuple(t.shape), dtype, torch.device(device))
    ring = _PINNED.get(key)
    if ring is None:
        ring = _PINNED[key] = {"bufs": [torch.empty(t.shape, dtype=dtype).pin_memory() for _ in range(4)], "evs": [None] * 4, "i": 0}
    k = ring["i"] % 4
    ring["i"] += 1
    if ring["evs"][k] is not None:
        ring["evs"][k].synchronize()      # the copy that last used this buffer (four steps ago) has long finished
    ring["bufs"][k].copy_(t)
    out = ring["bufs"][k].to(device, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    ring["evs"][k] = ev
    return out


def _render_batch(H, W, focal, pose_nerf, img_idx, half_res, render_kwargs_test):
    """The per-frame render loop of the reference's step (:340-348; it renders only pose_nerf[0], its batch size being 1) for the
    whole mini-batch: one batched launch per render stage (rendering.render_frames), then the bicubic enlargement per frame.
    Returns rgb [B,3,H,W], attached to pose_nerf."""
    if half_res:
        small = render_frames(H // 4, W // 4, focal / 4, pose_nerf, img_idx, **render_kwargs_test)
        return upsample_bicubic(small, H, W, nchw=True)
    return render_frames(H, W, focal, pose_nerf, img_idx, **render_kwargs_test).permute(0, 3, 1, 2)


def _target_features(args, data, feat_model, device):
    """Features of the target frames (no gradient): the levels the loss reads when PRUNE_FEATURE_LEVELS."""
    if hasattr(feat_model, "engine") and PRUNE_FEATURE_LEVELS:
        feat_model.engine().feature_levels_hint = sorted(set(int(l) for l in args.feature_matching_lvl))
    with torch.no_grad():
        ft, _ = inference_pose_regression(args, data, device, feat_model, retFeature=True, isSingleStream=True, return_pose=False)
    return ft


def _target_features_async(args, data, feat_model, device):
    """The same on a side stream, started before the pose regression: the target features depend on nothing the step computes, and both
    DFNet forwards leave most of the chip idle at training resolutions (conv5_x of a 4 x 240x320 batch is 64 workgroups).  Returns a
    thunk that makes the current stream wait for them and hands them over.  CPU tensors / no engine: computed in place."""
    if not (OVERLAP_TARGET_FEATURES and data.is_cuda and hasattr(feat_model, "engine")):
        return None
    side = _SIDE_STREAMS.get(data.device)
    if side is None:
        side = _SIDE_STREAMS[data.device] = torch.cuda.Stream(device=data.device)
    side.wait_stream(torch.cuda.current_stream(data.device))
    with torch.cuda.stream(side):
        ft = _target_features(args, data, feat_model, device)
    data.record_stream(side)

    def take():
        cur = torch.cuda.current_stream(data.device)
        cur.wait_stream(side)          # also orders the rendered frames' forward after it: both use the extractor's workspace
        for t in ft:
            t.record_stream(cur)
        return ft
    return take


def _losses(args, data, rgb, pose_, pose, feat_model, device, parts=False, target_features=None):
    """The loss block of train_on_batch (:350-370) on a tracked rendered batch rgb [B,3,H,W].  target_features: the thunk of
    _target_features_async, if the caller started them early."""
    B = data.shape[0]
    # The reference feeds cat([data, rgb]) through the siamese forward (:351); images are independent in this network,
    # so the target half (no gradient) and the rendered half (tracked) are run as two single-stream calls: the
    # backward then touches only the B rendered images instead of 2 B.
    # The loss reads only the levels of args.feature_matching_lvl (the reference computes all three and index_selects, :354-357): the
    # feature extractor is told so — the forwards stop after the deepest of them and skip the other adaptation branches (their planes
    # come back as zeros), the backward starts from them without scanning the gradient stack.  PRUNE_FEATURE_LEVELS = False computes
    # every level like the reference; loss and gradients are bit-identical either way (tests/test_gpu_grad.py).
    lv = sorted(set(int(l) for l in args.feature_matching_lvl))
    ft = target_features() if target_features is not None else _target_features(args, data, feat_model, device)
    hinted = hasattr(feat_model, "engine") and rgb.requires_grad and torch.is_grad_enabled()
    if hasattr(feat_model, "engine"):
        feat_model.engine().grad_levels_hint = lv
        if PRUNE_FEATURE_LEVELS:
            feat_model.engine().feature_levels_hint = lv
    fr, _ = inference_pose_regression(args, rgb, device, feat_model, retFeature=True, isSingleStream=True, return_pose=False)
    # (hinted: the extractor's backward was just told to read the gradient of levels `lv` only, so the loss need not zero the others)
    feat_l = feature_loss_batch(fr[0], ft[0], args.feature_matching_lvl, per_channel=args.per_channel,
                                lazy_grad=hinted and sorted(set(int(l) for l in args.feature_matching_lvl)) == lv)
    pose_gt = (pose if torch.is_tensor(pose) and pose.is_cuda else _to_device_async(pose, device)).reshape(B, 12).float()
    if getattr(args, "combine_loss", False):   # (:359-368) one fused forward / backward pair on the GPU
        loss, photo_l, pose_l = dm_combined_loss(rgb, data, pose_.reshape(B, 12), pose_gt, feat_l, args.combine_loss_w)
    else:
        # direct_feature_matching.py:371-376 defines `loss` under `if args.combine_loss:` only and then calls loss.backward(): without the
        # flag the reference's step dies with NameError.  Same error here (round-5 advisor: a silent `loss = feat_l` was an invented
        # behaviour); config_dfnetdm.txt sets combine_loss.
        raise NameError("name 'loss' is not defined: the reference's train_on_batch (direct_feature_matching.py:371-376) defines the loss "
                        "only under --combine_loss")
    return (loss, photo_l, feat_l, pose_l) if parts else (loss, photo_l)


def train_on_batch(args, data, model, feat_model, pose, img_idx, hwf, optimizer, half_res, device, world_setup_dict,
                   **render_kwargs_test):
    """One optimisation step of DFNet_dm (:322-390): returns (loss, psnr) as 1-element numpy arrays like the reference (which means
    waiting for the device: train_on_epoch uses train_on_batch_device and waits once per epoch)."""
    import numpy as np
    loss, psnr = train_on_batch_device(args, data, model, feat_model, pose, img_idx, hwf, optimizer, half_res, device, world_setup_dict,
                                       **render_kwargs_test)
    return np.array([float(loss)]), np.array([float(psnr)])


def train_on_batch_device(args, data, model, feat_model, pose, img_idx, hwf, optimizer, half_res, device, world_setup_dict,
                          **render_kwargs_test):
    """train_on_batch without the host round trip: (loss, psnr) as 0-dim device tensors.  Converting them to Python floats after
    every step (the reference's `.item()`, :376-390) drains the stream and leaves the GPU idle while the host enqueues the next
    step's ~190 launches (0.4 ms of a 14 ms step at batch 4)."""
    H, W, focal = hwf
    H, W = int(H), int(W)
    data = data.to(device)
    B = data.shape[0]
    # the step's small host inputs go up first, through page-locked staging: no host-to-device copy waits in the middle of the step
    img_idx = _to_device_async(img_idx, device).reshape(B, -1)
    pose = _to_device_async(pose, device)
    target_features = _target_features_async(args, data, feat_model, device)   # side stream, beside the pose regression
    with torch.enable_grad():
        _, pose_ = inference_pose_regression(args, data, device, model, retFeature=False)
        pose_nerf = fix_coord_supp(args, pose_ if pose_.requires_grad else pose_.clone(), world_setup_dict, device=device)
        # the reference renders pose 0 only (:342); every pose of the batch here, as one ray batch
        rgb = _render_batch(H, W, focal, pose_nerf, img_idx, half_res, render_kwargs_test)
        loss, photo_l = _losses(args, data, rgb, pose_, pose, feat_model, device, target_features=target_features)
        loss.backward()
    ddist.allreduce_gradients(model.parameters())   # data-parallel: one all-reduce of the regressor's gradients per step
    optimizer.step()
    optimizer.zero_grad()
    with torch.no_grad():
        psnr = -10. * torch.log10(photo_l.detach())
    return loss.detach(), psnr


def train_on_epoch(args, data_loaders, model, feat_model, hwf, optimizer, half_res, device, world_setup_dict,
                   **render_kwargs_test):
    """One epoch over the training loader (:392-410): mean (loss, psnr).  BatchNorm is not on the regressor's path."""
    import numpy as np
    train_dl = data_loaders[0]
    losses, psnrs = [], []
    for data, pose, img_idx in train_dl:
        l, p = train_on_batch_device(args, data, model, feat_model, pose, img_idx, hwf, optimizer, half_res, device,
                                     world_setup_dict, **render_kwargs_test)
        losses.append(l.reshape(1))
        psnrs.append(p.reshape(1))
    if not losses:
        return float("nan"), float("nan")
    # one wait per epoch; the means in float64 on the host like np.mean over the reference's per-step floats
    return (float(np.mean(torch.cat(losses).cpu().numpy().astype(np.float64))),
            float(np.mean(torch.cat(psnrs).cpu().numpy().astype(np.float64))))


def eval_on_batch(args, data, model, feat_model, pose, img_idx, hwf, half_res, device, world_setup_dict, **render_kwargs_test):
    """One validation step (:178-213): pose regression, then N_rand random rays of the batch's frames rendered at FULL resolution at
    the predicted poses.  Returns (pose MSE vs the ground truth, PSNR of those rays vs the frames) as 1-element numpy arrays.
    The reference hands `render` the [B, bins] histogram rows unchanged, which only works for B = 1; here every ray takes the row
    of the frame it came from."""
    import numpy as np
    from . import engine as _engine
    H, W, focal = int(hwf[0]), int(hwf[1]), hwf[2]
    with torch.no_grad():
        data = data.to(device)
        B = data.shape[0]
        _, pose_ = inference_pose_regression(args, data, device, model)
        pose_nerf = fix_coord_supp(args, pose_.clone(), world_setup_dict, device=device)
        hist = torch.as_tensor(img_idx, dtype=torch.float32, device=device).reshape(B, -1)
        rays = [_engine.raygen(H, W, float(focal), pose_nerf[b, :3, :4].contiguous(), want_viewdirs=False) for b in range(B)]
        ro = torch.cat([r[0].reshape(-1, 3) for r in rays])
        rd = torch.cat([r[1].reshape(-1, 3) for r in rays])
        target = data.permute(0, 2, 3, 1).reshape(-1, 3)
        sel = torch.randperm(ro.shape[0], device=device)[:int(args.N_rand)]        # prepare_batch_render: random over all frames (:169-174)
        rows = hist[sel // (H * W)]
        rgb = render(H, W, focal, chunk=args.chunk, rays=(ro[sel].contiguous(), rd[sel].contiguous()), img_idx=rows,
                     **render_kwargs_test)[0]
        loss = torch.nn.functional.mse_loss(pose_.reshape(B, 12), torch.as_tensor(pose, device=device).reshape(B, 12).float())
        psnr = -10. * torch.log10(torch.mean((rgb - target[sel]) ** 2))
    return np.array([float(loss)]), np.array([float(psnr)])


def eval_on_epoch(args, data_loaders, model, feat_model, hwf, half_res, device, world_setup_dict, **render_kwargs_test):
    """Mean validation (pose loss, PSNR) over val_dl (:215-233)."""
    import numpy as np
    model.eval()
    losses, psnrs = [], []
    for data, pose, img_idx in data_loaders[1]:
        l, p = eval_on_batch(args, data, model, feat_model, pose, img_idx, hwf, half_res, device, world_setup_dict, **render_kwargs_test)
        losses.append(l.item())
        psnrs.append(p.item())
    return float(np.mean(losses)), float(np.mean(psnrs))


def train_feature_matching(args, model, feat_model, optimizer, i_split, hwf, near, far, device, early_stopping, images=None,
                           poses_train=None, train_dl=None, val_dl=None, test_dl=None, n_epoch=2001):
    """The DFNet_dm fine-tuning loop (:412-471): per epoch one pass of train_on_epoch over train_dl, a validation pass, the
    EarlyStopping callback on the validation loss (checkpoint-<epoch>-<val>.pt like the reference), and every i_eval epochs the
    median pose error over val_dl.  NeRF-H and the feature extractor are frozen; with torch.distributed initialised the training
    frames are sharded over the ranks (one all-reduce of the regressor's gradients per step, dist.allreduce_gradients), every rank
    runs the same validation pass, so the replicas stop at the same epoch."""
    from .feature_misc import get_error_in_q
    from .nerfw import create_nerf
    _, render_kwargs_test, start, _, _ = create_nerf(args)
    render_kwargs_test.update({'near': near, 'far': far})
    for q in feat_model.parameters():
        q.requires_grad_(False)
    world_setup_dict = {k: getattr(train_dl.dataset, k) for k in ('pose_scale', 'pose_scale2', 'move_all_cam_vec')}
    rank, world = ddist.rank_world()
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(train_dl.dataset, num_replicas=world, rank=rank, shuffle=True)
        train_dl = torch.utils.data.DataLoader(train_dl.dataset, batch_size=args.batch_size, sampler=sampler)
    data_loaders = [train_dl, val_dl, test_dl]
    half_res = True
    for epoch in range(n_epoch):
        if epoch and hasattr(model, "recommit"):
            model.recommit()             # fresh split-f16 weight scales for the re-packed regressor
        if sampler is not None:
            sampler.set_epoch(epoch)     # a fresh permutation per epoch, the same on every rank
        loss, psnr = train_on_epoch(args, data_loaders, model, feat_model, hwf, optimizer, half_res, device, world_setup_dict,
                                    **render_kwargs_test)
        val_loss, val_psnr = eval_on_epoch(args, data_loaders, model, feat_model, hwf, half_res, device, world_setup_dict,
                                           **render_kwargs_test)
        if world > 1:                    # the validation PSNR is over random rays: agree on one figure so every replica stops together
            t = torch.tensor([val_loss, val_psnr], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t)
            val_loss, val_psnr = (t / world).tolist()
        if rank == 0:
            print('At epoch {0:4d} : train loss: {1:.4f}, train psnr: {2:.4f}, val loss: {3:.4f}, val psnr: {4:.4f}'.format(
                epoch, loss, psnr, val_loss, val_psnr))
        early_stopping(val_loss, model, epoch=epoch, save_multiple=(not args.no_save_multiple), save_all=args.save_all_ckpt,
                       val_psnr=val_psnr)
        if early_stopping.early_stop:
            print("Early stopping")
            break
        if epoch % args.i_eval == 0 and rank == 0:
            get_error_in_q(args, val_dl, model, len(val_dl.dataset), device, batch_size=1)

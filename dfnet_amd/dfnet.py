"""Host-side mirror of /root/reference/script/feature/dfnet.py (classes DFNet, DFNet_s).

The modules below are *parameter containers* with the reference's state_dict layout
(`encoder.<k>.*`, `adaptation_layers.adapt_layer_<i>.{0,2,3}.*`, `fc_pose.*`), so the reference's
`checkpoint-*.pt` files load unchanged; `forward` has the reference's signature and return
convention and is evaluated by the HIP feature-extractor (dfn_dfnet_forward).  BatchNorm runs in
eval mode on this path (reference: train.py:123, utils.py:30-39).

Autograd: with grad enabled and an input that requires grad (DFNet_dm: feat_model(cat([data, rgb])) with
the rendered rgb attached to the pose, direct_feature_matching.py:350-376) the feature maps come back
attached to the graph and their backward is the HIP input-gradient path (dfn_dfnet_backward_input).  On
that feature path the module's own weights are treated as frozen.  The POSE path (return_pose=True,
return_feature=False) is differentiable w.r.t. its parameters instead: encoder convs + fc_pose, by the HIP
weight-gradient kernels (dfn_dfnet_backward_params) — the regressor DFNet_dm trains.

Training DFNet itself (run_feature.py:166-230): with grad enabled, return_feature=True, an input
without grad and parameters that require grad, both heads are differentiable w.r.t. every trained tensor (encoder,
fc_pose, adaptation convs, BatchNorm affine) through dfn_dfnet_forward_train / dfn_dfnet_backward_all_params.  The
BatchNorm layers follow their own train()/eval() flag exactly as nn.BatchNorm2d does: train() = statistics of the
batch (running statistics updated with momentum 0.1), eval() under model.train() = --freezeBN
(utils/utils.py:30-39).

torchvision's pretrained VGG16 weights (dfnet.py:90) are a download and unavailable offline: the
encoder is created with default Conv2d init; load a checkpoint for real use.
"""
import torch
import torch.nn as nn

from .engine import DfnetEngine
from .synthetic import VGG16_CFG


def _vgg16_features():
    layers, cin = [], 3
    for v in VGG16_CFG:
        if v == "M":
            layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
        else:
            layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    return nn.Sequential(*layers)


class AdaptLayers(nn.Module):
    """adapt_layer_i = Conv1x1(C_i -> 64), ReLU, Conv5x5(64 -> output_dim, pad 2), BatchNorm2d (dfnet.py:42-72)."""

    def __init__(self, channel_sizes, output_dim=128):
        super().__init__()
        for i, c in enumerate(channel_sizes):
            self.add_module("adapt_layer_{}".format(i), nn.Sequential(
                nn.Conv2d(c, 64, kernel_size=1, stride=1, padding=0), nn.ReLU(),
                nn.Conv2d(64, output_dim, kernel_size=5, stride=1, padding=2), nn.BatchNorm2d(output_dim)))


class _FeatureFn(torch.autograd.Function):
    """Feature pyramid of a frozen DFNet with d L/d x as its backward."""

    @staticmethod
    def forward(ctx, x, engine, single, upH, upW, feature_levels=None, grad_levels=None):
        # feature_levels: the caller's "only these pyramid levels are read" for THIS call (prunes the forward); grad_levels: "my loss
        # reads these levels only", which belongs to THIS forward's graph.  Both arrive as explicit arguments (module.forward pops the
        # engine's one-shot hints in every branch), so neither can leak into a later, unrelated forward.
        feats, _ = engine.forward(x.detach(), True, single, False, upH, upW, levels=feature_levels, zero_unread=False)
        ctx.save_for_backward(x.detach())
        ctx.cfg = (engine, single, grad_levels)
        return (feats,) if single else (feats[0], feats[1])

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        engine, single, levels = ctx.cfg
        if single:
            g = grads[0]
        else:  # siamese halves back into the batch order of x: [target half, render half]
            shape = next(t for t in grads if t is not None).shape
            g = torch.cat([t if t is not None else x.new_zeros(shape) for t in grads], 1)
        if levels is None:   # otherwise: which levels carry gradient at all (a scan of g and a host sync per level)
            levels = [t for t in range(g.shape[0]) if bool((g[t] != 0).any())]
        if not levels:
            return torch.zeros_like(x), None, None, None, None, None, None
        return engine.backward_input(x, g.contiguous(), levels=levels), None, None, None, None, None, None


class _PoseFn(torch.autograd.Function):
    """Pose regression with the parameter gradients of its path as backward (the regressor being trained in DFNet_dm)."""

    @staticmethod
    def forward(ctx, x, module, *params):
        E = module.engine(train=True)   # the pose path reads no BatchNorm-folded weights
        pose, ctx.tape = E.forward_pose_keep(x.detach())   # activations kept: the backward recomputes nothing
        ctx.tape_version = module._version()
        ctx.save_for_backward(x.detach())
        ctx.module = module
        return pose

    @staticmethod
    def backward(ctx, g_pose):
        (x,) = ctx.saved_tensors
        m = ctx.module
        E = m.engine(train=True)
        tape = ctx.tape if E.holds(ctx.tape) and ctx.tape_version == m._version() else None
        grads = E.backward_params(x, g_pose.contiguous(), tape=tape)
        ctx.tape = None
        # hand the ONLY reference of each gradient to autograd: AccumulateGrad then adopts the tensor as .grad instead of cloning
        # it (one device-to-device copy per parameter and step otherwise)
        out = tuple(grads.pop(k) for k in m._pose_param_names())
        del grads
        return (None, None) + out


class FeaturePyramid:
    """One stream's feature stack [L, B, 128, H, W] of a siamese TRAINING forward, not materialised: the levels stay at their own
    resolution in the engine's tape (DFNet.pyramid_features = True).  The triplet losses of feature_misc accept a pair of these and
    compute the very loss of the enlarged stacks from the low-resolution maps (csrc/dfnet_triplet_pyr.hip); `shape` is the shape
    the reference's tensor would have.  Anything else that wants real tensors: leave pyramid_features off."""

    def __init__(self, token, state, half, shape):
        self.token, self.state, self.half, self.shape = token, state, half, tuple(shape)

    def __repr__(self):
        return f"FeaturePyramid(shape={self.shape}, stream={self.half})"


class _PyramidState:
    """What the two FeaturePyramid halves of one forward share: engine, tape, geometry — and, once a triplet loss has been taken,
    its device state for the backward."""

    def __init__(self, engine, tape, B, H, W, upH, upW, feature_images):
        self.engine, self.tape, self.B, self.H, self.W, self.upH, self.upW = engine, tape, B, H, W, upH, upW
        self.feature_images = feature_images   # the leading frames that are the siamese pair (B: all)
        self.triplet = None     # (device state, f1_half)


class _TrainFn(torch.autograd.Function):
    """Both heads of DFNet with the gradients of every trained tensor as backward (training DFNet itself)."""

    @staticmethod
    def forward(ctx, x, module, bn_batch, return_pose, single, upH, upW, pyramid, *params):
        x = x.detach()
        E = module.engine(train=True, running_stats=not bn_batch)
        if pyramid:
            # the siamese stacks stay a pyramid in the tape; the two outputs are tokens through which d L / d (triplet loss) returns
            nf = int(pyramid)      # the leading frames that are the siamese pair (frames beyond them: encoder + pose head only)
            pose, stats, tape = E.forward_train_pyramid(x, return_pose, bn_batch, feature_images=nf)
            if bn_batch:
                module._update_running_stats(stats, (nf,) + tuple(x.shape[1:]))
            ctx.save_for_backward(x)
            ctx.tape, ctx.tape_version = tape, module._version()
            ctx.cfg = (module, bn_batch, single)
            ctx.pyr = module._pyramid_state = _PyramidState(E, tape, x.shape[0], x.shape[2], x.shape[3], upH, upW, nf)
            return torch.zeros((), device=x.device), torch.zeros((), device=x.device), pose
        ctx.pyr = None
        # with a graph being recorded the forward keeps its activations (the "tape") and the backward recomputes nothing
        keep = any(ctx.needs_input_grad)
        out = E.forward_train(x, True, return_pose, bn_batch, upH, upW, keep=keep)
        feats, pose, stats = out[:3]
        if bn_batch:
            module._update_running_stats(stats, x.shape)
        ctx.save_for_backward(x)
        ctx.tape = out[3] if keep else None
        ctx.tape_version = module._version() if keep else None
        ctx.cfg = (module, bn_batch, single)
        if single:
            return feats, None, pose
        half = x.shape[0] // 2   # the two streams: halves of ONE [L,2B,128,H,W] tensor (no copy)
        return feats[:, :half], feats[:, half:], pose

    @staticmethod
    def backward(ctx, g_a, g_b, g_pose):
        (x,) = ctx.saved_tensors
        m, bn_batch, single = ctx.cfg
        E = m.engine(train=True, running_stats=not bn_batch)
        if ctx.pyr is not None:
            st, g_loss = ctx.pyr, (g_a if g_a is not None else g_b)
            if not (E.holds(ctx.tape) and ctx.tape_version == m._version()):
                raise RuntimeError("DFNet: the pyramid forward's tape is gone (weights moved or too many forwards since): backward must follow it")
            if g_loss is None or st.triplet is None:
                grads = {} if g_pose is None else E.backward_params(x, g_pose.contiguous(), tape=ctx.tape)
            else:
                state, f1_half = st.triplet
                grads = E.backward_all_params_triplet(x, None if g_pose is None else g_pose.contiguous(), g_loss, state, f1_half, st.upH, st.upW,
                                                      bn_batch, ctx.tape, feature_images=st.feature_images)
            ctx.tape = ctx.pyr = st.tape = st.triplet = None
            out = tuple(grads.pop(k, None) for k in E.train_param_names(True))
            del grads
            return (None,) * 8 + out
        if single or (g_a is None and g_b is None):
            g_feats = g_a
        else:
            ref = g_a if g_a is not None else g_b
            g_a = torch.zeros_like(ref) if g_a is None else g_a
            g_b = torch.zeros_like(ref) if g_b is None else g_b
            L, hb = ref.shape[0], ref.shape[1]
            slab = ref[0, 0].numel()
            C, H, W = ref.shape[2:]
            dense = (2 * hb * slab, slab, H * W, W, 1)
            if g_a.stride() == dense and g_b.stride() == dense and g_b.data_ptr() == g_a.data_ptr() + 4 * hb * slab and \
                    g_a.dtype == torch.float32:
                # the halves of one tensor already (the fused triplet loss writes them that way)
                g_feats = torch.as_strided(g_a, (L, 2 * hb) + tuple(ref.shape[2:]), dense)
            else:
                g_feats = torch.cat([g_a, g_b], 1)
        if g_feats is None and g_pose is None:
            grads = {}
        elif g_feats is None:
            grads = E.backward_params(x, g_pose.contiguous())
        else:
            # the tape is valid only while this was the engine's latest kept forward and no weight moved since
            tape = ctx.tape if E.holds(ctx.tape) and ctx.tape_version == m._version() else None
            grads = E.backward_all_params(x, None if g_pose is None else g_pose.contiguous(), g_feats.contiguous(), bn_batch=bn_batch,
                                          tape=tape)
            ctx.tape = None
        out = tuple(grads.pop(k, None) for k in E.train_param_names(True))   # sole references: .grad adopts them without a copy
        del grads
        return (None,) * 8 + out


class _DFNetBase(nn.Module):
    tap_channels = (64, 256, 512)
    mean = [0.485, 0.456, 0.406]
    std = [0.229, 0.224, 0.225]

    def __init__(self, feat_dim=12, places365_model_path='', precision="f16x3"):
        super().__init__()
        self.encoder = _vgg16_features()
        self.hypercolumn_indices = [2, 14, 28][:len(self.tap_channels)]
        self.scales = [1, 4, 16][:len(self.tap_channels)]
        self.adaptation_layers = AdaptLayers(self.tap_channels, 128)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc_pose = nn.Linear(512, feat_dim)
        self.feat_dim = feat_dim
        self.precision = precision
        self._engine = None
        self._engine_version = None
        self._folded_stale = False
        self._bn_stats_stale = False   # the engine's copy of the BatchNorm running statistics lags the module's (see _update_running_stats)

    def _pose_param_names(self):
        names = []
        for idx in DfnetEngine.CONV_INDEX:
            names += [f"encoder.{idx}.weight", f"encoder.{idx}.bias"]
        return names + ["fc_pose.weight", "fc_pose.bias"]

    def _train_param_names(self):
        names = list(self._pose_param_names())
        for t in range(len(self.tap_channels)):
            pre = f"adaptation_layers.adapt_layer_{t}"
            names += [f"{pre}.0.weight", f"{pre}.0.bias", f"{pre}.2.weight", f"{pre}.2.bias", f"{pre}.3.weight", f"{pre}.3.bias"]
        return names

    def _version(self):
        return {k: (p.data_ptr(), p._version) for k, p in list(self.named_parameters()) + list(self.named_buffers())}

    def _refresh_names(self):
        """Tensors dfn_dfnet_refresh_train_params_device re-packs, in its order."""
        names = list(self._pose_param_names())
        for t in range(len(self.tap_channels)):
            pre = f"adaptation_layers.adapt_layer_{t}"
            names += [f"{pre}.0.weight", f"{pre}.0.bias", f"{pre}.2.weight", f"{pre}.2.bias", f"{pre}.3.weight", f"{pre}.3.bias",
                      f"{pre}.3.running_mean", f"{pre}.3.running_var"]
        return names

    def _commit_from_host(self):
        if self._engine is None:
            self._engine = DfnetEngine(len(self.tap_channels), self.feat_dim, self.precision)
        self._engine.load_numpy({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})
        self._folded_stale = False
        self._bn_stats_stale = False

    def recommit(self):
        """Re-pack every fragment from the host, re-deriving the split-f16 power-of-two weight scales.  The device re-packs of a
        training run keep the scales of the last host commit (max |w| -> 2^10): weights that grew ~64x since then would overflow the
        f16 hi halves, weights that shrank lose their lo halves.  The training loops call this once per epoch (~0.3 s)."""
        if self._engine is not None:
            self._commit_from_host()
            self._engine_version = self._version()

    def engine(self, train=False, running_stats=False):
        """The HIP engine holding the current weights, re-packed whenever a tensor of the module changed.
        running_stats=True (a train-mode forward with FROZEN BatchNorm): the engine's copy of the running statistics must be current too.
        train=True (the training forward / backward and the pose path, none of which read BatchNorm-folded weights): on
        the device.  Otherwise
        (inference: BatchNorm folded into the 5x5 convs): on the device when only the pose path's parameters moved
        (an optimizer step of DFNet_dm), from the host in every other case — including the first inference after
        training steps, whose device re-packs leave the folded weights stale."""
        ver = self._version()
        if self._engine is None:
            self._commit_from_host()
        elif ver != self._engine_version or (train and running_stats and self._bn_stats_stale):
            changed = {k for k in ver if ver[k] != self._engine_version.get(k)}
            sd = dict(list(self.named_parameters()) + list(self.named_buffers()))
            pose_names, names = self._pose_param_names(), self._refresh_names()
            on_gpu = all(sd[k].is_cuda for k in names)
            if on_gpu and changed and changed <= set(pose_names) and (train or not self._folded_stale) and \
                    not (train and running_stats and self._bn_stats_stale):   # (a stale BatchNorm block needs the full train-mode re-pack)
                # an optimizer step of DFNet_dm moves the regressor's 28 tensors only: re-pack those (the train-mode re-pack below
                # would redo the adaptation layers and copy the BatchNorm blocks too: 24 kernels and 12 copies per step)
                self._engine.refresh_pose_params_device([sd[k].detach() for k in pose_names])
            elif train and on_gpu and all(k in names or k.endswith("num_batches_tracked") for k in changed):
                self._engine.refresh_train_params_device([sd[k].detach() for k in names])
                self._bn_stats_stale = False
                self._folded_stale = self._folded_stale or not changed <= set(pose_names)   # folded: adaptation layers only
            else:
                self._commit_from_host()
        elif not train and self._folded_stale:
            self._commit_from_host()
        self._engine_version = ver
        return self._engine

    def _update_running_stats(self, stats, xshape):
        """nn.BatchNorm2d's train()-mode side effect: running statistics move towards the batch mean / UNBIASED batch
        variance with the module's momentum; stats [n_taps, 2, 128] = mean, biased variance."""
        B, _, H, W = xshape
        with torch.no_grad():
            bns = [getattr(self.adaptation_layers, "adapt_layer_{}".format(t))[3] for t in range(len(self.scales))]
            ns = [B * (H // scale) * (W // scale) for scale in self.scales]
            moms = [0.1 if bn.momentum is None else bn.momentum for bn in bns]
            if stats.is_cuda and len(set(moms)) == 1 and all(b.device == stats.device for bn in bns for b in (bn.running_mean, bn.running_var)):
                # the same update for all levels in five multi-tensor launches instead of five per level
                mom = moms[0]
                means, variances = [bn.running_mean for bn in bns], [bn.running_var for bn in bns]
                torch._foreach_mul_(means + variances, 1 - mom)
                torch._foreach_add_(means, list(stats[:, 0].unbind(0)), alpha=mom)
                torch._foreach_add_(variances, torch._foreach_mul(list(stats[:, 1].unbind(0)), [mom * n / max(n - 1, 1) for n in ns]))
                torch._foreach_add_([bn.num_batches_tracked for bn in bns], 1)
            else:
                for t, bn in enumerate(bns):
                    bn.running_mean.mul_(1 - moms[t]).add_(stats[t, 0].to(bn.running_mean.device), alpha=moms[t])
                    bn.running_var.mul_(1 - moms[t]).add_(stats[t, 1].to(bn.running_var.device), alpha=moms[t] * ns[t] / max(ns[t] - 1, 1))
                    bn.num_batches_tracked += 1
            for t, bn in enumerate(bns):
                if self._engine_version is not None:
                    # Only a frozen-BatchNorm forward and the folded inference weights read the engine's copy of these buffers: mark it
                    # stale instead of letting the version bump re-pack all 52 tensors before the step's next kernel call.
                    pre = "adaptation_layers.adapt_layer_{}.3.".format(t)
                    for name, buf in (("running_mean", bn.running_mean), ("running_var", bn.running_var),
                                      ("num_batches_tracked", bn.num_batches_tracked)):
                        self._engine_version[pre + name] = (buf.data_ptr(), buf._version)
            self._bn_stats_stale = True
            self._folded_stale = True

    def forward(self, x, return_feature=False, isSingleStream=False, return_pose=True, upsampleH=240, upsampleW=427, feature_images=None):
        """Same contract as dfnet.py:109-172: returns (feature_maps, predict) with feature_maps None,
        [stack] (single stream: 1 x [L,B,128,H,W]) or [stack_t, stack_r] (siamese: 2 x [L,B/2,128,H,W]).
        feature_images (an addition; pyramid_features training forward only): the leading frames of x that are the siamese pair —
        frames beyond them get a pose prediction and nothing else, i.e. run_feature.py:211-222's second forward
        `feat_model(rgb_perturb, False)` rides in the same encoder pass (the encoder has no batch-coupled layer)."""
        # One-shot hints a caller left on the engine for THIS call (direct_feature_matching._losses / _target_features): popped before
        # any branch — a call that routes through _TrainFn / _PoseFn, or raises, must not leave them for the next, unrelated forward
        # (which would silently get unwritten planes for the unlisted levels).
        feature_levels = grad_levels = None
        if self._engine is not None:
            feature_levels, self._engine.feature_levels_hint = getattr(self._engine, "feature_levels_hint", None), None
            grad_levels, self._engine.grad_levels_hint = getattr(self._engine, "grad_levels_hint", None), None
        bn_batch = self.adaptation_layers.adapt_layer_0[3].training
        wants_grad = torch.is_grad_enabled() and not x.requires_grad and any(p.requires_grad for p in self.parameters())
        if return_feature and (wants_grad or bn_batch) and not x.requires_grad:
            # training DFNet itself: unfolded adaptation layers, BatchNorm by its own mode, every parameter gradient
            sd = dict(self.named_parameters())
            names = self._train_param_names()
            pyramid = bool(getattr(self, "pyramid_features", False)) and not isSingleStream and torch.is_grad_enabled()
            if feature_images is not None and not pyramid:
                raise NotImplementedError("feature_images needs the pyramid training forward (model.pyramid_features = True, siamese, grad enabled)")
            nf = x.shape[0] if feature_images is None else int(feature_images)
            fa, fb, pose = _TrainFn.apply(x, self, bool(bn_batch), bool(return_pose), bool(isSingleStream), int(upsampleH),
                                          int(upsampleW), nf if pyramid else 0, *[sd[k] for k in names])
            if pyramid:   # the reference's two stacks [L, B/2, 128, H, W], as a pyramid (feature_misc's triplet losses take these)
                shape = (len(self.tap_channels), nf // 2, 128, int(upsampleH), int(upsampleW))
                st = self._pyramid_state
                return [FeaturePyramid(fa, st, 0, shape), FeaturePyramid(fb, st, 1, shape)], pose
            return ([fa] if isSingleStream else [fa, fb]), pose
        if wants_grad and return_pose and not return_feature:
            # training the regressor (DFNet_dm): parameter gradients of the pose path come from the HIP wgrad kernels
            sd = dict(self.named_parameters())
            return None, _PoseFn.apply(x, self, *[sd[k] for k in self._pose_param_names()])
        if torch.is_grad_enabled() and x.requires_grad:
            if return_pose or not return_feature:
                raise NotImplementedError("autograd w.r.t. the INPUT through the pose head is not built; the feature path "
                                          "(return_feature=True, return_pose=False) is, and so are the pose path's "
                                          "parameter gradients (input without grad)")
            feats = _FeatureFn.apply(x, self.engine(), bool(isSingleStream), int(upsampleH), int(upsampleW), feature_levels, grad_levels)
            return list(feats), None
        # pose-only inference reads no BatchNorm-folded weights: the device re-pack of a training step is enough for it
        E = self.engine(train=not return_feature)
        levels = feature_levels if (return_feature and not return_pose) else None   # the caller's "only these levels are read"
        feats, pose = E.forward(x, return_feature, isSingleStream, return_pose, upsampleH, upsampleW, levels=levels, zero_unread=False)
        if feats is not None:
            feats = [feats] if isSingleStream else [feats[0], feats[1]]
        return feats, pose


class DFNet(_DFNetBase):
    ''' hypercolumns conv1_2, conv3_3, conv5_3 (dfnet.py:74-107) '''
    tap_channels = (64, 256, 512)


class DFNet_s(_DFNetBase):
    ''' conv1_2 only (dfnet.py:174-207) '''
    tap_channels = (64,)

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
    def forward(ctx, x, engine, single, upH, upW):
        feats, _ = engine.forward(x.detach(), True, single, False, upH, upW)
        ctx.save_for_backward(x.detach())
        ctx.cfg = (engine, single)
        return (feats,) if single else (feats[0], feats[1])

    @staticmethod
    def backward(ctx, *grads):
        (x,) = ctx.saved_tensors
        engine, single = ctx.cfg
        if single:
            g = grads[0]
        else:  # siamese halves back into the batch order of x: [target half, render half]
            shape = next(t for t in grads if t is not None).shape
            g = torch.cat([t if t is not None else x.new_zeros(shape) for t in grads], 1)
        levels = [t for t in range(g.shape[0]) if bool((g[t] != 0).any())]
        if not levels:
            return torch.zeros_like(x), None, None, None, None
        return engine.backward_input(x, g.contiguous(), levels=levels), None, None, None, None


class _PoseFn(torch.autograd.Function):
    """Pose regression with the parameter gradients of its path as backward (the regressor being trained in DFNet_dm)."""

    @staticmethod
    def forward(ctx, x, module, *params):
        _, pose = module.engine().forward(x.detach(), False, True, True)
        ctx.save_for_backward(x.detach())
        ctx.module = module
        return pose

    @staticmethod
    def backward(ctx, g_pose):
        (x,) = ctx.saved_tensors
        m = ctx.module
        grads = m.engine().backward_params(x, g_pose.contiguous())
        return (None, None) + tuple(grads[k] for k in m._pose_param_names())


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

    def _pose_param_names(self):
        names = []
        for idx in DfnetEngine.CONV_INDEX:
            names += [f"encoder.{idx}.weight", f"encoder.{idx}.bias"]
        return names + ["fc_pose.weight", "fc_pose.bias"]

    def _version(self):
        return {k: (p.data_ptr(), p._version) for k, p in list(self.named_parameters()) + list(self.named_buffers())}

    def engine(self):
        """The HIP engine holding the current weights.  Re-packed whenever a parameter tensor changed: on the device
        when only the pose path's parameters moved and they live on the GPU (an optimizer step of DFNet_dm), from the
        host otherwise."""
        ver = self._version()
        if self._engine is None or ver != self._engine_version:
            pose_names = self._pose_param_names()
            changed = None if self._engine_version is None else {k for k in ver if ver[k] != self._engine_version.get(k)}
            params = dict(self.named_parameters())
            if self._engine is not None and changed is not None and changed <= set(pose_names) and \
                    all(params[k].is_cuda for k in pose_names) and len(self.tap_channels) == 3:
                self._engine.refresh_pose_params_device([params[k].detach() for k in pose_names])
            else:
                if self._engine is None:
                    self._engine = DfnetEngine(len(self.tap_channels), self.feat_dim, self.precision)
                self._engine.load_numpy({k: v.detach().cpu().numpy() for k, v in self.state_dict().items()})
            self._engine_version = ver
        return self._engine

    def forward(self, x, return_feature=False, isSingleStream=False, return_pose=True, upsampleH=240, upsampleW=427):
        """Same contract as dfnet.py:109-172: returns (feature_maps, predict) with feature_maps None,
        [stack] (single stream: 1 x [L,B,128,H,W]) or [stack_t, stack_r] (siamese: 2 x [L,B/2,128,H,W])."""
        if torch.is_grad_enabled() and return_pose and not return_feature and not x.requires_grad and \
                any(p.requires_grad for p in self.parameters()):
            # training the regressor (DFNet_dm): parameter gradients of the pose path come from the HIP wgrad kernels
            sd = dict(self.named_parameters())
            return None, _PoseFn.apply(x, self, *[sd[k] for k in self._pose_param_names()])
        if torch.is_grad_enabled() and x.requires_grad:
            if return_pose or not return_feature:
                raise NotImplementedError("autograd w.r.t. the INPUT through the pose head is not built; the feature path "
                                          "(return_feature=True, return_pose=False) is, and so are the pose path's "
                                          "parameter gradients (input without grad)")
            feats = _FeatureFn.apply(x, self.engine(), bool(isSingleStream), int(upsampleH), int(upsampleW))
            return list(feats), None
        feats, pose = self.engine().forward(x, return_feature, isSingleStream, return_pose, upsampleH, upsampleW)
        if feats is not None:
            feats = [feats] if isSingleStream else [feats[0], feats[1]]
        return feats, pose


class DFNet(_DFNetBase):
    ''' hypercolumns conv1_2, conv3_3, conv5_3 (dfnet.py:74-107) '''
    tap_channels = (64, 256, 512)


class DFNet_s(_DFNetBase):
    ''' conv1_2 only (dfnet.py:174-207) '''
    tap_channels = (64,)

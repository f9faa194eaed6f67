"""CPU oracle for the DFNet feature-extractor forward.  TEST INFRASTRUCTURE ONLY.

A from-scratch CPU (torch fp32, functional) restatement of DFNet.forward
(/root/reference/script/feature/dfnet.py:109-172) and AdaptLayers (:42-72).  Only
`tests/`, `__graft_entry__.smoke()` and bench.py's cpu_baseline may import it.

The VGG16 `features` stack the reference takes from torchvision==0.10.0
(requirements.txt:96; call site dfnet.py:90-92) is third-party code absent from
/root/reference; its published architecture (cfg "D") is restated here: 13 x
[Conv3x3 stride 1 pad 1 + ReLU] with MaxPool(2,2) after convs 2, 4, 7, 10, 13.  No
reference test pins results at that boundary ("parity unpinned" for the third-party
stack); the code above it is pinned by tests/golden/g8_* captured from the reference's
DFNet.forward running on that restated stack.

Parameters: dict {state_dict key: tensor} with the reference's names
(`encoder.{k}.weight`, `adaptation_layers.adapt_layer_{i}.{0,2,3}.*`, `fc_pose.*`).
"""
import torch
import torch.nn.functional as F

VGG16_CFG = (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M")
MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
TAP_CONV_INDEX = {"conv1_2": 2, "conv3_3": 14, "conv5_3": 28}  # positions in the 31-module encoder


def encoder_taps(p, x, taps=(2, 14, 28), stop_after_last_tap=True):
    """Run the VGG stack; return (pre-ReLU conv outputs at `taps`, final tensor or None).

    Taps are cloned BEFORE the in-place ReLU (dfnet.py:126-131); with return_pose=False the
    loop breaks right after the last tap (dfnet.py:133-136)."""
    feats, idx = [], 0
    for v in VGG16_CFG:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            idx += 1
            continue
        x = F.conv2d(x, p[f"encoder.{idx}.weight"], p[f"encoder.{idx}.bias"], padding=1)
        if idx in taps:
            feats.append(x.clone())
            if idx == taps[-1] and stop_after_last_tap:
                return feats, None
        x = torch.relu(x)
        idx += 2
    return feats, x


def adapt(p, i, f, eps=1e-5, bn_stats=None):
    """Conv1x1 -> ReLU -> Conv5x5(pad 2) -> BatchNorm2d (dfnet.py:57-62).  bn_stats None: eval mode (running
    statistics).  A list: train() mode — batch statistics (nn.BatchNorm2d.forward with training=True); the batch mean
    and BIASED variance of this level are appended to it (the module then moves running_mean / running_var by
    momentum 0.1 towards mean / unbiased variance)."""
    pre = f"adaptation_layers.adapt_layer_{i}"
    f = torch.relu(F.conv2d(f, p[pre + ".0.weight"], p[pre + ".0.bias"]))
    f = F.conv2d(f, p[pre + ".2.weight"], p[pre + ".2.bias"], padding=2)
    if bn_stats is not None:
        bn_stats.append((f.detach().mean((0, 2, 3)), f.detach().var((0, 2, 3), unbiased=False)))
        return F.batch_norm(f, None, None, p[pre + ".3.weight"], p[pre + ".3.bias"], training=True, eps=eps)
    return F.batch_norm(f, p[pre + ".3.running_mean"], p[pre + ".3.running_var"],
                        p[pre + ".3.weight"], p[pre + ".3.bias"], training=False, eps=eps)


def upsample(f, H, W):
    """nn.UpsamplingBilinear2d(size) == bilinear, align_corners=True (dfnet.py:145)."""
    return F.interpolate(f, size=(H, W), mode="bilinear", align_corners=True)


def dfnet_forward(p, x, return_feature=False, isSingleStream=False, return_pose=True,
                  upsampleH=240, upsampleW=427, taps=(2, 14, 28), bn_stats=None):
    """(feature_maps | None, predict | None) exactly as DFNet.forward (dfnet.py:109-172); bn_stats: see adapt()."""
    mean = x.new_tensor(MEAN)[:, None, None]
    std = x.new_tensor(STD)[:, None, None]
    x = (x - mean) / std
    feats, last = encoder_taps(p, x, taps, stop_after_last_tap=not return_pose)
    maps = None
    if return_feature:
        ad = [adapt(p, i, f, bn_stats=bn_stats) for i, f in enumerate(feats)]
        if isSingleStream:
            maps = [torch.stack([upsample(f, upsampleH, upsampleW) for f in ad])]
        else:
            half = ad[0].shape[0] // 2
            maps = [torch.stack([upsample(f[:half], upsampleH, upsampleW) for f in ad]),
                    torch.stack([upsample(f[half:], upsampleH, upsampleW) for f in ad])]
    if not return_pose:
        return maps, None
    pooled = last.mean((2, 3))  # AdaptiveAvgPool2d(1) after relu5_3 + pool5
    return maps, F.linear(pooled, p["fc_pose.weight"], p["fc_pose.bias"])


def triplet_loss_cases(f1, f2):
    """The four full-tensor MSEs of the in-triplet hard-negative mining (misc.py:414-421): (f1, roll f2), (f2, roll f1),
    (f1, roll f1), (f2, roll f2); roll = torch.roll(., shifts=1, dims=1) — the previous image of the batch."""
    a_neg, neg = torch.roll(f1, 1, 1), torch.roll(f2, 1, 1)
    mse = lambda u, v: ((u - v) ** 2).mean()
    return torch.stack([mse(f1, neg), mse(f2, a_neg), mse(f1, a_neg), mse(f2, neg)])


def triplet_loss(f1, f2, margin=1.0, mining=2, eps=1e-6):
    """Triplet losses on feature stacks [lvl,B,C,H,W] written out from their definition (misc.py:355-435 call
    nn.TripletMarginLoss(margin, p=2, eps=1e-6, reduction='mean'), whose pairwise distance ||x - y + eps||_2 runs over
    the LAST axis): mining 0 = triplet_loss, 1 = ..._hard_negative_mining, 2 = ..._hard_negative_mining_plus."""
    a_neg, neg = torch.roll(f1, 1, 1), torch.roll(f2, 1, 1)
    with torch.no_grad():
        c = triplet_loss_cases(f1, f2)
        case = 0 if mining == 0 else ((0 if c[0] < c[1] else 1) if mining == 1 else int(torch.argmin(c)))
    x, y, z = [(f1, f2, neg), (f2, f1, a_neg), (f1, f2, a_neg), (f2, f1, neg)][case]
    d = lambda u, v: torch.sqrt(((u - v + eps) ** 2).sum(-1))
    return torch.clamp_min(d(x, y) - d(x, z) + margin, 0).mean(), case

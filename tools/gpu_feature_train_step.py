#!/usr/bin/env python3
"""One optimisation step of DFNet's own training (run_feature.py:166-230 with config_dfnet.txt: triplet loss, random
view synthesis): siamese forward on [target, render] (2B frames), pose forward on B synthesised views, triplet + pose
losses, backward of every parameter, Adam, device re-pack.  All DFNet arithmetic on the HIP path; losses/optimizer
are torch tensor ops.  Usage: gpu_feature_train_step.py [B] [iters] [H] [W] [freezebn].  Prints one JSON line."""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import synthetic as syn
from dfnet_amd import optim
from dfnet_amd.dfnet import DFNet
from dfnet_amd.feature_misc import PoseLoss, freeze_bn_layer, freeze_bn_layer_train, triplet_loss_hard_negative_mining_plus

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H = int(sys.argv[3]) if len(sys.argv) > 3 else 240
W = int(sys.argv[4]) if len(sys.argv) > 4 else 427
frozen = len(sys.argv) > 5 and sys.argv[5] == "freezebn"
m = DFNet()
m.load_state_dict({k: torch.from_numpy(v) for k, v in syn.dfnet_weights(3).items()}, strict=False)
if frozen:
    m = freeze_bn_layer(m)
m.to(dev).train()
if frozen:
    m = freeze_bn_layer_train(m)
# what script/run_feature.py ships under --tripletloss: the feature stacks stay a low-resolution pyramid (dfnet.FeaturePyramid);
# FT_STACKS=1 = the materialised stacks + stack triplet kernels (the round-5 form), for A/B
m.pyramid_features = not os.environ.get("FT_STACKS")
opt = (torch.optim.Adam if os.environ.get('DFN_TORCH_ADAM') == '1' else optim.Adam)(m.parameters(), lr=1e-6)
g = torch.Generator().manual_seed(1)
target, rgb, virt = (torch.rand(B, 3, H, W, generator=g).to(dev) for _ in range(3))
pose = torch.stack([torch.from_numpy(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(B)]).to(dev)
pose2 = torch.cat([pose, pose])
parts = {}


LOOP = bool(os.environ.get("FT_LOOP"))   # the shipped epoch loop: no waits inside or between the steps (losses stay on the device)


def step(update=True, rvs=True):
    t = [time.perf_counter()]
    def mark():
        if not LOOP:
            torch.cuda.synchronize()
        t.append(time.perf_counter())
    one_pass = rvs and m.pyramid_features and not os.environ.get("FT_TWO_PASS")   # script/run_feature.py: the synthesised views ride in the siamese pass
    if one_pass:
        feats, pall = m(torch.cat([target, rgb, virt]), True, upsampleH=H, upsampleW=W, feature_images=2 * B); mark()
        pred, vp = pall[:2 * B], pall[2 * B:]
    else:
        feats, pred = m(torch.cat([target, rgb]), True, upsampleH=H, upsampleW=W); mark()
    loss = PoseLoss(None, pred, pose2, dev) + triplet_loss_hard_negative_mining_plus(feats[1], feats[0], margin=1.0)
    if rvs:
        if not one_pass:
            _, vp = m(virt, False)
        loss = loss + PoseLoss(None, vp, pose, dev)
    mark()
    loss.backward(); mark()
    if update:
        opt.step()
    opt.zero_grad(); mark()
    for k, a, b in (("forward_siamese", 0, 1), ("losses_and_rvs_forward", 1, 2), ("backward", 2, 3), ("adam", 3, 4)):
        parts[k] = parts.get(k, 0.0) + (t[b + 0] - t[a]) * 1e3
    return loss.detach() if LOOP else float(loss)


def timed(fn):
    fn(); torch.cuda.synchronize()
    parts.clear()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out

full_ms, loss = timed(step)
loss = float(loss)
host_ms = None
if LOOP:   # host time to ENQUEUE one step on an idle device (nothing to wait for): well under step_ms = the loop is bound by the device
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); host_ms = (time.perf_counter() - t0) * 1e3; torch.cuda.synchronize()
breakdown = {k: v / iters for k, v in parts.items()}
print(json.dumps({"workload": f"DFNet training step (run_feature.py, triplet loss + RVS), featurenet_batch_size {B} -> {2 * B} siamese + {B} "
                              f"synthesised frames at {H}x{W}, BatchNorm {'frozen' if frozen else 'batch statistics'}",
                  "step_ms": full_ms, "epoch_loop_without_per_step_waits": LOOP, "host_enqueue_ms_of_one_step_on_an_idle_device": host_ms, "frames_per_s": 3 * B / full_ms * 1e3, "breakdown_ms_with_syncs": breakdown, "loss": loss,
                  "precision": "split-f16 (f16x3) forward, data-gradient AND weight-gradient products; fp32 accumulate",
                  "triplet_loss": "from the low-resolution pyramid (no enlarged stacks)" if m.pyramid_features else "on materialised [3,B,128,H,W] stacks", "peak_mem_GB": torch.cuda.max_memory_allocated() / 2 ** 30}))

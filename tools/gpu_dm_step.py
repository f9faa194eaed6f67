#!/usr/bin/env python3
"""DFNet_dm step timing (BASELINE configs[4] shape per GPU: batch 4, 240x320 frames, render 60x80 at 64+128 then
bicubic x4, level-0 cosine feature loss + photometric + pose terms): forward losses and the backward down to the
predicted pose, all arithmetic on the HIP path.  Prints one JSON line."""
import json, os, sys, time
from types import SimpleNamespace
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dfnet_amd import engine as eng, synthetic as syn
from dfnet_amd import optim
from dfnet_amd.dfnet import DFNet
from dfnet_amd.direct_feature_matching import matching_step_forward, matching_step_grad, train_on_batch, train_on_batch_device
from dfnet_amd.nerfw import HipQuery

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W, focal = 240, 320, 585.0 / 2
sd = {k: torch.from_numpy(v) for k, v in syn.dfnet_weights(3).items()}
model, feat_model = DFNet().to(dev).eval(), DFNet().to(dev).eval()
model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=False)
feat_model.load_state_dict({k: v.to(dev) for k, v in sd.items()}, strict=False)
for q in feat_model.parameters():
    q.requires_grad_(False)
cw, fw, ea, et = syn.nerfh_weights(0)
E = eng.NerfHEngine(precision="f16").load_numpy(cw, fw, ea, et)
kw = dict(network_query_fn=HipQuery(E), perturb=False, N_importance=128, N_samples=64, use_viewdirs=True,
          white_bkgd=False, raw_noise_std=0., test_time=True, ndc=False, lindisp=False, near=0., far=2.5)
setup = dict(pose_scale=1.0, pose_scale2=1.0, move_all_cam_vec=[0., 0., 1.0])
args = SimpleNamespace(svd_reg=True, chunk=32768, feature_matching_lvl=[0], per_channel=False, combine_loss=True,
                       combine_loss_w=[0.3, 0.2, 1.0])
data = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
gt = torch.stack([torch.from_numpy(syn.orbit_pose(k, 8))[:3, :4].reshape(12) for k in range(B)])
hist = torch.from_numpy(syn.HIST_IDX).repeat(B, 1)


def timed(fn):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3, out


from dfnet_amd import rendering
import dfnet_amd.direct_feature_matching as _dfm
if os.environ.get("DM_NO_OVERLAP"):   # A/B aids: the target features in line / every pyramid level computed
    _dfm.OVERLAP_TARGET_FEATURES = False
if os.environ.get("DM_ALL_LEVELS"):
    _dfm.PRUNE_FEATURE_LEVELS = False
if os.environ.get("DM_ONLY"):   # profiling aid: only the full optimisation step (rocprofv3 --stats then shows one step's kernels x iters)
    opt = (torch.optim.Adam if os.environ.get('DFN_TORCH_ADAM') == '1' else optim.Adam)(model.parameters(), lr=1e-7)
    if os.environ.get("DM_TRACE"):   # where the memcpys of a STEADY-STATE step come from (three warm steps first)
        import collections
        from torch.profiler import profile, ProfilerActivity
        step = lambda: train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], opt, True, dev, setup, **kw)
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        c = collections.Counter()
        for e in prof.events():
            if "emcpy" in e.name and e.cpu_parent is not None or e.name.startswith("hipMemcpy"):
                names, p = [], e.cpu_parent
                while p is not None and len(names) < 5:
                    names.append(p.name); p = p.cpu_parent
                st = [x for x in (e.stack or []) if "dfnet_amd" in x or "tools/" in x][:2]
                c[(e.name[:28], " < ".join(names)[:150], " | ".join(st)[:200])] += 1
        for k, v in c.most_common(40):
            print(v / 2, k)
        # every top-level aten op of a step (each is one or more launches) with the package frame that issued it
        c2 = collections.Counter()
        for e in prof.events():
            if not e.name.startswith("aten::"):
                continue
            p = e.cpu_parent
            if p is not None and (p.name.startswith("aten::") or "autograd" in p.name.lower() and False):
                continue
            st = [x for x in (e.stack or []) if "dfnet_amd" in x or "tools/" in x or "optim" in x][:1]
            c2[(e.name, (st[0] if st else (p.name if p is not None else "-"))[-110:])] += 1
        print("---- top-level aten ops per step")
        for k, v in c2.most_common(70):
            print(v / 2, k)
        sys.exit(0)
    fn = train_on_batch if os.environ.get("DM_HOST_FLOATS") else train_on_batch_device   # the epoch loop keeps the losses on the device
    ms, _ = timed(lambda: fn(args, data, model, feat_model, gt, hist, [H, W, focal], opt, True, dev, setup, **kw))
    print(json.dumps({"full_step_ms": ms, "iters_profiled": iters + 1, "host_floats_per_step": bool(os.environ.get("DM_HOST_FLOATS"))}))
    sys.exit(0)
results = {}
for mode, fp in (("fp32 forward state", "f32"), ("f16 forward state", None)):
    rendering.GRAD_FORWARD_PRECISION = fp
    ms, o = timed(lambda: matching_step_grad(args, data, model, feat_model, gt, hist, [H, W, focal], True, dev, setup, **kw))
    results[mode] = {"forward_backward_to_pose_ms": ms, "grad_pose": o["grad_pose"].cpu()}
ga, gb = results["fp32 forward state"]["grad_pose"], results["f16 forward state"]["grad_pose"]
mixed_rel = float((ga - gb).abs().max() / ga.abs().max())
rendering.GRAD_FORWARD_PRECISION = None
for gp in ("f32", "f16x3"):
    rendering.GRAD_PRECISION = gp
    ms, o = timed(lambda: matching_step_grad(args, data, model, feat_model, gt, hist, [H, W, focal], True, dev, setup, **kw))
    results["grad kernel " + gp] = {"ms": ms, "rel": float((o["grad_pose"].cpu() - ga).abs().max() / ga.abs().max())}
fwd_ms, _ = timed(lambda: matching_step_forward(args, data, model, feat_model, gt, hist, [H, W, focal], True, dev, setup, **kw))
step_ms, out = timed(lambda: matching_step_grad(args, data, model, feat_model, gt, hist, [H, W, focal], True, dev, setup, **kw))
# the full optimisation step: + regressor weight gradients (HIP) + Adam (torch) + device-side re-pack of the updated weights
opt = (torch.optim.Adam if os.environ.get('DFN_TORCH_ADAM') == '1' else optim.Adam)(model.parameters(), lr=1e-7)
class NoStep:   # gradients only: isolates the weight-gradient kernels from the optimizer / re-pack cost
    def step(self): pass
    def zero_grad(self):
        for q in model.parameters(): q.grad = None
wg_ms, _ = timed(lambda: train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], NoStep(), True, dev, setup, **kw))
full_ms, _ = timed(lambda: train_on_batch_device(args, data, model, feat_model, gt, hist, [H, W, focal], opt, True, dev, setup, **kw))
host_ms, _ = timed(lambda: train_on_batch(args, data, model, feat_model, gt, hist, [H, W, focal], opt, True, dev, setup, **kw))
print(json.dumps({"workload": f"DFNet_dm step, batch {B}, 240x320, render 60x80 @64+128 + bicubic x4, level-0 feature loss",
                  "forward_ms": fwd_ms, "forward_backward_to_pose_ms": step_ms,
                  "forward_backward_all_gradients_ms": wg_ms, "full_step_with_adam_and_device_repack_ms": full_ms,
                  "full_step_returning_host_floats_ms": host_ms, "ms_per_frame": step_ms / B,
                  "loss": float(out["loss"]), "grad_pose_absmax": float(out["grad_pose"].abs().max()),
                  "grad_kernel_modes": {k: v for k, v in results.items() if k.startswith("grad kernel")},
                  "all_fp32_tracked_forward_ms": results["fp32 forward state"]["forward_backward_to_pose_ms"],
                  "grad_pose_rel_diff_vs_all_fp32": mixed_rel,
                  "render_precision": "forward f16 (tracked and untracked); MLP gradient kernel f16x3 forward recompute + fp32 gradient chain",
                  "dfnet_precision": "f16x3 forward / f32 gradient convs"}))

"""Host-side mirror of /root/reference/script/models/rendering.py: same function names, arguments
and return shapes (`render`, `render_path`, `render_test`), evaluated by the HIP library.

What the reference does per ray chunk in Python (`batchify_rays` -> `render_rays` -> `netchunk`
loops, rendering.py:245-351) is one call into libdfnet_hip.so here; `chunk`/`netchunk` are accepted
and ignored (tiling is internal).  `render_path`'s serial frame loop (rendering.py:420) becomes a
frame-sharded loop with one gather when torch.distributed is initialised (dfnet_amd/dist.py).

Test-time kwargs (`render_kwargs_test`: perturb=0, raw_noise_std=0, test_time=True) run on the packed MFMA engine;
training kwargs (`render_kwargs_train`: test_time=False, perturb, raw_noise_std) run on the exact-fp32 training kernels and
return tensors attached to autograd with the reference's extras (dfnet_amd/nerf_train.py).  `lindisp` works everywhere; `ndc`
and `c2w_staticcam` at test time without autograd (their only use in the reference: LLFF-style visualisation).  `white_bkgd=True`
raises: it is not a working option of this path in the reference either (rendering.py:295 passes it to the coarse compositor as
`output_transient`, which fails with a TypeError at test time and mis-slices the 4-channel coarse output in training).

Autograd: when grad is enabled and `c2w` / `rays` require grad (the DFNet_dm step,
feature/direct_feature_matching.py:340-376), `rgb_map` is returned attached to the graph through
torch.autograd.Function wrappers: the forward is composed from the stage entry points and keeps (z_fine, raw);
the backward runs the HIP gradient kernels on that saved state (compositing backward, fused fine-MLP input
gradient, ray / pose reductions) — the stateless dfn_render_image_backward / dfn_render_rays_backward compute
the same thing with an internal re-render.  disp_map / acc_map are returned detached (the reference's losses use
rgb only).  The MLP gradient kernel defaults to the exact-fp32 MFMA path (`GRAD_PRECISION`), see
tests/test_gpu_grad.py.
"""
import os
import time

import numpy as np
import torch

from . import dist as ddist
from .nerfw import to8b
from .ray_utils import get_rays, ndc_rays  # noqa: F401  (re-exported like the reference's `from models.ray_utils import *`)

DEBUG = False


def _engine_of(kwargs):
    q = kwargs.get('network_query_fn')
    eng = getattr(q, 'engine', None)
    if eng is None:
        raise TypeError("render(): render kwargs must come from dfnet_amd.nerfw.create_nerf "
                        "(network_query_fn carries the HIP engine)")
    return eng


GRAD_PRECISION = "f16x3"  # MLP gradient kernel: forward recompute AND gradient chain in split-f16 (per-point power-of-two
                          # gradient scale, fp32-grade: parity 6e-7 vs autograd); "f32": all fp32 MFMA
# Arithmetic of the tracked forward, whose (z_fine, raw) the backward starts from.  None = the engine's own
# precision (what the user renders with, f16 by default): outputs identical to the untracked render, and in the
# DFNet_dm step a pose gradient within 7e-6 of the all-fp32 one (tools/gpu_dm_step.py) at 43 instead of 55 ms per
# step.  "f32" (or "f16x3", 2.7x faster) makes the whole tracked path fp32-grade — needed only where the loss makes d c2w a badly conditioned
# signed sum (tests/test_gpu_grad.py::test_render_autograd_drop_in: random per-pixel weights on a 12x16 image,
# where the f16 forward's 1e-4 error is amplified to 1e-2).
GRAD_FORWARD_PRECISION = None   # with GRAD_TWO_PASS (below) this governs the COARSE net only: the fine net is split-f16 then
# Two-pass gradient (default with the split-f16 gradient kernel): the tracked forward runs the fine net in split-f16 and
# records its ReLU signs (dfn_mlp_fine_saving); the backward then starts from (raw, masks) and recomputes NO forward
# (dfn_mlp_fine_backward_saved) — the gradient kernel was half forward recompute.  The tracked render is fp32-grade as
# a side effect.  False: the one-pass kernel (forward recompute inside the backward).
GRAD_TWO_PASS = True


def _two_pass():
    return GRAD_TWO_PASS and GRAD_PRECISION == "f16x3"


def _saving_forward(eng, o, d, v, hist, Nc, Ni, near, far):
    """(rgb, disp, acc, saved tensors for the backward)."""
    if eng.width != 128:
        # the register-resident gradient kernels (and their saved state) are netwidth 128: other widths render as usual and the
        # backward is the stateless generic-width gradient, which recomputes the forward layer by layer in exact fp32
        rgb, disp, acc, _ = eng.render_rays(o, d, hist, Nc, Ni, near, far)
        return rgb, disp, acc, (o, d, v, hist, torch.tensor([Nc, Ni, near, far, float(eng.lindisp)], dtype=torch.float64))
    if _two_pass():
        rgb, disp, acc, z, raw, masks = eng.render_rays_saving(o, d, v, hist, Nc, Ni, near, far, precision=GRAD_FORWARD_PRECISION,
                                                             with_masks=True)
        return rgb, disp, acc, (o, d, v, hist, z, raw, masks)
    rgb, disp, acc, z, raw = eng.render_rays_saving(o, d, v, hist, Nc, Ni, near, far, precision=GRAD_FORWARD_PRECISION)
    return rgb, disp, acc, (o, d, v, hist, z, raw)


def _saved_backward(eng, saved, g_rgb, g_raw=None):
    if g_rgb is None:
        g_rgb = torch.zeros(saved[0].shape[0], 3, device=saved[0].device)
    if len(saved) == 5:   # generic width: (rays, viewdirs = d / |d|, histograms, [Nc, Ni, near, far])
        o, d, _, hist, cfg = saved
        Nc, Ni, near, far, lindisp = cfg.tolist()
        eng.set_render_options(lindisp=bool(lindisp))
        return eng.render_rays_backward(o, d, hist, int(Nc), int(Ni), near, far, g_rgb.contiguous(), precision="generic")
    o, d, v, hist, z, raw = saved[:6]
    masks = saved[6] if len(saved) > 6 else None
    return eng.backward_from_saved(o, d, v, hist, z, raw, g_rgb.contiguous(), True, precision=GRAD_PRECISION, masks=masks, grad_raw=g_raw)


class _RenderImageFn(torch.autograd.Function):
    """rgb/disp/acc = render(c2w); backward: d L/d c2w from d L/d rgb.  The forward is composed from the stage
    entry points and keeps (rays, z_fine, raw), so backward differentiates from the saved state instead of
    re-rendering (dfn_render_image_backward is the stateless equivalent)."""

    @staticmethod
    def forward(ctx, c2w, eng, H, W, focal, hist, Nc, Ni, near, far):
        from . import engine as _e
        o, d, v = _e.raygen(H, W, focal, c2w.detach())
        rgb, disp, acc, saved = _saving_forward(eng, o, d, v, hist, Nc, Ni, near, far)
        ctx.save_for_backward(*saved)
        ctx.cfg = (eng, H, W, focal)
        disp, acc = disp.reshape(H, W), acc.reshape(H, W)
        ctx.mark_non_differentiable(disp, acc)
        return rgb.reshape(H, W, 3), disp, acc

    @staticmethod
    def backward(ctx, g_rgb, _g_disp, _g_acc):
        from . import engine as _e
        eng, H, W, focal = ctx.cfg
        go, gd, _ = _saved_backward(eng, ctx.saved_tensors, g_rgb)
        return (_e.raygen_backward(H, W, focal, go, gd),) + (None,) * 9


class _RaygenFn(torch.autograd.Function):
    """get_rays as an autograd node (ray_utils.py:5-15): (rays_o, rays_d) [H*W,3] of a pose; backward: d L / d c2w [3,4] from the
    ray gradients (dfn_raygen_backward).  What lets a pose that requires grad reach the training render's ray gradients."""

    @staticmethod
    def forward(ctx, c2w, H, W, focal):
        from . import engine as _e
        o, d, _ = _e.raygen(H, W, focal, c2w.detach(), want_viewdirs=False)
        ctx.cfg = (H, W, focal)
        return o.reshape(-1, 3), d.reshape(-1, 3)

    @staticmethod
    def backward(ctx, go, gd):
        from . import engine as _e
        H, W, focal = ctx.cfg
        z = lambda g: torch.zeros(H * W, 3, device=(go if go is not None else gd).device) if g is None else g.contiguous()
        return _e.raygen_backward(H, W, focal, z(go), z(gd)), None, None, None


class _RenderFramesFn(torch.autograd.Function):
    """rgb [B,H,W,3] = render of B frames (each with its own pose and histogram vector) as ONE ray batch: the DFNet_dm step
    renders every frame of its mini-batch (direct_feature_matching.py:340-348 loops over them); batching them gives each
    kernel B times the work per launch.  backward: d L/d c2w [B,3,4] from d L/d rgb."""

    @staticmethod
    def forward(ctx, c2ws, eng, H, W, focal, hists, Nc, Ni, near, far):
        from . import engine as _e
        B = c2ws.shape[0]
        o, d, v = (t.reshape(-1, 3) for t in _e.raygen_frames(H, W, focal, c2ws.detach()))
        hist = hists.reshape(B, 1, -1).expand(B, H * W, hists.shape[-1]).reshape(B * H * W, -1).contiguous()   # one row per ray
        rgb, _, _, saved = _saving_forward(eng, o, d, v, hist, Nc, Ni, near, far)
        ctx.save_for_backward(*saved)
        ctx.cfg = (eng, B, H, W, focal)
        return rgb.reshape(B, H, W, 3)

    @staticmethod
    def backward(ctx, g_rgb):
        from . import engine as _e
        eng, B, H, W, focal = ctx.cfg
        go, gd, _ = _saved_backward(eng, ctx.saved_tensors, g_rgb)
        gc = _e.raygen_frames_backward(H, W, focal, go.reshape(B, H * W, 3), gd.reshape(B, H * W, 3))
        return (gc,) + (None,) * 9


def render_frames(H, W, focal, c2ws, img_idx, **kwargs):
    """rgb [B,H,W,3] of B frames at poses c2ws [B,3,4] with histogram vectors img_idx [B,bins] — render(c2w=...) per frame,
    batched into one launch per stage, differentiable w.r.t. c2ws.  Same option checks as render()."""
    near, far = kwargs.pop('near', 0.), kwargs.pop('far', 1.)
    _check_test_time(kwargs, kwargs.get('ndc', False), None, kwargs.get('use_viewdirs', True), tracked=True)
    eng = _engine_of(kwargs)
    _set_options(eng, kwargs, near)
    c2ws = c2ws[:, :3, :4]
    hists = torch.as_tensor(img_idx, dtype=torch.float32, device=c2ws.device).reshape(c2ws.shape[0], -1)
    return _RenderFramesFn.apply(c2ws, eng, int(H), int(W), float(focal), hists, int(kwargs['N_samples']), int(kwargs['N_importance']),
                                 float(near), float(far))


class _RenderRaysFn(torch.autograd.Function):
    """rgb/disp/acc[/raw] = render(rays); backward: d L/d rays_o, d L/d rays_d (viewdirs = d/|d| differentiated) from d L/d rgb and,
    with retraw, d L/d raw [n,Nf,9] (rendering.py:353-400 with retraw=True under autograd: `raw` is a function of the sample points and
    view directions, the depths are detached).  The fine network's backward is a full vector-Jacobian product of all nine raw channels,
    so an external d L/d raw is ADDED to the compositor's d L/d raw before it."""

    @staticmethod
    def forward(ctx, rays_o, rays_d, eng, hist, Nc, Ni, near, far, retraw=False):
        o, d = rays_o.detach().contiguous(), rays_d.detach().contiguous()
        v = d / torch.norm(d, dim=-1, keepdim=True)
        rgb, disp, acc, saved = _saving_forward(eng, o, d, v, hist, Nc, Ni, near, far)
        ctx.eng = eng
        ctx.mark_non_differentiable(disp, acc)
        if retraw:
            if len(saved) == 5:
                raise NotImplementedError("render(): retraw together with autograd at a network width other than 128 (the generic-width "
                                          "gradient is stateless and takes d L/d rgb only)")
            raw = saved[5].clone()    # the caller's tensor: an in-place edit of it must not reach the state the backward reads
            ctx.save_for_backward(*saved)
            return rgb, disp, acc, raw
        ctx.save_for_backward(*saved)
        return rgb, disp, acc

    @staticmethod
    def backward(ctx, g_rgb, _g_disp, _g_acc, g_raw=None):
        go, gd, _ = _saved_backward(ctx.eng, ctx.saved_tensors, g_rgb, g_raw)
        return (go, gd) + (None,) * 7


def _set_options(eng, kw, near):
    """The render_rays keyword options the handle carries (lindisp, rendering.py:272-273)."""
    lindisp = bool(kw.get('lindisp', False))
    if lindisp and not float(near) > 0.:
        raise ValueError("render(): lindisp=True needs near > 0 (the depths are 1 / ((1 - t) / near + t / far))")
    eng.set_render_options(lindisp=lindisp)


def _check_test_time(kw, ndc, c2w_staticcam, use_viewdirs, training=False, tracked=False):
    bad = []
    if kw.get('white_bkgd', False):
        raise TypeError("render(): white_bkgd=True is not a working option of the NeRF-H path: the reference hands it to the coarse "
                        "compositor as output_transient (models/rendering.py:295) and fails the same way")
    if (ndc or c2w_staticcam is not None) and (training or tracked):
        bad.append("ndc / c2w_staticcam together with " + ("training-mode rendering" if training else "autograd"))
    if not training:
        if not kw.get('test_time', False):
            bad.append("test_time=False without a trainer (render kwargs must come from create_nerf with gradient updates enabled)")
        if float(kw.get('perturb', 0.) or 0.) > 0.:
            bad.append("perturb>0 at test time")
        if float(kw.get('raw_noise_std', 0.) or 0.) != 0.:
            bad.append("raw_noise_std!=0 at test time")
    if not use_viewdirs:
        bad.append("use_viewdirs=False")
    if int(kw.get('N_importance', 0)) <= 0:
        bad.append("N_importance=0")
    if bad:
        raise NotImplementedError("dfnet_amd render(): not implemented natively yet: " + "; ".join(bad) +
                                  " — there is deliberately no CPU fallback")


def render(H, W, focal, chunk=1024 * 32, rays=None, c2w=None, ndc=True, near=0., far=1., use_viewdirs=False,
           c2w_staticcam=None, img_idx=torch.Tensor(0), **kwargs):
    """Drop-in for rendering.py:353-400.  Returns [rgb_map, disp_map, acc_map, extras].

    c2w given: full image, outputs [H,W,3], [H,W], [H,W].  Otherwise `rays` = (rays_o, rays_d) (a tuple
    or a stacked [2,N,3] tensor), outputs shaped like rays_d[..., :1].  `img_idx`: the 10-bin histogram
    index vector, shape [10], [1,10] or [N,10]."""
    eng = _engine_of(kwargs)
    _set_options(eng, kwargs, near)
    trainer = getattr(kwargs.get('network_query_fn'), 'trainer', None)
    if not kwargs.get('test_time', False) and trainer is not None:
        # training mode (rendering.py:245-337 with test_time=False): stratified depths, coarse rgb + noise, importance sampling
        # with random u, the training extras — on the exact-fp32 training kernels, attached to autograd (nerf_train.py)
        _check_test_time(kwargs, ndc, c2w_staticcam, use_viewdirs, training=True)
        # The reference's training render is differentiable w.r.t. its rays / pose too: rays that require grad make the autograd node
        # run the exact-fp32 step and return d L / d rays (nerf_train._RenderTrainFn); c2w reaches them through get_rays' own node.
        from . import nerf_train
        dev = torch.device("cuda", torch.cuda.current_device())
        if c2w is not None:
            c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
            if torch.is_grad_enabled() and c2w.requires_grad:
                rays_o, rays_d = _RaygenFn.apply(c2w[:3, :4].contiguous(), int(H), int(W), float(focal))
                rays_o, rays_d = rays_o.reshape(int(H), int(W), 3), rays_d.reshape(int(H), int(W), 3)
            else:
                rays_o, rays_d = get_rays(H, W, focal, c2w)
        else:
            rays_o, rays_d = rays
        rays_o = torch.as_tensor(rays_o, dtype=torch.float32, device=dev)
        rays_d = torch.as_tensor(rays_d, dtype=torch.float32, device=dev)
        lead = list(rays_d.shape[:-1])
        hist = torch.as_tensor(img_idx, dtype=torch.float32, device=dev).reshape(-1, eng.hist_bin)
        rgb, disp, acc, extras = nerf_train.render_train(trainer, rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), hist, int(kwargs['N_samples']),
                                                         int(kwargs['N_importance']), near, far, float(kwargs.get('perturb', 0.) or 0.),
                                                         float(kwargs.get('raw_noise_std', 0.) or 0.), bool(kwargs.get('retraw', False)),
                                                         draws=kwargs.get('draws'))
        extras = {k: v.reshape(lead + list(v.shape[1:])) for k, v in extras.items()}
        # (no `stale` mark here: a render does not move the weights; HipQuery.refresh() sees optimizer steps through the tensors'
        #  version counters, and the training loop marks them itself)
        return [rgb.reshape(lead + [3]), disp.reshape(lead), acc.reshape(lead), extras]
    kwargs.get('network_query_fn').refresh() if hasattr(kwargs.get('network_query_fn'), 'refresh') else None
    def _needs_grad(t):
        if torch.is_tensor(t):
            return t.requires_grad
        return isinstance(t, (tuple, list)) and any(_needs_grad(u) for u in t)
    track = torch.is_grad_enabled() and (_needs_grad(c2w) or _needs_grad(rays))
    _check_test_time(kwargs, ndc, c2w_staticcam, use_viewdirs, tracked=track)
    Nc, Ni = int(kwargs['N_samples']), int(kwargs['N_importance'])
    retraw = bool(kwargs.get('retraw', False))
    dev = torch.device("cuda", torch.cuda.current_device())
    hist = torch.as_tensor(img_idx, dtype=torch.float32, device=dev)
    if ndc or c2w_staticcam is not None:
        # rendering.py:364-376: the view directions come from the given rays / pose; the rays themselves are then replaced by the
        # static camera's and / or mapped to normalised device coordinates (ndc_rays at near = 1)
        if c2w is not None:
            rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w, dtype=torch.float32, device=dev))
        else:
            rays_o, rays_d = (torch.as_tensor(t, dtype=torch.float32, device=dev) for t in rays)
        view = rays_d.reshape(-1, 3)
        view = view / torch.norm(view, dim=-1, keepdim=True)
        if c2w_staticcam is not None:
            rays_o, rays_d = get_rays(H, W, focal, torch.as_tensor(c2w_staticcam, dtype=torch.float32, device=dev))
        if ndc:
            rays_o, rays_d = ndc_rays(H, W, focal, 1., rays_o, rays_d)
        hist = hist.reshape(-1, eng.hist_bin)
        lead = list(rays_d.shape[:-1])
        rgb, disp, acc, raw = eng.render_rays(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), hist, Nc, Ni, near, far, viewdirs=view,
                                              retraw=retraw)
        extras = {'raw': raw.reshape(lead + list(raw.shape[1:]))} if retraw else {}
        return [rgb.reshape(lead + [3]), disp.reshape(lead), acc.reshape(lead), extras]
    if c2w is not None:
        c2w = torch.as_tensor(c2w, dtype=torch.float32, device=dev)
        if track and not retraw and hist.numel() == eng.hist_bin:
            rgb, disp, acc = _RenderImageFn.apply(c2w[:3, :4], eng, int(H), int(W), float(focal), hist.reshape(-1), Nc, Ni,
                                                  float(near), float(far))
            return [rgb, disp, acc, {}]
        if retraw or hist.numel() != eng.hist_bin:
            if track and c2w.requires_grad:    # the pose reaches the ray gradients through get_rays' own node
                o, d = (t.reshape(int(H), int(W), 3) for t in _RaygenFn.apply(c2w[:3, :4].contiguous(), int(H), int(W), float(focal)))
            else:
                o, d = get_rays(H, W, focal, c2w)
            return render(H, W, focal, chunk, rays=(o, d), ndc=ndc, near=near, far=far, use_viewdirs=use_viewdirs,
                          img_idx=img_idx, **kwargs)
        rgb, disp, acc = eng.render_image(c2w, int(H), int(W), float(focal), hist, Nc, Ni, near, far)
        return [rgb, disp, acc, {}]
    rays_o, rays_d = rays
    rays_o = torch.as_tensor(rays_o, dtype=torch.float32, device=dev)
    rays_d = torch.as_tensor(rays_d, dtype=torch.float32, device=dev)
    sh = rays_d.shape
    n = rays_d.numel() // 3
    hist = hist.reshape(-1, eng.hist_bin)
    if hist.shape[0] not in (1, n):
        raise ValueError(f"img_idx must have 1 or {n} rows of {eng.hist_bin} bins, got {tuple(hist.shape)}")
    lead = list(sh[:-1])
    if track:
        out = _RenderRaysFn.apply(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), eng, hist, Nc, Ni, float(near), float(far), retraw)
        rgb, disp, acc = out[:3]
        extras = {'raw': out[3].reshape(lead + list(out[3].shape[1:]))} if retraw else {}
        return [rgb.reshape(lead + [3]), disp.reshape(lead), acc.reshape(lead), extras]
    rgb, disp, acc, raw = eng.render_rays(rays_o.reshape(-1, 3), rays_d.reshape(-1, 3), hist, Nc, Ni, near, far,
                                          retraw=retraw)
    extras = {'raw': raw.reshape(lead + list(raw.shape[1:]))} if retraw else {}
    return [rgb.reshape(lead + [3]), disp.reshape(lead), acc.reshape(lead), extras]


def _write_png(path, arr8):
    try:
        from PIL import Image
        Image.fromarray(arr8).save(path)
    except ImportError:  # minimal zlib PNG writer (8-bit gray or RGB)
        import struct
        import zlib
        a = np.ascontiguousarray(arr8)
        h, w = a.shape[:2]
        ctype = 2 if a.ndim == 3 else 0
        raw = b"".join(b"\x00" + a[r].tobytes() for r in range(h))
        chunk = lambda t, d: struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        with open(path, "wb") as fh:
            fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                     chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


POST_BATCH = 16     # frames per dfn_frame_post call / device -> host copy / PNG job
PNG_WORKERS = 8     # host threads encoding PNGs while the GPU renders on (zlib releases the GIL)


def _write_frames(ev, host, savedir, first):
    """PNG job of one batch (runs on a pool thread): wait for the batch's device -> host copy, then `{:03d}.png`,
    `{:03d}_GT.png`, `{:03d}_disp.png` per frame, numbered by GLOBAL frame index (rendering.py:438-452)."""
    ev.synchronize()
    rgb8, disp8, gt8 = (None if t is None else t.numpy() for t in host)
    for k in range(rgb8.shape[0]):
        i = first + k
        _write_png(os.path.join(savedir, '{:03d}.png'.format(i)), rgb8[k])
        if gt8 is not None:
            _write_png(os.path.join(savedir, '{:03d}_GT.png'.format(i)), gt8[k] if gt8.ndim == 4 else gt8)
        _write_png(os.path.join(savedir, '{:03d}_disp.png'.format(i)), disp8[k])


def render_path(args, render_poses, hwf, chunk, render_kwargs, gt_imgs=None, savedir=None, render_factor=0,
                single_gt_img=False, img_ids=torch.Tensor(0)):
    """Drop-in for rendering.py:403-458: returns (rgbs [N,H,W,3], disps [N,H,W]) as float32 numpy.

    With torch.distributed initialised (world > 1) each rank renders its contiguous block of frames.  The per-frame back-end
    (to8b truncation, disp / max(disp), the PSNR's mean squared error: rendering.py:423-452) runs on the device in batches
    of POST_BATCH frames (dfn_frame_post); what comes back to the host is uint8, and EVERY RANK writes the PNGs of its own
    block (`{:03d}.png`, `{:03d}_GT.png`, `{:03d}_disp.png` by global frame index) on a thread pool while its GPU renders the
    next batch.  One gather at the end brings the fp32 frames (the returned arrays) and the per-frame errors to rank 0, which
    prints the mean PSNR; ranks != 0 return (None, None)."""
    from concurrent.futures import ThreadPoolExecutor
    from . import engine as eng
    H, W, focal = hwf
    if render_factor != 0:
        H, W, focal = int(H // render_factor), int(W // render_factor), focal / render_factor
    H, W = int(H), int(W)
    dev = torch.device("cuda", torch.cuda.current_device())
    render_poses = torch.as_tensor(render_poses, dtype=torch.float32, device=dev)
    img_ids = torch.as_tensor(img_ids, dtype=torch.float32, device=dev)
    N = render_poses.shape[0]
    rank, world = ddist.rank_world()
    if ddist.active() and ddist.band_mode(N, world):
        return _render_path_bands(render_poses, img_ids, H, W, focal, chunk, render_kwargs, gt_imgs, savedir, single_gt_img)
    lo, hi = ddist.frame_block(N, rank, world)
    n_loc = hi - lo
    # rank 0 renders straight into its block of the final [N, ...] tensors: the end gather receives the other blocks in place
    outs, (rgbs, disps, mse) = ddist.root_buffers([(H, W, 3), (H, W), ()], N, dev)
    mse.zero_()
    gt_one = None
    if gt_imgs is not None and single_gt_img:
        gt_one = torch.as_tensor(np.asarray(gt_imgs), dtype=torch.float32).to(dev)
    pool = ThreadPoolExecutor(max_workers=PNG_WORKERS) if savedir is not None else None
    jobs = []
    t0 = time.time()
    t_post = 0.0
    for j0 in range(0, n_loc, POST_BATCH):
        j1 = min(n_loc, j0 + POST_BATCH)
        for j in range(j0, j1):
            i = lo + j
            rgb, disp, _, _ = render(H, W, focal, chunk=chunk, c2w=render_poses[i][:3, :4], img_idx=img_ids[i], **render_kwargs)
            rgbs[j].copy_(rgb)
            disps[j].copy_(disp)
            if i == 0:
                print(rgb.shape, disp.shape)
        if savedir is None and gt_imgs is None:
            continue
        tp = time.time()
        gt = gt_one
        if gt_imgs is not None and not single_gt_img:
            gt = torch.as_tensor(np.asarray(gt_imgs[lo + j0:lo + j1]), dtype=torch.float32).to(dev, non_blocking=True)
        post = eng.frame_post(rgbs[j0:j1], disps[j0:j1], gt, want_gt8=savedir is not None)
        if gt is not None:
            mse[j0:j1].copy_(post["mse"])
        if savedir is not None:
            host = []
            for t in (post["rgb8"], post["disp8"], post["gt8"]):
                host.append(None if t is None else torch.empty(t.shape, dtype=torch.uint8, pin_memory=True).copy_(t, non_blocking=True))
            ev = torch.cuda.Event()
            ev.record()
            jobs.append(pool.submit(_write_frames, ev, host, savedir, lo + j0))
        t_post += time.time() - tp
    torch.cuda.synchronize()
    t_render = time.time() - t0
    # The range guard of the narrow arithmetic modes: a rank that raised before the collective would leave the others waiting in
    # it, and a rank that did NOT raise after it would walk into the caller's next collective alone.  So every rank's flag word goes
    # to EVERY rank (a 4-byte int32 all-gather behind the frame gather: bits OR, nothing is lost to a float MAX) and all of them raise
    # the same error, after the collective and after the PNG pool has been drained.
    eng_h = _engine_of(render_kwargs)
    my_flags = eng_h.range_flags()
    all_flags = [my_flags]
    tg = time.time()
    try:
        (all_rgb, all_disp, all_mse), _ = ddist.gather_frames_direct([rgbs, disps, mse], N, outs=outs)   # the path's ONE data collective
        all_flags = ddist.all_gather_flags(my_flags, dev)
        torch.cuda.synchronize()
    finally:
        t_gather = time.time() - tg
        tw = time.time()
        errs = []
        for jb in jobs:
            try:
                jb.result()
            except Exception as e:   # noqa: BLE001 - re-raised below, after the pool is shut down
                errs.append(e)
        if pool is not None:
            pool.shutdown(wait=True)
        t_tail = time.time() - tw
    if errs:
        raise errs[0]
    bad = 0
    for f in all_flags:
        bad |= int(f)
    if bad:   # the same message on every rank
        eng_h.raise_range(bad, where=f"render_path: frames {lo}..{hi - 1}" if len(all_flags) == 1 else
                          "render_path: ranks " + ", ".join(f"{r} (flags {f:#x})" for r, f in enumerate(all_flags) if f))
    if gt_imgs is None:
        all_mse = None
    n_gathered = ddist.gathered_bytes([rgbs, disps, mse], N)
    if rank != 0:
        return None, None
    rgbs = all_rgb.cpu().numpy()
    disps = all_disp.cpu().numpy()
    print(f"rendered {N} frames of {W}x{H} on {world} GPU(s) in {t_render:.2f} s (post-processing launches {t_post:.2f} s inside it, "
          f"PNG tail after the last frame {t_tail:.2f} s)")
    if all_mse is not None:
        psnr = -10. * np.log10(all_mse.cpu().numpy())
        print("Mean PSNR of this run is:", np.mean(psnr, 0))
    render_path.last_timing = {"render_s": t_render, "post_launch_s": t_post, "gather_s": t_gather, "png_tail_s": t_tail, "frames": N, "world": world,
                               "gathered_bytes": n_gathered}
    return rgbs, disps


def render_band(H, W, focal, chunk, c2w, img_idx, r0, r1, render_kwargs):
    """Rows [r0, r1) of the frame at pose c2w: the band's rays out of get_rays (models/ray_utils.py:5-15) through render(rays=...).
    Rays are independent, so a band is bit-identical to the same rows of the full-frame render (tests/test_gpu_dist.py)."""
    rays_o, rays_d = get_rays(H, W, focal, c2w)
    rgb, disp, _, _ = render(H, W, focal, chunk=chunk, rays=(rays_o[r0:r1].contiguous(), rays_d[r0:r1].contiguous()), img_idx=img_idx,
                             **render_kwargs)
    return rgb, disp


def _render_path_bands(render_poses, img_ids, H, W, focal, chunk, render_kwargs, gt_imgs, savedir, single_gt_img):
    """SURVEY 8(e)'s small-batch fallback of render_path — fewer frames than ranks (the single validation frames of run_nerf.py:200,228;
    configs[0]'s four frames on eight GPUs): every rank renders ONE row band of one frame (dist.band_unit), the bands are gathered in
    place into rank 0's final [N, H, ...] tensors (dist.gather_bands_direct: the same single grouped send / receive batch), and the
    per-FRAME back-end — disp / max(disp) over the whole frame, the PSNR's mean squared error, to8b, the PNGs: rendering.py:423-452 —
    runs on rank 0 on the assembled frames (N < world <= 8 of them), so every number is the one a single GPU produces."""
    from . import engine as eng
    dev = render_poses.device
    N = render_poses.shape[0]
    rank, world = ddist.rank_world()
    f, r0, r1 = ddist.band_unit(N, H, rank, world)
    t0 = time.time()
    if r1 > r0:
        rgb, disp = render_band(H, W, focal, chunk, render_poses[f][:3, :4], img_ids[f], r0, r1, render_kwargs)
    else:   # more ranks on this frame than rows
        rgb, disp = torch.empty(0, W, 3, device=dev), torch.empty(0, W, device=dev)
    if rank == 0:
        print(torch.Size([H, W, 3]), torch.Size([H, W]))
    torch.cuda.synchronize() if dev.type == "cuda" else None
    t_render = time.time() - t0
    eng_h = _engine_of(render_kwargs)
    my_flags = eng_h.range_flags()
    tg = time.time()
    all_rgb, all_disp = ddist.gather_bands_direct([rgb, disp], N, H)      # the path's ONE data collective
    all_flags = ddist.all_gather_flags(my_flags, dev)
    t_gather = time.time() - tg
    bad = 0
    for fl in all_flags:
        bad |= int(fl)
    if bad:   # the same message on every rank, after the collective
        eng_h.raise_range(bad, where="render_path (row bands): ranks " + ", ".join(f"{r} (flags {fl:#x})" for r, fl in enumerate(all_flags) if fl))
    if rank != 0:
        return None, None
    tp = time.time()
    mse = None
    if savedir is not None or gt_imgs is not None:
        gt = None
        if gt_imgs is not None:
            g = np.asarray(gt_imgs)
            gt = torch.as_tensor(g if single_gt_img else g[:N], dtype=torch.float32).to(dev)
        post = eng.frame_post(all_rgb, all_disp, gt, want_gt8=savedir is not None)
        mse = post["mse"] if gt is not None else None
        if savedir is not None:
            host = [None if t is None else t.cpu() for t in (post["rgb8"], post["disp8"], post["gt8"])]

            class _Done:
                def synchronize(self):
                    pass
            _write_frames(_Done(), host, savedir, 0)
    t_post = time.time() - tp
    print(f"rendered {N} frames of {W}x{H} as row bands on {world} GPU(s) in {t_render:.2f} s (gather {t_gather:.2f} s, frame back-end on "
          f"rank 0 {t_post:.2f} s)")
    if mse is not None:
        print("Mean PSNR of this run is:", np.mean(-10. * np.log10(mse.cpu().numpy()), 0))
    per_frame = (H * W * 4) * 4
    own = r1 - r0
    render_path.last_timing = {"render_s": t_render, "post_launch_s": t_post, "gather_s": t_gather, "png_tail_s": 0.0, "frames": N, "world": world,
                               "gathered_bytes": N * per_frame - own * W * 16, "row_bands": True}
    return all_rgb.cpu().numpy(), all_disp.cpu().numpy()


def _drain(dl):
    imgs, poses, idxs = [], [], []
    for img, pose, img_idx in dl:
        imgs.append(img.permute(0, 2, 3, 1))
        p = torch.zeros(1, 4, 4)
        p[0, :3, :4] = pose.reshape(3, 4)[:3, :4]
        p[0, 3, 3] = 1.
        poses.append(p)
        idxs.append(img_idx)
    return torch.cat(imgs, 0).cpu().numpy(), torch.cat(poses, 0), torch.cat(idxs, 0)   # gt frames may be device tensors


def render_test(args, train_dl, val_dl, hwf, start, render_kwargs_test, decoder_coarse=None, decoder_fine=None):
    """Drop-in for rendering.py:460-530: render the train split then the val split, write PNGs under
    basedir/expname/evaluate_{train,val}_{test|path}_{start:06d}."""
    tag = 'test' if args.render_test else 'path'
    for name, dl in (("train", train_dl), ("val", val_dl)):
        savedir = os.path.join(args.basedir, args.expname, 'evaluate_{}_{}_{:06d}'.format(name, tag, start))
        os.makedirs(savedir, exist_ok=True)
        images, poses, index = _drain(dl)
        print(f'{name} poses shape', poses.shape)
        with torch.no_grad():
            render_path(args, poses, hwf, args.chunk, render_kwargs_test, gt_imgs=images, savedir=savedir,
                        img_ids=index)
        print(f'Saved {name} set')
    return

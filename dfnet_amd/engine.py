"""NerfHEngine — host-side owner of a `dfn_nerfh_t` handle and thin tensor-level wrappers of the
stage and whole-path entry points of libdfnet_hip.so.

All tensors are fp32 CUDA tensors owned by torch (device memory + streams are torch's job; the
arithmetic is the HIP library's).  Work is enqueued on torch's current stream.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, current_stream, ptr


def _grow_workspace(old, nbytes, device):
    """A persistent byte workspace of >= nbytes on `device` (the old one if it fits).  Engines keep one workspace across calls that
    may come from different torch streams (direct_feature_matching runs the target features on a side stream), so a new block is
    always taken from the DEFAULT stream's pool — never tied to a side stream — the calling stream waits for whatever the default
    stream still has queued on a recycled block, and the block being replaced is marked in use by the calling stream: the allocator
    hands it out again only after the kernels queued there have finished."""
    if old is not None and old.device == device and old.numel() >= nbytes:
        return old
    if device.type != "cuda":
        return torch.empty(nbytes, dtype=torch.uint8, device=device)
    cur, default = torch.cuda.current_stream(device), torch.cuda.default_stream(device)
    if old is not None and old.is_cuda:
        old.record_stream(cur)
    with torch.cuda.stream(default):
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if cur != default:
        cur.wait_stream(default)
    return ws


# DFN_DEBUG_POISON_UNREAD=1: feature planes / gradient planes that a "read only these levels" promise leaves unwritten (zero_unread=False,
# lazy_grad) are filled with NaN instead of being left as uninitialised memory — the DFNet_dm gradient tests run once under it
# (tests/test_gpu_grad.py) to prove that no unhinted level is ever read.
POISON_UNREAD = os.environ.get("DFN_DEBUG_POISON_UNREAD", "0") == "1"


def _f32c(t):
    return t.contiguous().float()


class NerfHEngine:
    """NeRF-H coarse+fine networks resident on one GPU in MFMA-fragment layout."""

    def __init__(self, depth=8, width=128, multires=10, multires_views=4, hist_bin=10, dim_a=5, dim_t=2,
                 n_vocab=1000, precision="f16"):
        self.lib = _lib.load()
        self.desc = _lib.NerfhDesc(depth, width, multires, multires_views, hist_bin, dim_a, dim_t, n_vocab)
        self.handle = ctypes.c_void_p()
        check(self.lib.dfn_nerfh_create(ctypes.byref(self.desc), ctypes.byref(self.handle)), "dfn_nerfh_create")
        self.precision = precision
        self.hist_bin = hist_bin
        self.width = width
        self.fast = width in (128, 256)   # register-resident MFMA kernels; other widths run the generic layer-by-layer fp32 path
        self._ws = None
        self.lindisp = False
        self.coarse_f16 = False

    def set_render_options(self, lindisp=False, coarse_f16=None):
        """render_rays options that every entry point of this handle applies (dfn_nerfh_set_render_options): lindisp = coarse
        depths linear in disparity (rendering.py:272-273); coarse_f16 (None = leave as is) = the coarse network of the test-time
        render runs with f16 inputs whatever precision the call names (it only places the fine samples)."""
        lindisp = bool(lindisp)
        coarse_f16 = self.coarse_f16 if coarse_f16 is None else bool(coarse_f16)
        if lindisp != self.lindisp or coarse_f16 != self.coarse_f16:
            flags = (_lib.RENDER_LINDISP if lindisp else 0) | (_lib.RENDER_COARSE_F16 if coarse_f16 else 0)
            check(self.lib.dfn_nerfh_set_render_options(self.handle, flags), "dfn_nerfh_set_render_options")
            self.lindisp, self.coarse_f16 = lindisp, coarse_f16
        return self

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.dfn_nerfh_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_numpy(self, coarse, fine, emb_a, emb_t):
        """coarse/fine: {state_dict key: ndarray}; emb_a/emb_t: ndarrays."""
        items = [("coarse." + k, v) for k, v in coarse.items()] + [("fine." + k, v) for k, v in fine.items()]
        items += [("embedding_a.weight", emb_a), ("embedding_t.weight", emb_t)]
        for name, arr in items:
            a = np.ascontiguousarray(arr, dtype=np.float32)
            check(self.lib.dfn_nerfh_set_param(self.handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                  f"dfn_nerfh_set_param({name})")
        check(self.lib.dfn_nerfh_commit(self.handle), "dfn_nerfh_commit")
        return self

    def load_modules(self, network_fn, network_fine, embedding_a, embedding_t):
        """Take the weights of torch modules with the reference's state_dict layout."""
        sd = lambda m: {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
        return self.load_numpy(sd(network_fn), sd(network_fine), embedding_a.weight.detach().cpu().numpy(),
                               embedding_t.weight.detach().cpu().numpy())

    def range_flags(self):
        """DFN_RANGE_* bits raised by the MLP kernels since the last call (waits for the current stream, clears them)."""
        v = ctypes.c_int(0)
        check(self.lib.dfn_nerfh_range_status(self.handle, ctypes.byref(v), current_stream()), "dfn_nerfh_range_status")
        return v.value

    def check_range(self):
        """Raise DfnError if an f16 / split-f16 activation overflowed / saturated in a render since the last check: such frames
        are not the network's output (render with precision='f32')."""
        check(self.lib.dfn_nerfh_range_status(self.handle, None, current_stream()), "NeRF-H range guard")

    def raise_range(self, flags, where=""):
        """The range guard's error for flags already fetched with range_flags() (render_path carries them through its gather so that
        every rank leaves the collective before anyone raises)."""
        if flags:
            narrow = "f16" if flags & 1 else "split-f16"
            why = ("an f16 layer output overflowed to inf: |activation| > 65504" if flags & 1
                   else "a split-f16 hi half saturated: |activation| >= 4094 (hidden layers of the render kernels: 4094 x the largest |weight|)")
            raise _lib.DfnError(f"NeRF-H range guard ({where}): activations left the range of the {narrow} arithmetic ({why}); the frames "
                                "rendered since the last check are not the network's output: render them with precision='f32'")

    def _prec(self, precision):
        return _lib.PRECISIONS[precision or self.precision]

    def _workspace(self, nbytes, device):
        self._ws = _grow_workspace(self._ws, nbytes, device)
        return self._ws

    # ------------------------------------------------------------------ stages
    def mlp_coarse(self, rays_o, rays_d, Nc, near, far, precision=None):
        rays_o, rays_d = _f32c(rays_o), _f32c(rays_d)
        n = rays_o.shape[0]
        sigma = torch.empty(n, Nc, device=rays_o.device)
        check(self.lib.dfn_mlp_coarse(self.handle, self._prec(precision), ptr(rays_o), ptr(rays_d), n, Nc,
                                      float(near), float(far), ptr(sigma), current_stream()), "dfn_mlp_coarse")
        return sigma

    def mlp_fine(self, rays_o, rays_d, viewdirs, hist, z_fine, precision=None):
        rays_o, rays_d, viewdirs, z_fine = _f32c(rays_o), _f32c(rays_d), _f32c(viewdirs), _f32c(z_fine)
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        n, Nf = z_fine.shape
        raw = torch.empty(n, Nf, 9, device=rays_o.device)
        bias = torch.empty(self.lib.dfn_fine_bias_bytes(n), dtype=torch.uint8, device=rays_o.device)
        check(self.lib.dfn_mlp_fine(self.handle, self._prec(precision), ptr(rays_o), ptr(rays_d), ptr(viewdirs),
                                    ptr(hist), hist.shape[0], n, ptr(z_fine), Nf, ptr(raw),
                                    ctypes.c_void_p(bias.data_ptr()), current_stream()), "dfn_mlp_fine")
        return raw

    def mlp_fine_backward(self, rays_o, rays_d, viewdirs, hist, z_fine, grad_raw, precision=None):
        """d L/d raw [n,Nf,9] -> [n,Nf,6]: d L/d sample point (3) and d L/d viewdir through that sample (3)."""
        rays_o, rays_d, viewdirs, z_fine, grad_raw = (_f32c(t) for t in (rays_o, rays_d, viewdirs, z_fine, grad_raw))
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        n, Nf = z_fine.shape
        gpts = torch.empty(n, Nf, 6, device=rays_o.device)
        bias = torch.empty(self.lib.dfn_fine_bias_bytes(n), dtype=torch.uint8, device=rays_o.device)
        check(self.lib.dfn_mlp_fine_backward(self.handle, self._prec(precision), ptr(rays_o), ptr(rays_d), ptr(viewdirs),
                                             ptr(hist), hist.shape[0], n, ptr(z_fine), Nf, ptr(grad_raw), ptr(gpts),
                                             ctypes.c_void_p(bias.data_ptr()), current_stream()), "dfn_mlp_fine_backward")
        return gpts

    def mlp_fine_saving(self, rays_o, rays_d, viewdirs, hist, z_fine):
        """mlp_fine in split-f16 that also records the ReLU signs: (raw [n,Nf,9], masks) for mlp_fine_backward_saved()."""
        rays_o, rays_d, viewdirs, z_fine = _f32c(rays_o), _f32c(rays_d), _f32c(viewdirs), _f32c(z_fine)
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        n, Nf = z_fine.shape
        dev = rays_o.device
        raw = torch.empty(n, Nf, 9, device=dev)
        masks = torch.empty(self.lib.dfn_mlp_fine_mask_bytes(n * Nf), dtype=torch.uint8, device=dev)
        bias = torch.empty(self.lib.dfn_fine_bias_bytes(n), dtype=torch.uint8, device=dev)
        check(self.lib.dfn_mlp_fine_saving(self.handle, _lib.PRECISIONS["f16x3"], ptr(rays_o), ptr(rays_d), ptr(viewdirs), ptr(hist),
                                           hist.shape[0], n, ptr(z_fine), Nf, ptr(raw), ctypes.c_void_p(masks.data_ptr()),
                                           ctypes.c_void_p(bias.data_ptr()), current_stream()), "dfn_mlp_fine_saving")
        return raw, masks

    def mlp_fine_backward_saved(self, rays_o, rays_d, viewdirs, z_fine, raw, masks, grad_raw):
        """The gradient of mlp_fine_backward from the saved forward (raw + ReLU masks): no forward recompute."""
        rays_o, rays_d, viewdirs, z_fine, raw, grad_raw = (_f32c(t) for t in (rays_o, rays_d, viewdirs, z_fine, raw, grad_raw))
        n, Nf = z_fine.shape
        gpts = torch.empty(n, Nf, 6, device=rays_o.device)
        check(self.lib.dfn_mlp_fine_backward_saved(self.handle, _lib.PRECISIONS["f16x3"], ptr(rays_o), ptr(rays_d), ptr(viewdirs), n,
                                                   ptr(z_fine), Nf, ptr(raw), ctypes.c_void_p(masks.data_ptr()), ptr(grad_raw),
                                                   ptr(gpts), current_stream()), "dfn_mlp_fine_backward_saved")
        return gpts

    # ------------------------------------------------------------------ whole path
    GENERIC_GRAD_CHUNK = 8192   # rays per pass of the generic-width gradient (every fine activation is kept: ~1.3 MB per ray at 64+128, netwidth 256)
    GENERIC_CHUNK = 4096   # rays per pass of the generic path (its activations live in HBM: ~1.2 MB per ray at 64+128, W=128)

    def generic_render_rays(self, rays_o, rays_d, hist, Nc, Ni, near, far, retraw=False):
        """Test-time render on the generic-width path (dfn_nerfh_generic_render_rays): exact fp32, any netwidth."""
        rays_o, rays_d = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3)
        n, dev = rays_o.shape[0], rays_o.device
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        rgb, disp, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
        Nf = Nc + Ni
        raw_all = torch.empty(n, Nf, 9, device=dev) if retraw else None
        C = self.GENERIC_CHUNK
        ws = self._workspace(self.lib.dfn_nerfh_generic_workspace_bytes(self.handle, min(n, C), Nc, Ni), dev)
        raw_tmp = None if retraw else torch.empty(min(n, C), Nf, 9, device=dev)
        for r0 in range(0, n, C):
            m = min(C, n - r0)
            h = hist if hist.shape[0] == 1 else hist[r0:r0 + m]
            raw = raw_all[r0:r0 + m] if retraw else raw_tmp[:m]
            check(self.lib.dfn_nerfh_generic_render_rays(self.handle, ptr(rays_o[r0:r0 + m]), ptr(rays_d[r0:r0 + m]), ptr(h), h.shape[0], m,
                                                         Nc, Ni, float(near), float(far), ptr(rgb[r0:r0 + m]), ptr(disp[r0:r0 + m]),
                                                         ptr(acc[r0:r0 + m]), ptr(raw), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                                         current_stream()), "dfn_nerfh_generic_render_rays")
        return rgb, disp, acc, raw_all

    def render_rays(self, rays_o, rays_d, hist, Nc, Ni, near, far, viewdirs=None, retraw=False, precision=None):
        """Test-time render of a ray batch -> (rgb [n,3], disp [n], acc [n], raw|None)."""
        if not self.fast or precision == "generic":
            if viewdirs is not None:
                raise NotImplementedError("explicit viewdirs on the generic-width path")
            return self.generic_render_rays(rays_o, rays_d, hist, Nc, Ni, near, far, retraw)
        rays_o, rays_d = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3)
        n = rays_o.shape[0]
        dev = rays_o.device
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        if viewdirs is not None:
            viewdirs = _f32c(viewdirs).reshape(-1, 3)
        rgb = torch.empty(n, 3, device=dev)
        disp = torch.empty(n, device=dev)
        acc = torch.empty(n, device=dev)
        raw = torch.empty(n, Nc + Ni, 9, device=dev) if retraw else None
        nbytes = self.lib.dfn_render_workspace_bytes(n, Nc, Ni)
        ws = self._workspace(nbytes, dev)
        check(self.lib.dfn_render_rays(self.handle, self._prec(precision), ptr(rays_o), ptr(rays_d), ptr(viewdirs),
                                       ptr(hist), hist.shape[0], n, Nc, Ni, float(near), float(far), ptr(rgb),
                                       ptr(disp), ptr(acc), ptr(raw), ctypes.c_void_p(ws.data_ptr()), ws.numel(),
                                       current_stream()), "dfn_render_rays")
        return rgb, disp, acc, raw

    def render_image(self, c2w, H, W, focal, hist, Nc, Ni, near, far, precision=None, out=None):
        """Test-time render of a full image from a [3,4] (or [4,4]) c2w -> (rgb [H,W,3], disp, acc [H,W])."""
        c2w = _f32c(c2w)[:3, :4].contiguous()
        dev = c2w.device
        hist = _f32c(hist).reshape(-1)[: self.hist_bin].contiguous()
        if not self.fast or precision == "generic":
            o, d, _ = raygen(H, W, focal, c2w, want_viewdirs=False)
            rgb, disp, acc, _ = self.generic_render_rays(o.reshape(-1, 3), d.reshape(-1, 3), hist, Nc, Ni, near, far)
            res = (rgb.reshape(H, W, 3), disp.reshape(H, W), acc.reshape(H, W))
            if out is not None:
                for dst, src in zip(out, res):
                    dst.copy_(src)
                return out
            return res
        if out is None:
            out = (torch.empty(H, W, 3, device=dev), torch.empty(H, W, device=dev), torch.empty(H, W, device=dev))
        rgb, disp, acc = out
        nbytes = self.lib.dfn_render_workspace_bytes(H * W, Nc, Ni)
        ws = self._workspace(nbytes, dev)
        check(self.lib.dfn_render_image(self.handle, self._prec(precision), ptr(c2w), H, W, float(focal), float(near),
                                        float(far), Nc, Ni, ptr(hist), ptr(rgb), ptr(disp), ptr(acc),
                                        ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_render_image")
        return rgb, disp, acc


    # ------------------------------------------------------------------ staged render that keeps what backward needs
    def render_rays_saving(self, rays_o, rays_d, viewdirs, hist, Nc, Ni, near, far, precision=None, with_masks=False):
        """render_rays composed from the stage entry points, returning (rgb, disp, acc, z_fine, raw[, masks]): the state
        from which backward_from_saved() differentiates without recomputing the forward.  with_masks: the fine net runs
        in split-f16 and records its ReLU signs, so the backward needs no forward pass of its own at all."""
        rays_o, rays_d, viewdirs = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3), _f32c(viewdirs).reshape(-1, 3)
        sigma = self.mlp_coarse(rays_o, rays_d, Nc, near, far, precision)
        z = sample_fine(sigma, Ni, near, far, lindisp=self.lindisp)
        if with_masks:
            raw, masks = self.mlp_fine_saving(rays_o, rays_d, viewdirs, hist, z)
            out = composite_fine(raw, z)
            return out["rgb"], out["disp"], out["acc"], z, raw, masks
        raw = self.mlp_fine(rays_o, rays_d, viewdirs, hist, z, precision)
        out = composite_fine(raw, z)
        return out["rgb"], out["disp"], out["acc"], z, raw

    def backward_from_saved(self, rays_o, rays_d, viewdirs, hist, z, raw, grad_rgb, derive_viewdirs=True, precision=None, masks=None,
                            grad_raw=None):
        """d L/d (rays_o, rays_d[, viewdirs]) from the saved (z_fine, raw[, masks]) of render_rays_saving().  grad_raw [n,Nf,9]: a
        gradient that reaches `raw` directly (render(retraw=True) under autograd), added to the compositor's."""
        rays_o, rays_d, viewdirs = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3), _f32c(viewdirs).reshape(-1, 3)
        graw = composite_fine_backward(raw, z, _f32c(grad_rgb).reshape(-1, 3))
        if grad_raw is not None:
            graw += _f32c(grad_raw).reshape(graw.shape)
        if masks is not None:
            gpts = self.mlp_fine_backward_saved(rays_o, rays_d, viewdirs, z, raw, masks, graw)
        else:
            gpts = self.mlp_fine_backward(rays_o, rays_d, viewdirs, hist, z, graw, precision)
        n, Nf = z.shape
        go, gd = torch.empty(n, 3, device=z.device), torch.empty(n, 3, device=z.device)
        gv = None if derive_viewdirs else torch.empty(n, 3, device=z.device)
        check(self.lib.dfn_ray_grad_reduce(ptr(gpts), ptr(z), ptr(rays_d), n, Nf, int(derive_viewdirs), ptr(go), ptr(gd), ptr(gv),
                                           current_stream()), "dfn_ray_grad_reduce")
        return go, gd, gv

    # ------------------------------------------------------------------ gradient of the whole path
    def render_rays_backward(self, rays_o, rays_d, hist, Nc, Ni, near, far, grad_rgb, viewdirs=None, precision=None):
        """d L/d (rays_o, rays_d[, viewdirs]) of render_rays from d L/d rgb [n,3].  With viewdirs=None they are
        d/|d| and their gradient is folded into grad_rays_d (what autograd does for render(rays=...))."""
        rays_o, rays_d = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3)
        grad_rgb = _f32c(grad_rgb).reshape(-1, 3)
        n, dev = rays_o.shape[0], rays_o.device
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        go, gd = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev)
        gv = None
        if viewdirs is not None:
            viewdirs = _f32c(viewdirs).reshape(-1, 3)
            gv = torch.empty(n, 3, device=dev)
        if self.width != 128 or precision == "generic":
            # the register-resident gradient kernels are netwidth 128; every other width takes the layer-by-layer exact-fp32 path
            C = self.GENERIC_GRAD_CHUNK
            ws = self._workspace(self.lib.dfn_nerfh_generic_backward_workspace_bytes(self.handle, min(n, C), Nc, Ni), dev)
            for r0 in range(0, n, C):
                m = min(C, n - r0)
                hh = hist if hist.shape[0] == 1 else hist[r0:r0 + m]
                check(self.lib.dfn_nerfh_generic_render_rays_backward(
                    self.handle, ptr(rays_o[r0:r0 + m]), ptr(rays_d[r0:r0 + m]), ptr(None if viewdirs is None else viewdirs[r0:r0 + m]),
                    ptr(hh), hh.shape[0], m, Nc, Ni, float(near), float(far), ptr(grad_rgb[r0:r0 + m]), ptr(go[r0:r0 + m]), ptr(gd[r0:r0 + m]),
                    ptr(None if gv is None else gv[r0:r0 + m]), ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
                    "dfn_nerfh_generic_render_rays_backward")
            return go, gd, gv
        ws = self._workspace(self.lib.dfn_render_backward_workspace_bytes(n, Nc, Ni), dev)
        check(self.lib.dfn_render_rays_backward(self.handle, self._prec(precision), ptr(rays_o), ptr(rays_d), ptr(viewdirs),
                                                ptr(hist), hist.shape[0], n, Nc, Ni, float(near), float(far),
                                                ptr(grad_rgb), ptr(go), ptr(gd), ptr(gv),
                                                ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_render_rays_backward")
        return go, gd, gv

    def render_image_backward(self, c2w, H, W, focal, hist, Nc, Ni, near, far, grad_rgb, precision=None):
        """d L/d c2w [3,4] of render_image from d L/d rgb [H,W,3]."""
        c2w = _f32c(c2w)[:3, :4].contiguous()
        dev = c2w.device
        hist = _f32c(hist).reshape(-1)[: self.hist_bin].contiguous()
        grad_rgb = _f32c(grad_rgb).reshape(H, W, 3)
        gc = torch.empty(3, 4, device=dev)
        if self.width != 128 or precision == "generic":   # get_rays, the generic-width ray gradient, get_rays backward
            o, d, _ = raygen(H, W, focal, c2w, want_viewdirs=False)
            go, gd, _ = self.render_rays_backward(o.reshape(-1, 3), d.reshape(-1, 3), hist, Nc, Ni, near, far, grad_rgb.reshape(-1, 3),
                                                  precision="generic")
            return raygen_backward(H, W, focal, go, gd)
        ws = self._workspace(self.lib.dfn_render_backward_workspace_bytes(H * W, Nc, Ni), dev)
        check(self.lib.dfn_render_image_backward(self.handle, self._prec(precision), ptr(c2w), H, W, float(focal),
                                                 float(near), float(far), Nc, Ni, ptr(hist), ptr(grad_rgb), ptr(gc),
                                                 ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_render_image_backward")
        return gc


class DfnetEngine:
    """DFNet / DFNet_s feature extractor resident on one GPU (convs packed as MFMA fragments)."""

    def __init__(self, n_taps=3, feat_dim=12, precision="f16x3"):
        self.lib = _lib.load()
        self.n_taps, self.feat_dim, self.precision = n_taps, feat_dim, precision
        self.kept_tapes = []   # data_ptr()s of the workspaces the last kept forwards filled (the handle remembers 8)
        self.handle = ctypes.c_void_p()
        check(self.lib.dfn_dfnet_create(n_taps, feat_dim, ctypes.byref(self.handle)), "dfn_dfnet_create")
        self._ws = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.dfn_dfnet_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def load_numpy(self, state):
        """state: {state_dict key: ndarray} with the reference's DFNet names (num_batches_tracked ignored)."""
        for name, arr in state.items():
            if name.endswith("num_batches_tracked"):
                continue
            a = np.ascontiguousarray(arr, dtype=np.float32)
            check(self.lib.dfn_dfnet_set_param(self.handle, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                  f"dfn_dfnet_set_param({name})")
        check(self.lib.dfn_dfnet_commit(self.handle), "dfn_dfnet_commit")
        return self

    def forward(self, x, return_feature=False, isSingleStream=False, return_pose=True, upsampleH=240, upsampleW=427,
                precision=None, levels=None, zero_unread=True):
        """(features, pose): features is None, [n_taps,B,128,uH,uW] (single stream) or a (target, render) pair of
        [n_taps,B/2,128,uH,uW]; pose is None or [B, feat_dim].
        levels (features without pose only): the pyramid levels the caller will read (the reference's
        index_select(features, 0, args.feature_matching_lvl)): only those are computed (dfn_dfnet_forward_levels), the planes of
        the others are zeros — or, with zero_unread=False (a caller that promises to read only `levels`: the DFNet_dm step's
        one-shot hint), left UNWRITTEN: zeroing two unread 39 MB planes costs 0.06 ms per forward at batch 4."""
        x = _f32c(x)
        B, C, H, W = x.shape
        assert C == 3
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        feats = pose = None
        if return_feature:
            shape = (self.n_taps, B, 128, upsampleH, upsampleW) if isSingleStream else \
                (2, self.n_taps, B // 2, 128, upsampleH, upsampleW)
            pruned = levels is not None and not return_pose and set(int(t) for t in levels) != set(range(self.n_taps))
            feats = torch.zeros(shape, device=dev) if (pruned and zero_unread) else torch.empty(shape, device=dev)
            if pruned and not zero_unread and POISON_UNREAD:
                feats.fill_(float("nan"))   # debug: a consumer that reads a level it did not ask for sees NaN, not stale memory
        if return_pose:
            pose = torch.empty(B, self.feat_dim, device=dev)
        nbytes = self.lib.dfn_dfnet_workspace_bytes(self.handle, prec, B, H, W)
        self._ws = _grow_workspace(self._ws, nbytes, dev)
        if return_feature and pruned:
            mask = sum(1 << int(t) for t in set(levels))
            check(self.lib.dfn_dfnet_forward_levels(self.handle, prec, ptr(x), B, H, W, int(not isSingleStream), mask, int(upsampleH),
                                                    int(upsampleW), ptr(feats), ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                                    current_stream()), "dfn_dfnet_forward_levels")
            return ((feats[0], feats[1]) if not isSingleStream else feats), None
        check(self.lib.dfn_dfnet_forward(self.handle, prec, ptr(x), B, H, W, int(return_feature),
                                         int(not isSingleStream), int(return_pose), int(upsampleH), int(upsampleW),
                                         ptr(feats), ptr(pose), ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                         current_stream()), "dfn_dfnet_forward")
        if return_feature and not isSingleStream:
            feats = (feats[0], feats[1])
        return feats, pose

    CONV_INDEX = (0, 2, 5, 7, 10, 12, 14, 17, 19, 21, 24, 26, 28)  # encoder positions of the 13 convs (VGG16 cfg "D")

    def _remember(self, ws):
        self.kept_tapes = [p for p in self.kept_tapes if p != ws.data_ptr()][-7:] + [ws.data_ptr()]

    def holds(self, tape):
        """True if `tape` is (still) one of the kept forwards the handle can run a backward from."""
        return tape is not None and tape.data_ptr() in self.kept_tapes

    def forward_pose_keep(self, x, precision=None):
        """Pose regression keeping the encoder's activations: (pose [B, feat_dim], tape) for backward_params(tape=...)."""
        x = _f32c(x)
        B, C, H, W = x.shape
        prec = _lib.PRECISIONS[precision or self.precision]
        pose = torch.empty(B, self.feat_dim, device=x.device)
        ws = torch.empty(self.lib.dfn_dfnet_backward_params_workspace_bytes(self.handle, prec, B, H, W), dtype=torch.uint8, device=x.device)
        check(self.lib.dfn_dfnet_forward_train(self.handle, prec, ptr(x), B, H, W, 0, 1, 0, 1, 0, 0, None, ptr(pose), None,
                                               ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_dfnet_forward_train")
        self._remember(ws)
        return pose, ws

    def backward_params(self, x, grad_pose, precision=None, tape=None):
        """Gradients of the pose-regression path w.r.t. its parameters: dict {state_dict key: tensor} for
        encoder.<k>.weight|bias (13 convs) and fc_pose.weight|bias, from d L/d pose [B, feat_dim].  tape: the state
        forward_pose_keep / forward_train(keep=True) left (no forward recompute)."""
        x, gp = _f32c(x), _f32c(grad_pose).reshape(x.shape[0], self.feat_dim)
        B, C, H, W = x.shape
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        chans, cin, names, grads = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512], 3, [], []
        for idx, co in zip(self.CONV_INDEX, chans):
            names += [f"encoder.{idx}.weight", f"encoder.{idx}.bias"]
            grads += [torch.empty(co, cin, 3, 3, device=dev), torch.empty(co, device=dev)]
            cin = co
        names += ["fc_pose.weight", "fc_pose.bias"]
        grads += [torch.empty(self.feat_dim, 512, device=dev), torch.empty(self.feat_dim, device=dev)]
        ptrs = (ctypes.c_void_p * len(grads))(*[g.data_ptr() for g in grads])
        if tape is not None:
            check(self.lib.dfn_dfnet_backward_all_params(self.handle, prec, ptr(x), B, H, W, ptr(gp), None, 0, 0, 0, 0, 1, ptrs, len(grads),
                                                         ctypes.c_void_p(tape.data_ptr()), tape.numel(), current_stream()),
                  "dfn_dfnet_backward_all_params")
            return dict(zip(names, grads))
        nbytes = self.lib.dfn_dfnet_backward_params_workspace_bytes(self.handle, prec, B, H, W)
        self._ws = _grow_workspace(self._ws, nbytes, dev)
        check(self.lib.dfn_dfnet_backward_params(self.handle, prec, ptr(x), B, H, W, ptr(gp), ptrs, len(grads),
                                                 ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(), current_stream()),
              "dfn_dfnet_backward_params")
        return dict(zip(names, grads))

    def forward_train(self, x, isSingleStream=False, return_pose=True, bn_batch=True, upsampleH=240, upsampleW=427,
                      precision=None, keep=False):
        """DFNet.forward while the module is being trained: (features, pose, bn_stats).  bn_batch: BatchNorm on the
        statistics of this batch (train() mode) — bn_stats [n_taps, 2, 128] = batch mean, biased variance — or on its
        running statistics (--freezeBN; bn_stats None).  keep=True: returns (features, pose, bn_stats, tape) — the tape
        holds every activation for backward_all_params(tape=...), which then recomputes nothing."""
        x = _f32c(x)
        B, C, H, W = x.shape
        assert C == 3
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        shape = (self.n_taps, B, 128, upsampleH, upsampleW) if isSingleStream else (2, self.n_taps, B // 2, 128, upsampleH, upsampleW)
        feats = torch.empty(shape, device=dev)
        pose = torch.empty(B, self.feat_dim, device=dev) if return_pose else None
        stats = torch.empty(self.n_taps, 2, 128, device=dev) if bn_batch else None
        if keep:
            ws = torch.empty(self.lib.dfn_dfnet_backward_params_workspace_bytes(self.handle, prec, B, H, W), dtype=torch.uint8, device=dev)
        else:
            nbytes = self.lib.dfn_dfnet_workspace_bytes(self.handle, prec, B, H, W)
            self._ws = _grow_workspace(self._ws, nbytes, dev)
            ws = self._ws
        check(self.lib.dfn_dfnet_forward_train(self.handle, prec, ptr(x), B, H, W, int(not isSingleStream), int(return_pose),
                                               int(bool(bn_batch)), int(bool(keep)), int(upsampleH), int(upsampleW), ptr(feats), ptr(pose),
                                               ptr(stats), ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_dfnet_forward_train")
        if not isSingleStream:
            feats = (feats[0], feats[1])
        if keep:
            self._remember(ws)
        return (feats, pose, stats, ws) if keep else (feats, pose, stats)

    # ------------------------------------------------------------------ training forward that keeps the pyramid, enlarges nothing
    def forward_train_pyramid(self, x, return_pose=True, bn_batch=True, precision=None, feature_images=None):
        """The siamese training forward (x = cat([stream A, stream B])) WITHOUT the enlarged feature stacks: every level's adapted
        map stays at its own resolution in the tape; triplet_pyramid_forward / backward_all_params_triplet work from there
        (dfn_dfnet_forward_train_pyramid).  feature_images: the leading frames that are the siamese pair (default: all); frames
        beyond them run the encoder and the pose head only.  Returns (pose | None, bn_stats | None, tape)."""
        x = _f32c(x)
        B, C, H, W = x.shape
        nf = B if feature_images is None else int(feature_images)
        assert C == 3 and nf % 2 == 0 and 2 <= nf <= B
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        pose = torch.empty(B, self.feat_dim, device=dev) if return_pose else None
        stats = torch.empty(self.n_taps, 2, 128, device=dev) if bn_batch else None
        ws = torch.empty(self.lib.dfn_dfnet_backward_params_workspace_bytes(self.handle, prec, B, H, W), dtype=torch.uint8, device=dev)
        check(self.lib.dfn_dfnet_forward_train_pyramid(self.handle, prec, ptr(x), B, nf, H, W, int(return_pose), int(bool(bn_batch)), ptr(pose),
                                                       ptr(stats), ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_dfnet_forward_train_pyramid")
        self._remember(ws)
        return pose, stats, ws

    def triplet_pyramid_forward(self, tape, B, H, W, upH, upW, f1_half, margin, mining, precision=None, feature_images=None):
        """Triplet loss (mining 0 / 1 / 2 = misc.py:355 / :371 / :399) between the two halves of the kept siamese batch as enlarged
        to [upH, upW], from the low-resolution levels: (loss 0-dim device tensor, state for backward_all_params_triplet)."""
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = tape.device
        nf = B if feature_images is None else int(feature_images)
        nbytes = self.lib.dfn_dfnet_triplet_pyramid_state_bytes(self.handle, nf, int(upH))
        state = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        loss = torch.empty(1, device=dev)
        check(self.lib.dfn_dfnet_triplet_pyramid_forward(self.handle, prec, B, nf, H, W, int(upH), int(upW), int(f1_half), float(margin), int(mining),
                                                         ptr(loss), ctypes.c_void_p(state.data_ptr()), nbytes,
                                                         ctypes.c_void_p(tape.data_ptr()), tape.numel(), current_stream()),
              "dfn_dfnet_triplet_pyramid_forward")
        return loss.reshape(()), state

    def backward_all_params_triplet(self, x, grad_pose, grad_loss, state, f1_half, upH, upW, bn_batch, tape, precision=None,
                                    feature_images=None):
        """backward_all_params whose feature gradient is grad_loss (0-dim / [1] device tensor) x the gradient of the pyramid triplet
        loss recorded in `state`; dict keyed as train_param_names(bn_affine=bn_batch)."""
        x = _f32c(x)
        B, C, H, W = x.shape
        gp = None if grad_pose is None else _f32c(grad_pose).reshape(B, self.feat_dim)
        gl = _f32c(grad_loss).reshape(1)
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        names = self.train_param_names(bn_affine=bn_batch)
        grads = [torch.empty(sh, device=dev) for sh in self._train_grad_shapes(bn_batch)]
        ptrs = (ctypes.c_void_p * len(grads))(*[t.data_ptr() for t in grads])
        nf = B if feature_images is None else int(feature_images)
        check(self.lib.dfn_dfnet_backward_all_params_triplet(self.handle, prec, ptr(x), B, nf, H, W, ptr(gp), ptr(gl),
                                                             ctypes.c_void_p(state.data_ptr()), state.numel(), int(f1_half), int(upH), int(upW),
                                                             int(bool(bn_batch)), ptrs, len(grads), ctypes.c_void_p(tape.data_ptr()),
                                                             tape.numel(), current_stream()),
              "dfn_dfnet_backward_all_params_triplet")
        return dict(zip(names, grads))

    def _train_grad_shapes(self, bn_batch):
        """Shapes of the gradient tensors in train_param_names(bn_affine=bn_batch) order."""
        chans, cin, shapes = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512], 3, []
        for co in chans:
            shapes += [(co, cin, 3, 3), (co,)]
            cin = co
        shapes += [(self.feat_dim, 512), (self.feat_dim,)]
        for t, c in zip(range(self.n_taps), (64, 256, 512)):
            shapes += [(64, c, 1, 1), (64,), (128, 64, 5, 5), (128,)] + ([(128,), (128,)] if bn_batch else [])
        return shapes

    def train_param_names(self, bn_affine=True):
        """state_dict keys in the order of backward_all_params' gradients."""
        names = []
        for idx in self.CONV_INDEX:
            names += [f"encoder.{idx}.weight", f"encoder.{idx}.bias"]
        names += ["fc_pose.weight", "fc_pose.bias"]
        for t in range(self.n_taps):
            pre = f"adaptation_layers.adapt_layer_{t}"
            names += [f"{pre}.0.weight", f"{pre}.0.bias", f"{pre}.2.weight", f"{pre}.2.bias"]
            if bn_affine:
                names += [f"{pre}.3.weight", f"{pre}.3.bias"]
        return names

    def backward_all_params(self, x, grad_pose, grad_features, levels=None, bn_batch=False, precision=None, tape=None):
        """Gradients of BOTH heads w.r.t. every trained parameter (run_feature.py:166-230): dict keyed as
        train_param_names(bn_affine=bn_batch).  bn_batch False: BatchNorm frozen on its running statistics
        (--freezeBN); True: batch statistics, with the gradients of BatchNorm's weight and bias.
        grad_pose [B, feat_dim] or None; grad_features single-stream [n_taps, B, 128, uH, uW]."""
        x, g = _f32c(x), _f32c(grad_features)
        B, C, H, W = x.shape
        assert C == 3 and g.shape[:3] == (self.n_taps, B, 128), (x.shape, g.shape)
        gp = None if grad_pose is None else _f32c(grad_pose).reshape(B, self.feat_dim)
        mask = sum(1 << int(t) for t in (range(self.n_taps) if levels is None else levels))
        prec = _lib.PRECISIONS[precision or self.precision]
        dev = x.device
        names = self.train_param_names(bn_affine=bn_batch)
        shapes = self._train_grad_shapes(bn_batch)
        n_pose = 2 * 13 + 2
        # zeros for the adaptation layers: levels outside the mask are not written
        grads = [torch.empty(sh, device=dev) if i < n_pose else torch.zeros(sh, device=dev) for i, sh in enumerate(shapes)]
        ptrs = (ctypes.c_void_p * len(grads))(*[t.data_ptr() for t in grads])
        if tape is None:
            nbytes = self.lib.dfn_dfnet_backward_params_workspace_bytes(self.handle, prec, B, H, W)
            self._ws = _grow_workspace(self._ws, nbytes, dev)
        ws = self._ws if tape is None else tape
        check(self.lib.dfn_dfnet_backward_all_params(self.handle, prec, ptr(x), B, H, W, ptr(gp), ptr(g), g.shape[3], g.shape[4], mask,
                                                     int(bool(bn_batch)), int(tape is not None), ptrs, len(grads),
                                                     ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_dfnet_backward_all_params")
        return dict(zip(names, grads))

    def refresh_train_params_device(self, tensors, precisions=None):
        """Re-pack every tensor the training path reads from device tensors: train_param_names(True) with, per level,
        .3.running_mean and .3.running_var after .3.bias.  The folded inference weights are NOT updated."""
        ts = [_f32c(t) for t in tensors]
        assert len(ts) == 28 + 8 * self.n_taps and all(t.is_cuda for t in ts)
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        check(self.lib.dfn_dfnet_refresh_train_params_device(self.handle, ptrs, len(ts), self._prec_mask(precisions), current_stream()),
              "dfn_dfnet_refresh_train_params_device")

    def _prec_mask(self, precisions):
        """Bit mask of the precisions a device re-pack renews: the engine's own by default, 'all', or a list of names."""
        if precisions == 'all':
            return 7
        names = [self.precision] if precisions is None else list(precisions)
        return sum(1 << _lib.PRECISIONS[p] for p in set(names))

    def refresh_pose_params_device(self, tensors, precisions=None):
        """Re-pack the pose path's parameters from device tensors (order: encoder.<k>.weight, .bias for the 13 convs,
        fc_pose.weight, fc_pose.bias) — the fast path after an optimizer step."""
        ts = [_f32c(t) for t in tensors]
        assert len(ts) == 28 and all(t.is_cuda for t in ts)
        ptrs = (ctypes.c_void_p * len(ts))(*[t.data_ptr() for t in ts])
        check(self.lib.dfn_dfnet_refresh_pose_params_device(self.handle, ptrs, len(ts), self._prec_mask(precisions), current_stream()),
              "dfn_dfnet_refresh_pose_params_device")

    def backward_input(self, x, grad_features, levels=None, precision=None):
        """d L/d x [B,3,H,W] from d L/d features in the single-stream layout [n_taps,B,128,uH,uW]; `levels` lists the
        pyramid levels that carry gradient (default: all).  Weights are frozen (DFNet_dm's feat_model)."""
        x, g = _f32c(x), _f32c(grad_features)
        B, C, H, W = x.shape
        assert C == 3 and g.shape[:3] == (self.n_taps, B, 128), (x.shape, g.shape)
        mask = sum(1 << int(t) for t in (range(self.n_taps) if levels is None else levels))
        prec = _lib.PRECISIONS[precision or self.precision]
        gx = torch.empty_like(x)
        nbytes = self.lib.dfn_dfnet_backward_workspace_bytes(self.handle, prec, B, H, W)
        self._ws = _grow_workspace(self._ws, nbytes, x.device)
        check(self.lib.dfn_dfnet_backward_input(self.handle, prec, ptr(x), B, H, W, g.shape[3], g.shape[4], ptr(g), mask,
                                                ptr(gx), ctypes.c_void_p(self._ws.data_ptr()), self._ws.numel(),
                                                current_stream()), "dfn_dfnet_backward_input")
        return gx


# ---------------------------------------------------------------------- weight-free stage wrappers
def raygen(H, W, focal, c2w, want_viewdirs=True):
    lib = _lib.load()
    c2w = _f32c(c2w)[:3, :4].contiguous()
    dev = c2w.device
    o = torch.empty(H, W, 3, device=dev)
    d = torch.empty(H, W, 3, device=dev)
    v = torch.empty(H, W, 3, device=dev) if want_viewdirs else None
    check(lib.dfn_raygen(H, W, float(focal), ptr(c2w), ptr(o), ptr(d), ptr(v), current_stream()), "dfn_raygen")
    return o, d, v


def raygen_frames(H, W, focal, c2ws, want_viewdirs=True):
    """get_rays for the B poses c2ws [B,3,4] in one launch: rays_o, rays_d (, viewdirs) [B,H,W,3]."""
    c2ws = _f32c(c2ws)[:, :3, :4].contiguous()
    B, dev = c2ws.shape[0], c2ws.device
    o = torch.empty(B, H, W, 3, device=dev)
    d = torch.empty(B, H, W, 3, device=dev)
    v = torch.empty(B, H, W, 3, device=dev) if want_viewdirs else None
    check(_lib.load().dfn_raygen_frames(B, H, W, float(focal), ptr(c2ws), ptr(o), ptr(d), ptr(v), current_stream()), "dfn_raygen_frames")
    return o, d, v


def raygen_frames_backward(H, W, focal, grad_o, grad_d):
    """d L/d c2w [B,3,4] from d L/d rays_o, rays_d [B,H*W,3]."""
    grad_o, grad_d = _f32c(grad_o), _f32c(grad_d)
    B = grad_o.shape[0]
    gc = torch.empty(B, 3, 4, device=grad_o.device)
    check(_lib.load().dfn_raygen_frames_backward(B, int(H), int(W), float(focal), ptr(grad_o), ptr(grad_d), ptr(gc), current_stream()),
          "dfn_raygen_frames_backward")
    return gc


def upsample_bicubic_frames(imgs, outH, outW, nchw=False):
    """[B,H,W,C] -> [B,outH,outW,C] (nchw: [B,C,outH,outW]) in one launch (upsample_bicubic per frame)."""
    imgs = _f32c(imgs)
    B, H, W, C = imgs.shape
    out = torch.empty((B, C, int(outH), int(outW)) if nchw else (B, int(outH), int(outW), C), device=imgs.device)
    check(_lib.load().dfn_upsample_bicubic_frames(ptr(imgs), B, H, W, C, int(outH), int(outW), 1 if nchw else 0, ptr(out), current_stream()),
          "dfn_upsample_bicubic_frames")
    return out


def upsample_bicubic_frames_backward(grad_out, H, W, nchw=False):
    """Adjoint of upsample_bicubic_frames: [B,outH,outW,C] (nchw: [B,C,outH,outW]) -> [B,H,W,C]."""
    g = _f32c(grad_out)
    if nchw:
        B, C, outH, outW = g.shape
    else:
        B, outH, outW, C = g.shape
    out = torch.empty(B, int(H), int(W), C, device=g.device)
    check(_lib.load().dfn_upsample_bicubic_frames_backward(ptr(g), B, int(H), int(W), C, outH, outW, 1 if nchw else 0, ptr(out),
                                                           current_stream()), "dfn_upsample_bicubic_frames_backward")
    return out


def posenc(x, L, fast=False):
    lib = _lib.load()
    x = _f32c(x).reshape(-1, 3)
    out = torch.empty(x.shape[0], 3 + 6 * L, device=x.device)
    check(lib.dfn_posenc(ptr(x), x.shape[0], L, 1 if fast else 0, ptr(out), current_stream()), "dfn_posenc")
    return out


def coarse_weights(sigma, z):
    lib = _lib.load()
    sigma, z = _f32c(sigma), _f32c(z)
    w = torch.empty_like(sigma)
    check(lib.dfn_coarse_weights(ptr(sigma), ptr(z), sigma.shape[0], sigma.shape[1], ptr(w), current_stream()),
          "dfn_coarse_weights")
    return w


def sample_pdf(bins, weights, Ni, u=None):
    lib = _lib.load()
    bins, weights = _f32c(bins), _f32c(weights)
    if u is not None:
        u = _f32c(u)
    out = torch.empty(bins.shape[0], Ni, device=bins.device)
    check(lib.dfn_sample_pdf(ptr(bins), ptr(weights), bins.shape[0], bins.shape[1], Ni, ptr(u), ptr(out),
                             current_stream()), "dfn_sample_pdf")
    return out


def ndc_rays(H, W, focal, near, rays_o, rays_d):
    """ray_utils.ndc_rays (models/ray_utils.py:27-46) on the device: (rays_o, rays_d) in normalised device coordinates."""
    lib = _lib.load()
    o, d = _f32c(rays_o), _f32c(rays_d)
    oo, od = torch.empty_like(o), torch.empty_like(d)
    check(lib.dfn_ndc_rays(int(H), int(W), float(focal), float(near), ptr(o), ptr(d), o.numel() // 3, ptr(oo), ptr(od), current_stream()),
          "dfn_ndc_rays")
    return oo, od


def sample_fine(sigma, Ni, near, far, want_aux=False, lindisp=False):
    lib = _lib.load()
    sigma = _f32c(sigma)
    n, Nc = sigma.shape
    z = torch.empty(n, Nc + Ni, device=sigma.device)
    w = torch.empty(n, Nc, device=sigma.device) if want_aux else None
    zs = torch.empty(n, Ni, device=sigma.device) if want_aux else None
    check(lib.dfn_sample_fine_opt(ptr(sigma), n, Nc, Ni, float(near), float(far), _lib.RENDER_LINDISP if lindisp else 0, ptr(z), ptr(w),
                                  ptr(zs), current_stream()), "dfn_sample_fine")
    return (z, w, zs) if want_aux else z


def composite_fine(raw, z, beta_min=0.1, test_time=True, static_only=True, white_bkgd=False, want_aux=False):
    lib = _lib.load()
    raw, z = _f32c(raw), _f32c(z)
    n, Nf = z.shape
    dev = raw.device
    rgb, disp, acc = torch.empty(n, 3, device=dev), torch.empty(n, device=dev), torch.empty(n, device=dev)
    depth = torch.empty(n, device=dev) if want_aux else None
    w = torch.empty(n, Nf, device=dev) if want_aux else None
    beta = torch.empty(n, device=dev) if want_aux else None
    flags = (_lib.COMP_TEST_TIME if test_time else 0) | (_lib.COMP_STATIC_ONLY if static_only else 0) | \
            (_lib.COMP_WHITE_BKGD if white_bkgd else 0)
    check(lib.dfn_composite_fine(ptr(raw), ptr(z), n, Nf, float(beta_min), flags, ptr(rgb), ptr(disp), ptr(acc),
                                 ptr(depth), ptr(w), ptr(beta), current_stream()), "dfn_composite_fine")
    out = dict(rgb=rgb, disp=disp, acc=acc)
    if want_aux:
        out.update(depth=depth, weights=w, beta=beta)
    return out


def upsample_bicubic(img, outH, outW, out=None):
    """[H,W,C] -> [outH,outW,C], torch's nn.Upsample(mode='bicubic') semantics (align_corners=False)."""
    lib = _lib.load()
    img = _f32c(img)
    H, W, C = img.shape
    if out is None:
        out = torch.empty(outH, outW, C, device=img.device)
    check(lib.dfn_upsample_bicubic(ptr(img), H, W, C, int(outH), int(outW), ptr(out), current_stream()),
          "dfn_upsample_bicubic")
    return out


def raygen_backward(H, W, focal, grad_o, grad_d):
    """d L/d c2w [3,4] from d L/d rays_o, rays_d [H*W,3] (get_rays backward)."""
    grad_o, grad_d = _f32c(grad_o).reshape(-1, 3), _f32c(grad_d).reshape(-1, 3)
    gc = torch.empty(3, 4, device=grad_o.device)
    check(_lib.load().dfn_raygen_backward(int(H), int(W), float(focal), ptr(grad_o), ptr(grad_d), ptr(gc), current_stream()),
          "dfn_raygen_backward")
    return gc


def upsample_bicubic_backward(grad_out, H, W, out=None):
    """Adjoint of upsample_bicubic: [outH,outW,C] -> [H,W,C]."""
    g = _f32c(grad_out)
    outH, outW, C = g.shape
    if out is None:
        out = torch.empty(H, W, C, device=g.device)
    check(_lib.load().dfn_upsample_bicubic_backward(ptr(g), int(H), int(W), C, outH, outW, ptr(out), current_stream()),
          "dfn_upsample_bicubic_backward")
    return out


def composite_fine_backward(raw, z, grad_rgb):
    """d L/d raw [n,Nf,9] from d L/d rgb [n,3] through the test-time fine compositing (rgb only)."""
    raw, z, grad_rgb = _f32c(raw), _f32c(z), _f32c(grad_rgb)
    n, Nf = z.shape
    graw = torch.empty(n, Nf, 9, device=raw.device)
    check(_lib.load().dfn_composite_fine_backward(ptr(raw), ptr(z), ptr(grad_rgb), n, Nf, ptr(graw), current_stream()),
          "dfn_composite_fine_backward")
    return graw


def frame_prep(rgb_u8, H, W, hist_bins=10):
    """Dataset front-end of one frame on the device (dfn_frame_prep; seven_scenes.py:324-352): rgb_u8 CUDA uint8
    [h, w, 3] -> (img fp32 [3, H, W] in [0, 1], INTER_AREA-downscaled; hist fp32 [hist_bins], NeRF-H's histogram index
    vector)."""
    assert rgb_u8.is_cuda and rgb_u8.dtype == torch.uint8 and rgb_u8.dim() == 3 and rgb_u8.shape[2] == 3
    rgb_u8 = rgb_u8.contiguous()
    lib = _lib.load()
    h, w = int(rgb_u8.shape[0]), int(rgb_u8.shape[1])
    img = torch.empty(3, int(H), int(W), device=rgb_u8.device)
    hist = torch.empty(int(hist_bins), device=rgb_u8.device)
    scratch = torch.empty(lib.dfn_frame_prep_scratch_bytes(), dtype=torch.uint8, device=rgb_u8.device)
    check(lib.dfn_frame_prep(ctypes.c_void_p(rgb_u8.data_ptr()), h, w, int(H), int(W), int(hist_bins), ptr(img), ptr(hist),
                             ctypes.c_void_p(scratch.data_ptr()), current_stream()), "dfn_frame_prep")
    return img, hist


def frame_post(rgb, disp, gt=None, want_gt8=True):
    """render_path's per-frame back-end on the device (dfn_frame_post; models/rendering.py:423-452): rgb [n,H,W,3], disp [n,H,W]
    fp32 CUDA, gt None | [n,H,W,3] | [H,W,3] (one ground-truth frame for every render) -> dict(rgb8, disp8 uint8 CUDA,
    gt8 | None, mse fp32 [n] | None, disp_max fp32 [n])."""
    lib = _lib.load()
    rgb, disp = _f32c(rgb), _f32c(disp)
    n, H, W = disp.shape
    dev = rgb.device
    per_frame = gt is not None and gt.dim() == 4
    if gt is not None:
        gt = _f32c(gt)
    rgb8 = torch.empty(n, H, W, 3, dtype=torch.uint8, device=dev)
    disp8 = torch.empty(n, H, W, dtype=torch.uint8, device=dev)
    gt8 = None
    if gt is not None and want_gt8:
        gt8 = torch.empty((n, H, W, 3) if per_frame else (H, W, 3), dtype=torch.uint8, device=dev)
    mse = torch.empty(n, device=dev) if gt is not None else None
    dmax = torch.empty(n, device=dev)
    scratch = torch.empty(lib.dfn_frame_post_scratch_bytes(n), dtype=torch.uint8, device=dev)
    raw = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    check(lib.dfn_frame_post(ptr(rgb), ptr(disp), ptr(gt), 1 if per_frame else 0, n, H, W, raw(rgb8), raw(disp8), raw(gt8), ptr(mse),
                             ptr(dmax), raw(scratch), current_stream()), "dfn_frame_post")
    return {"rgb8": rgb8, "disp8": disp8, "gt8": gt8, "mse": mse, "disp_max": dmax}

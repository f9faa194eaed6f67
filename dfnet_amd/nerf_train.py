"""NeRF-H training on the HIP path: host side of dfn_nerfh_train_forward / dfn_nerfw_loss /
dfn_nerfh_train_backward.

Mirrors what /root/reference/script/run_nerf.py:32-80 does per step — render(**render_kwargs_train) -> NerfWLoss ->
loss.backward() -> optimizer.step() — with every network product on the exact-fp32 MFMA training kernels.  The
parameters stay torch's master weights (the kernels read the nn.Module tensors in place and write into .grad), so the
optimizer (torch.optim.Adam, as the reference) needs no re-pack; the packed test-time engine is refreshed only when a
validation render is asked for (`HipQuery.refresh`).

Two equivalent surfaces:
  * NerfHTrainer.train_step(...)  — forward + fused NerfWLoss + backward, three library calls (run_nerf.py uses this);
  * rendering.render(..., **render_kwargs_train) — returns tensors attached to autograd (`_RenderTrainFn`), so the
    reference's own loop shape (loss_func(results, target); loss.backward()) runs unchanged.
The reference's random draws (stratified jitter, coarse-density noise, importance-sampling u) are drawn with torch on
the device and handed to the library as inputs, which is what makes the path checkable against the reference.
"""
import ctypes
import warnings

import torch

from . import _lib
from ._lib import check, current_stream, ptr


def _f32c(t):
    return t.contiguous().float()


class NerfHTrainer:
    """Owns the workspace of one training step and the canonical parameter list of (network_fn, network_fine,
    embedding_a, embedding_t)."""

    def __init__(self, engine, network_fn, network_fine, embedding_a, embedding_t):
        self.lib = _lib.load()
        self.engine = engine            # NerfHEngine: the handle carries the network geometry
        self.hist_bin = engine.hist_bin
        n = self.lib.dfn_nerfh_train_param_count()
        names = [self.lib.dfn_nerfh_train_param_name(i).decode() for i in range(n)]
        lookup = {}
        for pre, mod in (("coarse.", network_fn), ("fine.", network_fine), ("embedding_a.", embedding_a), ("embedding_t.", embedding_t)):
            for k, p in mod.named_parameters():
                lookup[pre + k] = p
        missing = [k for k in names if k not in lookup]
        if missing or len(lookup) != len(names):
            raise ValueError(f"NerfHTrainer: parameter set does not match the NeRF-H layout (missing {missing[:3]}...)")
        self.names = names
        self.params = [lookup[k] for k in names]
        for k, p in zip(names, self.params):
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError(f"NerfHTrainer: parameter {k} must be a contiguous fp32 CUDA tensor")
        self._ws = None
        self._saved = None
        self.exact = False   # default implementation of the step (fused at netwidth 128; other widths always run the exact one)
        self.fused_split = False   # fused step: keep hi | lo planes of the FINE network's stored operands too (DFN_TRAIN_FUSED_SPLIT)
        # The fused step splits the LIVE master weights at the operand scale of the last dfn_nerfh_commit (64x headroom): a weight that
        # outgrows it saturates in the packed blob and the chains OR a bit into the step's range word.  The library then leaves
        # ZEROS in every gradient tensor of that step (csrc/nerfh_fused_train.h: GuardArgs — a skipped step, as a loss scaler skips
        # one) and passes the bits on to the handle's range flag.  What train_step() does about it (`range_check`):
        #   "skip"   (default) reads the flag WITHOUT draining the stream (dfn_nerfh_range_status_async into pinned memory, looked at
        #            one or two steps later): the flagged step's gradients stay zero — the caller's optimizer still steps on them
        #            (Adam coasts on its moments: not a loss-scaler style no-op; see train_step) — and the operand scale is re-committed
        #            at the live weights (the exact-fp32 step takes over if a step right after a re-commit is flagged again);
        #   "repeat" waits for the stream after every `range_check_every`-th step and, if flagged, re-commits and REPEATS the step
        #            (then on the exact-fp32 path if that does not clear it): p.grad always holds the step's gradients;
        #   None     never looks (flagged steps are still skipped by the library).
        self.range_check = "skip"
        self.range_check_every = 1
        self.range_recoveries = 0
        self.flagged_steps = 0     # steps whose gradients the range guard zeroed (range_check = "skip")
        self._steps = 0
        self._pending = []          # ("skip"): [(event, pinned int32 slot)] of flag reads still in flight
        self._slots = None
        self._last_recommit_step = -10

    # ------------------------------------------------------------------ plumbing
    def _ptr_array(self, tensors):
        return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    def _workspace(self, n, Nc, Ni, dev):
        nbytes = self.lib.dfn_nerfh_train_workspace_bytes(self.engine.handle, n, Nc, Ni)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != dev:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        return self._ws

    @staticmethod
    def draw(n, Nc, Ni, perturb, dev, generator=None):
        """The three draws of one render_rays call, in the reference's order (rendering.py:282, :173, :35)."""
        t_rand = torch.rand(n, Nc, device=dev, generator=generator) if perturb > 0. else None
        noise = torch.randn(n, Nc, device=dev, generator=generator)   # drawn even when raw_noise_std = 0 (quirk Q4)
        u = torch.rand(n, Ni, device=dev, generator=generator) if perturb > 0. else None
        return t_rand, noise, u

    # ------------------------------------------------------------------ the three calls
    def set_mode(self, exact):
        """DFN_TRAIN_FUSED (netwidth 128: the register-resident chains of csrc/nerfh_fused_*.hip) or DFN_TRAIN_EXACT (layer by layer,
        exact fp32, activations kept: what backward_rays() needs)."""
        check(self.lib.dfn_nerfh_set_train_mode(self.engine.handle, 1 if exact else (2 if self.fused_split else 0)), "dfn_nerfh_set_train_mode")

    def forward(self, rays_o, rays_d, hist, Nc, Ni, near, far, t_rand=None, noise=None, raw_noise_std=0., u=None, exact=None):
        """Training-mode render_rays -> dict(rgb_map, disp_map, acc_map, raw, rgb0, disp0, acc0, z_std, beta,
        transient_sigmas); keeps what backward() needs.  exact: force the layer-by-layer exact-fp32 step (None: `self.exact`)."""
        exact = self.exact if exact is None else bool(exact)
        self.set_mode(exact)
        exact = exact or self.engine.width != 128    # what RAN: only netwidth 128 has the fused chains, any other width is the exact step
        rays_o, rays_d = _f32c(rays_o).reshape(-1, 3), _f32c(rays_d).reshape(-1, 3)
        n, dev = rays_o.shape[0], rays_o.device
        hist = _f32c(hist).reshape(-1, self.hist_bin)
        if hist.shape[0] not in (1, n):
            raise ValueError(f"img_idx must have 1 or {n} rows of {self.hist_bin} bins")
        t_rand, noise, u = (None if t is None else _f32c(t) for t in (t_rand, noise, u))
        if noise is not None and float(raw_noise_std) == 0.:
            noise = None   # adds exactly zero: skip the read
        Nf = Nc + Ni
        out = {k: torch.empty(n, *sh, device=dev) for k, sh in (("rgb_map", (3,)), ("disp_map", ()), ("acc_map", ()), ("raw", (Nf, 9)),
                                                                 ("rgb0", (3,)), ("disp0", ()), ("acc0", ()), ("z_std", ()), ("beta", ()))}
        ws = self._workspace(n, Nc, Ni, dev)
        check(self.lib.dfn_nerfh_train_forward(self.engine.handle, self._ptr_array(self.params), ptr(rays_o), ptr(rays_d), ptr(hist),
                                               hist.shape[0], n, Nc, Ni, float(near), float(far), ptr(t_rand), ptr(noise),
                                               float(raw_noise_std), ptr(u), ptr(out["rgb_map"]), ptr(out["disp_map"]), ptr(out["acc_map"]),
                                               ptr(out["raw"]), ptr(out["rgb0"]), ptr(out["disp0"]), ptr(out["acc0"]), ptr(out["z_std"]),
                                               ptr(out["beta"]), ctypes.c_void_p(ws.data_ptr()), ws.numel(), current_stream()),
              "dfn_nerfh_train_forward")
        out["transient_sigmas"] = out["raw"][..., 7]
        self._saved = dict(hist=hist, n=n, Nc=Nc, Ni=Ni, noise=noise, raw_noise_std=float(raw_noise_std), raw=out["raw"], ws=ws, exact=exact,
                           rays_o=rays_o, rays_d=rays_d)
        return out

    def loss(self, out, target, coef=1., lambda_u=0.01):
        """NerfWLoss forward + gradient in one kernel -> (loss5 tensor [c_l, f_l, b_l, s_l, psnr], (g_rgb, g_rgb0, g_beta),
        constant d L / d transient_sigma)."""
        target = _f32c(target).reshape(-1, 3)
        n, Nf = out["raw"].shape[0], out["raw"].shape[1]
        dev = target.device
        loss_buf = torch.empty(160, device=dev)   # DFN_NERFW_LOSS_FLOATS: 5 results + reduction scratch
        loss5 = loss_buf[:5]
        g_rgb, g_rgb0, g_beta = torch.empty(n, 3, device=dev), torch.empty(n, 3, device=dev), torch.empty(n, device=dev)
        check(self.lib.dfn_nerfw_loss(ptr(out["rgb_map"]), ptr(out["rgb0"]), ptr(out["beta"]), ptr(out["raw"]), ptr(target), n, Nf,
                                      float(coef), float(lambda_u), ptr(loss_buf), ptr(g_rgb), ptr(g_rgb0), ptr(g_beta), current_stream()),
              "dfn_nerfw_loss")
        return loss5, (g_rgb, g_rgb0, g_beta), float(coef) * float(lambda_u) / (n * Nf)

    def backward(self, g_rgb, g_rgb0, g_beta, g_tsigma=0., g_tsigma_dense=None, grads=None, saved=None):
        """Gradients of every parameter from the last forward() (or from `saved`, the state an autograd node took from its own
        forward).  grads=None: written into (freshly allocated) p.grad."""
        s = self._saved if saved is None else saved
        if s is None:
            raise RuntimeError("NerfHTrainer.backward() without a forward()")
        if grads is None:
            for p in self.params:
                if p.grad is None or not p.grad.is_contiguous():
                    p.grad = torch.empty_like(p)
            grads = [p.grad for p in self.params]
        g_rgb, g_rgb0, g_beta = _f32c(g_rgb).reshape(-1, 3), _f32c(g_rgb0).reshape(-1, 3), _f32c(g_beta).reshape(-1)
        gd = None if g_tsigma_dense is None else _f32c(g_tsigma_dense).reshape(s["n"], s["Nc"] + s["Ni"])
        self.set_mode(s["exact"])   # the workspace was laid out by the forward's implementation
        check(self.lib.dfn_nerfh_train_backward(self.engine.handle, self._ptr_array(self.params), ptr(s["hist"]), s["hist"].shape[0], s["n"],
                                                s["Nc"], s["Ni"], ptr(s["noise"]), s["raw_noise_std"], ptr(s["raw"]), ptr(g_rgb), ptr(g_rgb0),
                                                ptr(g_beta), float(g_tsigma), ptr(gd), self._ptr_array(grads),
                                                ctypes.c_void_p(s["ws"].data_ptr()), s["ws"].numel(), current_stream()),
              "dfn_nerfh_train_backward")
        return grads

    def backward_rays(self, g_rgb, g_rgb0, g_beta, g_tsigma=0., g_tsigma_dense=None, saved=None):
        """(d L / d rays_o, d L / d rays_d) [n,3] of the last forward(exact=True): the reference's training render is differentiable
        w.r.t. its rays under autograd (rendering.py:245-337); both networks contribute."""
        s = self._saved if saved is None else saved
        if s is None or not s["exact"]:
            raise RuntimeError("NerfHTrainer.backward_rays() needs a forward(exact=True): the fused chain keeps no activations")
        g_rgb, g_rgb0, g_beta = _f32c(g_rgb).reshape(-1, 3), _f32c(g_rgb0).reshape(-1, 3), _f32c(g_beta).reshape(-1)
        gd = None if g_tsigma_dense is None else _f32c(g_tsigma_dense).reshape(s["n"], s["Nc"] + s["Ni"])
        dev = s["raw"].device
        go, gdir = torch.empty(s["n"], 3, device=dev), torch.empty(s["n"], 3, device=dev)
        nb = self.lib.dfn_nerfh_train_backward_rays_scratch_bytes(s["n"], s["Nc"], s["Ni"])
        scratch = torch.empty(nb, dtype=torch.uint8, device=dev)
        check(self.lib.dfn_nerfh_train_backward_rays(self.engine.handle, self._ptr_array(self.params), ptr(s["rays_o"]), ptr(s["rays_d"]),
                                                     ptr(s["hist"]), s["hist"].shape[0], s["n"], s["Nc"], s["Ni"], ptr(s["noise"]),
                                                     s["raw_noise_std"], ptr(s["raw"]), ptr(g_rgb), ptr(g_rgb0), ptr(g_beta), float(g_tsigma), ptr(gd),
                                                     ptr(go), ptr(gdir), ctypes.c_void_p(s["ws"].data_ptr()), s["ws"].numel(),
                                                     ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), current_stream()),
              "dfn_nerfh_train_backward_rays")
        return go, gdir

    def recommit(self):
        """Re-pack the engine from the live master weights: dfn_nerfh_commit re-derives the split-f16 operand scales, which is also
        what the fused training chains scale the step's weights by (csrc/nerfh_fused_api.hip: train_scale)."""
        sd = {k: p.detach().cpu().numpy() for k, p in zip(self.names, self.params)}
        cut = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        self.engine.load_numpy(cut("coarse."), cut("fine."), sd["embedding_a.weight"], sd["embedding_t.weight"])

    def _post_flag_read(self, dev):
        """Enqueue a read-and-clear of the range flag behind this step's kernels (no wait): pinned slot + event."""
        if self._slots is None:
            self._slots = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(4)]
        slot = self._slots[self._steps % len(self._slots)]
        check(self.lib.dfn_nerfh_range_status_async(self.engine.handle, ctypes.c_void_p(slot.data_ptr()), current_stream()),
              "dfn_nerfh_range_status_async")
        ev = torch.cuda.Event()
        ev.record()
        self._pending.append((ev, slot, self._steps))

    def _drain_flags(self, block_beyond=0):
        """Look at the flag reads that have landed; wait for the oldest ones while more than `block_beyond` are in flight."""
        flagged = None
        while self._pending:
            ev, slot, step = self._pending[0]
            if not ev.query():
                if len(self._pending) <= block_beyond:
                    break
                ev.synchronize()
            self._pending.pop(0)
            if int(slot[0]):
                flagged = (int(slot[0]), step)
        if flagged is None:
            return
        flags, step = flagged
        self.range_recoveries += 1
        self.flagged_steps += 1
        if step - self._last_recommit_step <= 3:
            warnings.warn(f"NerfHTrainer: step {step} left the split-f16 operand range again (flags {flags:#x}) right after a re-commit: "
                          "its gradients were zeroed; the following steps run on the exact-fp32 path", RuntimeWarning)
            self.exact = True
            return
        warnings.warn(f"NerfHTrainer: step {step} of the fused training path left the split-f16 operand range (flags {flags:#x}): its "
                      "gradients were zeroed (a skipped step); re-committing the operand scale at the live weights", RuntimeWarning)
        torch.cuda.current_stream().synchronize()
        self._pending.clear()
        self.recommit()
        self.engine.range_flags()   # the steps enqueued between the flagged one and this re-commit ran at the old scale too
        self._last_recommit_step = self._steps

    def last_step_flagged(self):
        """True if the step just taken raised the range flag (its gradients are zeros): waits for the flag read of that step —
        for loops that would rather skip `optimizer.step()` than let Adam coast (range_check = "skip" without this call never waits)."""
        if not self._pending:
            return False
        ev, slot, _ = self._pending[-1]
        ev.synchronize()
        return bool(int(slot[0]))

    def flush_range_check(self):
        """Wait for the flag reads in flight and act on them (end of a training loop / before a checkpoint)."""
        self._drain_flags(block_beyond=0)
        return self.range_recoveries

    def train_step(self, rays_o, rays_d, hist, target, Nc, Ni, near, far, perturb=1., raw_noise_std=0., draws=None, coef=1.,
                   lambda_u=0.01):
        """run_nerf.py:50-66 without the optimizer: forward, NerfWLoss, backward into p.grad.  Returns (loss dict of 0-dim
        tensors c_l/f_l/b_l/s_l, psnr, render outputs).  A fused step whose operands left the split-f16 range (range flag) leaves
        ZERO gradients (the library's guard: p.grad never holds clamped gradients); `self.range_check` decides what follows:
          "repeat"  the flag is read after the step (a stream drain, ~0.35 ms): the step is repeated at a re-derived operand scale and,
                    if that does not clear it, on the exact-fp32 step — no update is ever taken from a flagged step;
          "skip"    the flag is read WITHOUT draining the stream and is known one or two steps later.  Until then the caller's
                    optimizer steps on the zero gradients — NOT a no-op under Adam, which coasts on its moments for those steps (and in
                    a data-parallel run the flagged rank dilutes the averaged gradient) — then the scale is re-committed;
                    `flagged_steps` counts them, `last_step_flagged()` tells (with a drain) whether the step just taken was one.
        The event is rare (64x headroom at commit) and both modes are loud (RuntimeWarning)."""
        n = rays_o.reshape(-1, 3).shape[0]
        t_rand, noise, u = draws if draws is not None else self.draw(n, Nc, Ni, perturb, rays_o.device)

        def run(exact):
            out = self.forward(rays_o, rays_d, hist, Nc, Ni, near, far, t_rand, noise, raw_noise_std, u, exact=exact)
            loss5, (g_rgb, g_rgb0, g_beta), g_ts = self.loss(out, target, coef, lambda_u)
            self.backward(g_rgb, g_rgb0, g_beta, g_ts)
            return out, loss5

        if self.range_check == "skip":
            self._drain_flags(block_beyond=1)
        out, loss5 = run(None)
        every = int(self.range_check_every or 0)
        if self.range_check == "skip":
            if not self._saved["exact"]:
                self._post_flag_read(rays_o.device)
        elif self.range_check == "repeat" and not self._saved["exact"] and every > 0 and self._steps % every == 0:
            flags = self.engine.range_flags()
            if flags:
                self.range_recoveries += 1
                warnings.warn(f"NerfHTrainer: the fused step left the split-f16 operand range (flags {flags:#x}): re-committing the "
                              "operand scale at the live weights and repeating the step", RuntimeWarning)
                self.recommit()
                out, loss5 = run(None)
                if self.engine.range_flags():
                    warnings.warn("NerfHTrainer: still out of range after the re-commit: this step runs on the exact-fp32 path",
                                  RuntimeWarning)
                    out, loss5 = run(True)
        self._steps += 1
        return {k: loss5[i] for i, k in enumerate(("c_l", "f_l", "b_l", "s_l"))}, loss5[4], out


class _RenderTrainFn(torch.autograd.Function):
    """render_rays in training mode as an autograd node: outputs (rgb, disp, acc, raw, rgb0, disp0, acc0, z_std, beta,
    transient_sigmas); gradients flow from rgb, rgb0, beta and transient_sigmas to every parameter (what NerfWLoss
    uses, losses.py:43-52); the other outputs are marked non-differentiable."""

    @staticmethod
    def forward(ctx, trainer, rays_o, rays_d, hist, Nc, Ni, near, far, t_rand, noise, raw_noise_std, u, *params):
        # gradients w.r.t. the rays (pose optimisation through the training render) come from the exact step, which keeps activations
        ctx.want_rays = bool(ctx.needs_input_grad[1] or ctx.needs_input_grad[2])
        out = trainer.forward(rays_o, rays_d, hist, Nc, Ni, near, far, t_rand, noise, raw_noise_std, u, exact=True if ctx.want_rays else None)
        ctx.trainer = trainer
        ctx.saved = trainer._saved
        ts = out["transient_sigmas"].contiguous()
        nd = (out["disp_map"], out["acc_map"], out["raw"], out["disp0"], out["acc0"], out["z_std"])
        ctx.mark_non_differentiable(*nd)
        return (out["rgb_map"], out["disp_map"], out["acc_map"], out["raw"], out["rgb0"], out["disp0"], out["acc0"], out["z_std"],
                out["beta"], ts)

    @staticmethod
    def backward(ctx, g_rgb, _gd, _ga, _graw, g_rgb0, _gd0, _ga0, _gz, g_beta, g_ts):
        tr = ctx.trainer
        if tr._saved is not ctx.saved:
            raise RuntimeError("render(): backward through a training render after another forward reused its workspace")
        n = ctx.saved["n"]
        dev = ctx.saved["raw"].device
        z3, z1 = torch.zeros(n, 3, device=dev), torch.zeros(n, device=dev)
        grads = [torch.empty_like(p) for p in tr.params]
        gs = (z3 if g_rgb is None else g_rgb, z3 if g_rgb0 is None else g_rgb0, z1 if g_beta is None else g_beta)
        g_o = g_d = None
        if ctx.want_rays:   # before the weight gradients: they reuse the gradient buffers of the workspace
            g_o, g_d = tr.backward_rays(*gs, 0., g_ts, saved=ctx.saved)
        if any(ctx.needs_input_grad[12:]):
            tr.backward(*gs, 0., g_ts, grads=grads, saved=ctx.saved)
            if not ctx.saved["exact"] and tr.range_check == "skip":
                tr._post_flag_read(dev)           # found out without a drain, acted on by the next train_step / flush_range_check
            elif not ctx.saved["exact"] and tr.range_check:
                flags = tr.engine.range_flags()   # an autograd node cannot repeat its forward: fail loudly instead of clamped gradients
                if flags:
                    raise _lib.DfnError(f"render(): the fused training step left the split-f16 operand range (flags {flags:#x}); call "
                                        "trainer.recommit() (weights outgrew the committed scale) or set trainer.exact = True")
        else:
            grads = [None] * len(tr.params)
        return (None, g_o if ctx.needs_input_grad[1] else None, g_d if ctx.needs_input_grad[2] else None) + (None,) * 9 + tuple(grads)


def render_train(trainer, rays_o, rays_d, hist, Nc, Ni, near, far, perturb, raw_noise_std, retraw, draws=None):
    """The training branch of rendering.render(): [rgb, disp, acc, extras] with the reference's extras keys."""
    n = rays_o.reshape(-1, 3).shape[0]
    t_rand, noise, u = draws if draws is not None else NerfHTrainer.draw(n, Nc, Ni, float(perturb), rays_o.device)
    (rgb, disp, acc, raw, rgb0, disp0, acc0, z_std, beta, ts) = _RenderTrainFn.apply(
        trainer, rays_o, rays_d, hist, int(Nc), int(Ni), float(near), float(far), t_rand, noise, float(raw_noise_std), u, *trainer.params)
    extras = {'rgb0': rgb0, 'disp0': disp0, 'acc0': acc0, 'z_std': z_std, 'transient_sigmas': ts, 'beta': beta}
    if retraw:
        extras['raw'] = raw
    return [rgb, disp, acc, extras]

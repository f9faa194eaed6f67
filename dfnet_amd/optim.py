"""torch.optim.Adam whose step() is ONE multi-tensor HIP launch (dfn_adam_step).

The reference steps `torch.optim.Adam` after every `loss.backward()` — /root/reference/script/run_nerf.py:65 (created in
models/nerfw.py:436 with betas (0.9, 0.999)), script/run_feature.py:65 / :211, script/train.py:60 with
feature/direct_feature_matching.py:237.  This class IS a `torch.optim.Adam`: same constructor, `param_groups`, `state`
(`step`, `exp_avg`, `exp_avg_sq` per parameter), `state_dict()` / `load_state_dict()` — a checkpoint's `optimizer_state_dict`
moves freely between the two, and `ReduceLROnPlateau` / the manual decay of run_nerf.py:71-73 act on `param_groups` as before.
Only `step()` differs: torch's multi-tensor step is eight to ten launches, each a pass over every tensor; here every element is
read and written once by one launch (csrc/optim_adam.hip: same arithmetic in the same order, bias corrections in double on the
host as torch computes them).

Options the kernel does not implement (amsgrad, maximize, capturable, differentiable, decoupled weight decay, tensor lr / betas,
parameters that are not fp32 CUDA tensors, sparse gradients) run torch's own step: they are torch's features, and no training loop
of the reference uses them."""
import ctypes

import numpy as np
import torch
from torch.autograd.graph import increment_version

from . import _lib
from ._lib import check, current_stream


# dfn_adam_tensor (include/dfnet_hip.h): four pointers, size_t, two floats = 48 bytes
_ADAM_TENSOR = np.dtype([("param", "u8"), ("grad", "u8"), ("exp_avg", "u8"), ("exp_avg_sq", "u8"), ("numel", "u8"),
                         ("step_size", "f4"), ("bias_correction2_sqrt", "f4")])
assert _ADAM_TENSOR.itemsize == ctypes.sizeof(_lib.AdamTensor) == 48
_ONE = torch.tensor(1.0)


class Adam(torch.optim.Adam):
    def _native_ok(self, group):
        if group.get("amsgrad") or group.get("maximize") or group.get("capturable") or group.get("differentiable"):
            return False
        if group.get("decoupled_weight_decay"):
            return False
        b1, b2 = group["betas"]
        if any(torch.is_tensor(x) for x in (group["lr"], b1, b2, group["eps"], group["weight_decay"])):
            return False
        for p in group["params"]:
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_cuda and p.grad.dtype == torch.float32
                    and not p.grad.is_sparse and p.grad.is_contiguous()):
                return False
        return True

    @torch.no_grad()
    def step(self, closure=None):
        if not all(self._native_ok(g) for g in self.param_groups):
            return super().step(closure)
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        for group in self.param_groups:
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            b1, b2 = group["betas"]
            lr = float(group["lr"])
            states = [self.state[p] for p in params]
            for p, st in zip(params, states):
                if len(st) == 0:     # as torch.optim.Adam._init_group: a float32 scalar step on the host, zero moments
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps = [st["step"] for st in states]
            if any(s.is_cuda for s in steps):
                raise RuntimeError("dfnet_amd.optim.Adam: device-resident step counters (a capturable / fused state_dict) are not supported")
            torch._foreach_add_(steps, _ONE, alpha=1.0)
            jobs = np.zeros(len(params), dtype=_ADAM_TENSOR)
            jobs["param"] = [p.data_ptr() for p in params]
            jobs["grad"] = [p.grad.data_ptr() for p in params]
            jobs["exp_avg"] = [st["exp_avg"].data_ptr() for st in states]
            jobs["exp_avg_sq"] = [st["exp_avg_sq"].data_ptr() for st in states]
            jobs["numel"] = [p.numel() for p in params]
            # bias corrections per tensor, in double, as torch: tensors of a group normally share one step count
            corr = {}
            for i, s in enumerate(steps):
                k = float(s)
                if k not in corr:
                    corr[k] = (-(lr / (1 - b1 ** k)), (1 - b2 ** k) ** 0.5)
                jobs["step_size"][i], jobs["bias_correction2_sqrt"][i] = corr[k]
            check(lib.dfn_adam_step(ctypes.c_void_p(jobs.ctypes.data), len(params), float(b1), float(b2), float(group["eps"]),
                                    float(group["weight_decay"]), current_stream()), "dfn_adam_step")
            # The library wrote the parameters behind torch's back: advance their in-place version counters, which is what every
            # in-place torch op does and how the engines notice that the master weights moved (nerfw.HipQuery.refresh, DFNet's
            # device refresh).  No kernel.
            increment_version(params)
        return loss

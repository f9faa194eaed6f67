"""Host-side checks of dfnet_amd.optim.Adam without a GPU: the descriptor layout matches the header, parameters the HIP kernel does not
take (here: CPU tensors) run torch's own step — the class IS a torch.optim.Adam — and its state_dict is torch's."""
import copy
import ctypes
import os
import re

import numpy as np
import torch

from dfnet_amd import _lib, optim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_descriptor_layout_matches_the_header():
    src = open(os.path.join(ROOT, "include", "dfnet_hip.h")).read()
    body = re.search(r"typedef struct dfn_adam_tensor \{(.*?)\} dfn_adam_tensor;", src, flags=re.S).group(1)
    fields = [re.sub(r"\s+", " ", f.strip()).split(" ")[-1].lstrip("*") for f in body.split(";") if f.strip()]
    assert fields == [n for n, _ in _lib.AdamTensor._fields_] == list(optim._ADAM_TENSOR.names)
    assert ctypes.sizeof(_lib.AdamTensor) == optim._ADAM_TENSOR.itemsize == 48
    for (name, _), off in zip(_lib.AdamTensor._fields_, (0, 8, 16, 24, 32, 40, 44)):
        assert getattr(_lib.AdamTensor, name).offset == optim._ADAM_TENSOR.fields[name][1] == off


def test_cpu_parameters_take_torchs_own_step_and_the_state_dict_is_torchs():
    g = torch.Generator().manual_seed(0)
    shapes = [(7, 5), (5,), (3, 2, 2)]
    mk = lambda: [torch.nn.Parameter(torch.randn(*s, generator=torch.Generator().manual_seed(i))) for i, s in enumerate(shapes)]
    a_p, b_p = mk(), mk()
    a, b = optim.Adam(a_p, lr=1e-2, betas=(0.9, 0.999)), torch.optim.Adam(b_p, lr=1e-2, betas=(0.9, 0.999))
    assert isinstance(a, torch.optim.Adam)
    for _ in range(4):
        for p, q in zip(a_p, b_p):
            gr = torch.randn(*p.shape, generator=g)
            p.grad, q.grad = gr.clone(), gr.clone()
        a.step(); b.step()
    assert all(torch.equal(p, q) for p, q in zip(a_p, b_p))                     # the parent class's arithmetic, bit for bit
    sa, sb = a.state_dict(), b.state_dict()
    assert sa["param_groups"] == sb["param_groups"] and sa["state"].keys() == sb["state"].keys()
    for k in sa["state"]:
        assert sa["state"][k].keys() == sb["state"][k].keys() == {"step", "exp_avg", "exp_avg_sq"}
        assert all(torch.equal(sa["state"][k][n], sb["state"][k][n]) for n in sa["state"][k])
    c = torch.optim.Adam(mk(), lr=1e-2)
    c.load_state_dict(copy.deepcopy(sa))                                           # a checkpoint written with ours opens in torch's
    assert float(c.state[c.param_groups[0]["params"][0]]["step"]) == 4
    # ReduceLROnPlateau / the manual decay of run_nerf.py:71-73 act on param_groups as on any torch optimizer
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(a, factor=0.5, patience=0)
    sch.step(1.0); sch.step(2.0)
    assert np.isclose(a.param_groups[0]["lr"], 5e-3)

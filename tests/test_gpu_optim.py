"""dfnet_amd.optim.Adam (one multi-tensor HIP launch per step, csrc/optim_adam.hip) against torch.optim.Adam on the same device:
the optimizer the reference creates in models/nerfw.py:436 (lr 5e-4, betas (0.9, 0.999)), run_feature.py:211 and train.py:60."""
import copy

import numpy as np
import pytest
import torch

from dfnet_amd import optim

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(*s, generator=g) * 0.3).to(DEV)) for s in shapes]


# tensor shapes of the NeRF-H networks and of DFNet's tail: odd sizes, a scalar, a size that is not a multiple of four, > one block
SHAPES = [(128, 63), (128,), (128, 191), (1,), (3, 64), (7,), (1000, 5), (12, 512), (64, 64, 5, 5), (4099,), (2, 3, 1)]


@pytest.mark.parametrize("kw", [dict(lr=5e-4, betas=(0.9, 0.999)), dict(lr=1e-3), dict(lr=2e-3, betas=(0.8, 0.9), eps=1e-6, weight_decay=1e-2)])
def test_adam_matches_torch_step_by_step(kw):
    ref_p, got_p = _params(1, SHAPES), _params(1, SHAPES)
    ref, got = torch.optim.Adam(ref_p, **kw), optim.Adam(got_p, **kw)
    g = torch.Generator().manual_seed(2)
    worst = 0.
    for step in range(1, 13):
        grads = [(torch.randn(*s, generator=g) * (10. ** float(torch.randint(-4, 1, (1,), generator=g)))).to(DEV) for s in SHAPES]
        if step == 7:                                  # the manual decay of run_nerf.py:71-73 acts on param_groups
            for o in (ref, got):
                for grp in o.param_groups:
                    grp["lr"] *= 0.5
        for p, q, gr in zip(ref_p, got_p, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        v0 = [q._version for q in got_p]
        ref.step()
        got.step()
        assert all(q._version > v for q, v in zip(got_p, v0))      # the engines watch these counters
        for p, q in zip(ref_p, got_p):
            for a, b in ((p, q), (ref.state[p]["exp_avg"], got.state[q]["exp_avg"]), (ref.state[p]["exp_avg_sq"], got.state[q]["exp_avg_sq"])):
                err = float((a.detach() - b.detach()).abs().max() / a.detach().abs().max().clamp_min(1e-30))
                worst = max(worst, err)
            assert float(got.state[q]["step"]) == float(ref.state[p]["step"]) == step
    # same arithmetic in the same order, fp32: a few ulp after twelve steps (fused multiply-adds contract differently)
    assert worst < 2e-6, worst
    print(f"Adam vs torch.optim.Adam, 12 steps, {len(SHAPES)} tensors: worst relative difference of params / moments {worst:.1e}")


def test_state_dict_moves_between_the_two_and_unsupported_options_run_torchs_step():
    shapes = [(33, 7), (5,)]
    a_p, b_p = _params(3, shapes), _params(3, shapes)
    a, b = optim.Adam(a_p, lr=1e-3), torch.optim.Adam(b_p, lr=1e-3)
    g = torch.Generator().manual_seed(4)
    for _ in range(3):
        for p, q in zip(a_p, b_p):
            gr = torch.randn(*p.shape, generator=g).to(DEV)
            p.grad, q.grad = gr.clone(), gr.clone()
        a.step(); b.step()
    # torch -> ours -> torch: the checkpoint's optimizer_state_dict (run_nerf.py:136) is interchangeable
    sd = copy.deepcopy(b.state_dict())
    a2 = optim.Adam(_params(3, shapes), lr=1e-3)
    a2.load_state_dict(sd)
    b2 = torch.optim.Adam(_params(3, shapes), lr=1e-3)
    b2.load_state_dict(copy.deepcopy(a.state_dict()))
    for o, ps in ((a2, a2.param_groups[0]["params"]), (b2, b2.param_groups[0]["params"])):
        for p, src in zip(ps, a_p):
            p.data.copy_(src.data)
            p.grad = torch.ones_like(p)
        o.step()
    for p, q in zip(a2.param_groups[0]["params"], b2.param_groups[0]["params"]):
        assert float((p.detach() - q.detach()).abs().max()) < 1e-6 * float(q.detach().abs().max())
        assert float(a2.state[p]["step"]) == float(b2.state[q]["step"]) == 4
    # amsgrad is torch's feature: the subclass runs torch's own step, same numbers as the parent class
    c_p, d_p = _params(5, shapes), _params(5, shapes)
    c, d = optim.Adam(c_p, lr=1e-3, amsgrad=True), torch.optim.Adam(d_p, lr=1e-3, amsgrad=True)
    for p, q in zip(c_p, d_p):
        p.grad = torch.full_like(p, 0.25); q.grad = torch.full_like(q, 0.25)
    c.step(); d.step()
    assert all(torch.equal(p, q) for p, q in zip(c_p, d_p)) and "max_exp_avg_sq" in c.state[c_p[0]]
    # parameters without a gradient are left alone, as torch leaves them
    e_p = _params(6, shapes)
    e = optim.Adam(e_p, lr=1e-2)
    e_p[0].grad = torch.ones_like(e_p[0])
    before = e_p[1].detach().clone()
    e.step()
    assert torch.equal(e_p[1], before) and len(e.state[e_p[1]]) == 0 and float(e.state[e_p[0]]["step"]) == 1


def test_abi_argument_errors():
    import ctypes
    from dfnet_amd import _lib
    lib = _lib.load()
    t = (_lib.AdamTensor * 1)()
    assert lib.dfn_adam_step(ctypes.cast(t, ctypes.c_void_p), 1, 0.9, 0.999, 1e-8, 0., None) == -1 and b"null pointer" in lib.dfn_last_error()
    assert lib.dfn_adam_step(None, 0, 0.9, 0.999, 1e-8, 0., None) == 0

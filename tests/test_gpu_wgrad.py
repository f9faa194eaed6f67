"""dfn_conv_wgrad — the split-storage weight-gradient stream (csrc/dfnet_wgrad_s.hip) — against autograd of
torch.nn.functional.conv2d in fp64: dW[co][ci][ky][kx] and db[co] of one stride-1 "same" convolution, the product behind
every Conv2d parameter gradient of the DFNet training steps (reference: feature/dfnet.py:8-40, :57-62 under loss.backward(),
feature/direct_feature_matching.py:372-374, run_feature.py:166-230).  Shapes cover both strip widths (32 / 48 columns), ragged
strips, odd row counts (the odd row phase idles on the last row), 1x1 / 3x3 / 5x5, several frames per workgroup, and the
DFNet_dm step's own conv4 shape."""
import ctypes

import pytest
import torch

from dfnet_amd import _lib
from dfnet_amd._lib import check, current_stream, ptr

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def conv_wgrad(g, x, ks, want_bias=True):
    lib = _lib.load()
    B, cout, H, W = g.shape
    cin = x.shape[1]
    nb = lib.dfn_conv_wgrad_scratch_bytes(B, H, W, cout, cin, ks)
    assert nb > 0
    scratch = torch.empty(nb, dtype=torch.uint8, device=g.device)
    dW = torch.full((cout, cin, ks, ks), float("nan"), device=g.device)
    db = torch.full((cout,), float("nan"), device=g.device) if want_bias else None
    check(lib.dfn_conv_wgrad(ptr(g), ptr(x), B, H, W, cout, cin, ks, ptr(dW), ptr(db), ctypes.c_void_p(scratch.data_ptr()), scratch.numel(),
                             current_stream()), "dfn_conv_wgrad")
    return dW, db


def reference(g, x, ks):
    """autograd of F.conv2d in fp64 on the CPU"""
    g64, x64 = g.double().cpu(), x.double().cpu()
    w = torch.zeros(g.shape[1], x.shape[1], ks, ks, dtype=torch.float64, requires_grad=True)
    bias = torch.zeros(g.shape[1], dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x64, w, bias, padding=ks // 2)
    y.backward(g64)
    return w.grad, bias.grad


@pytest.mark.parametrize("B,H,W,cout,cin,ks", [
    (2, 30, 40, 128, 64, 3),     # conv4-like row width: one 48-column strip
    (2, 15, 20, 64, 64, 3),      # conv5-like: one ragged 32-column strip, odd row count
    (1, 12, 80, 64, 128, 3),     # three 32-column strips, the last one half empty
    (3, 9, 33, 64, 64, 3),       # nothing divides anything
    (2, 16, 100, 128, 64, 5),    # the adaptation layers' 5x5: one kernel row per launch
    (2, 14, 40, 64, 128, 1),     # ... and their 1x1
    (20, 8, 16, 256, 256, 3),    # more (pairs x frames) than CUs: several frames per workgroup
    (4, 30, 40, 512, 256, 3),    # conv4_1 of the DFNet_dm step (batch 4, 240x320 frames)
])
def test_conv_wgrad_vs_autograd(B, H, W, cout, cin, ks):
    gen = torch.Generator().manual_seed(B * 1000 + H * 10 + W + cout + cin + ks)
    # gradients with a wide dynamic range (most elements far below the maximum), ReLU'd activations with exact zeros
    g = (torch.randn(B, cout, H, W, generator=gen) * torch.exp(2.0 * torch.randn(B, cout, 1, 1, generator=gen))).to(DEV)
    x = torch.relu(torch.randn(B, cin, H, W, generator=gen) + 0.3).to(DEV)
    dW, db = conv_wgrad(g, x, ks)
    ref_w, ref_b = reference(g, x, ks)
    ew = float((dW.double().cpu() - ref_w).abs().max() / ref_w.abs().max())
    el = float((dW.double().cpu() - ref_w).norm() / ref_w.norm())
    eb = float((db.double().cpu() - ref_b).abs().max() / ref_b.abs().max())
    print(f"conv wgrad {ks}x{ks} B{B} {H}x{W} {cin}->{cout}: dW max-rel {ew:.2e}, rel L2 {el:.2e}, db max-rel {eb:.2e}")
    assert ew < 5e-6 and el < 5e-6 and eb < 5e-6
    dW2, db2 = conv_wgrad(g, x, ks)
    assert torch.equal(dW, dW2) and torch.equal(db, db2)   # fixed summation order: bit-identical reruns
    dW3, none = conv_wgrad(g, x, ks, want_bias=False)
    assert none is None and torch.equal(dW, dW3)


def test_conv_wgrad_zero_gradient_and_bad_shapes():
    lib = _lib.load()
    g = torch.zeros(1, 64, 8, 8, device=DEV)
    x = torch.rand(1, 64, 8, 8, device=DEV)
    dW, db = conv_wgrad(g, x, 3)
    assert float(dW.abs().max()) == 0.0 and float(db.abs().max()) == 0.0
    assert lib.dfn_conv_wgrad_scratch_bytes(1, 8, 8, 32, 64, 3) == 0     # channel counts: multiples of 64
    assert lib.dfn_conv_wgrad_scratch_bytes(1, 8, 8, 64, 64, 7) == 0
    rc = lib.dfn_conv_wgrad(ptr(g), ptr(x), 1, 8, 8, 96, 64, 3, ptr(dW), None, ctypes.c_void_p(1), 1, current_stream())
    assert rc != 0 and b"multiples of 64" in lib.dfn_last_error()

#!/usr/bin/env python3
"""Where the cycles of conv_x3_kernel go (DFN_TIMING build, `make -C dfnet_amd/csrc timing`): one split-f16 DFNet forward
per layer type, lane-0 cycle counters summed over all waves."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DFN_LIB_PATH"] = os.environ.get("TIMING_LIB") or os.path.join(ROOT, "dfnet_amd", "libdfnet_hip_timing.so")
sys.path.insert(0, ROOT)
from dfnet_amd import _lib, engine as eng, synthetic as syn
lib = _lib.load()
lib.dfn_debug_conv_cycles.argtypes = [ctypes.c_void_p, ctypes.c_int]
E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
x = torch.rand(4, 3, 480, 640, device="cuda:0")
names = ["dma/patch wait", "barrier", "patch split+store", "frag reads + MFMA issue", "epilogue"]
for what, kw in (("encoder only (3x3 convs)", dict(return_feature=False, return_pose=True)),
                 ("features (3x3 + 1x1 + 5x5)", dict(return_feature=True, return_pose=False))):
    E.forward(x, kw["return_feature"], True, kw["return_pose"], 480, 640, precision="f16x3")
    torch.cuda.synchronize()
    lib.dfn_debug_conv_cycles(None, 1)
    E.forward(x, kw["return_feature"], True, kw["return_pose"], 480, 640, precision="f16x3")
    torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 8)()
    lib.dfn_debug_conv_cycles(out, 1)
    tot = float(out[5])
    print(f"{what}: {out[6]} waves, mean {tot / max(out[6], 1):.0f} cycles (100 MHz s_memtime ticks) per wave")
    for i, n in enumerate(names):
        print(f"  {n:26s} {100 * out[i] / tot:5.1f} %")

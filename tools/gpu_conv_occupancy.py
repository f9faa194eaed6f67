#!/usr/bin/env python3
"""Resident workgroups per CU of the conv kernels as the HIP runtime computes them (DFN_TIMING build: make -C dfnet_amd/csrc conv_timing)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import torch
torch.zeros(1, device="cuda:0")
lib = ctypes.CDLL(os.path.join(ROOT, "dfnet_amd", "libdfnet_hip_timing.so"))
out = (ctypes.c_int * 8)()
n = lib.dfn_debug_conv_occupancy(out, 8)
print(dict(zip(["x3<3,16,MB2> 8x32", "x3s<3,16,MB2> 8x32", "x3<3,16,MB1> 16x16", "x3s<1,16>", "f16<3,16,MB4>"], list(out)[:n])))

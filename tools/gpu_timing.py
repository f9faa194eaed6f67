#!/usr/bin/env python3
"""Per-wave cycle accounting of the fine MLP kernel (DFN_TIMING build: libdfnet_hip_timing.so)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["DFN_LIB_PATH"] = os.environ.get("TIMING_LIB") or os.path.join(ROOT, "dfnet_amd", "libdfnet_hip_timing.so")
sys.path.insert(0, ROOT)
from dfnet_amd import _lib, engine as eng, synthetic as syn
lib = _lib.load()
dev = "cuda:0"
cw, fw, ea, et = syn.nerfh_weights(0)
E = eng.NerfHEngine(precision=os.environ.get("PREC", "f16")).load_numpy(cw, fw, ea, et)
NW = 8192
buf = torch.zeros(NW * 4 + 8 * 192 + 64, dtype=torch.int64, device=dev)
c2w = torch.from_numpy(syn.orbit_pose(0, 8)).to(dev)
hist = torch.from_numpy(syn.HIST_IDX).to(dev)
E.render_image(c2w, 480, 640, 585.0, hist, 64, 128, 0., 2.5)
lib.dfn_debug_set_timing_buffer.argtypes = [ctypes.c_void_p]
lib.dfn_debug_set_timing_buffer(ctypes.c_void_p(buf.data_ptr()))
E.render_image(c2w, 480, 640, 585.0, hist, 64, 128, 0., 2.5)
torch.cuda.synchronize()
raw = buf.cpu().numpy()
t = raw[:NW * 4].reshape(-1, 4).astype(np.float64)
t = t[t[:, 0] > 0]
tot = t[:, 0]
print("waves", len(t), "total cycles/wave mean %.3g" % tot.mean(), "(last launch of the frame)")
for i, n in ((1, "dma wait (vmcnt0)"), (2, "barrier"), (3, "tile input wait")):
    print("%-20s %.1f%% (min %.1f%% max %.1f%%)" % (n, 100 * (t[:, i] / tot).mean(), 100 * (t[:, i] / tot).min(), 100 * (t[:, i] / tot).max()))
w = t.reshape(-1, 8, 4)
print("per wave-slot mean barrier%:", np.round(100 * (w[:, :, 2] / w[:, :, 0]).mean(0), 1))
print("per wave-slot mean dma-wait%:", np.round(100 * (w[:, :, 1] / w[:, :, 0]).mean(0), 1))

tr = raw[NW * 4: NW * 4 + 8 * 192].reshape(8, 96, 2).astype(np.int64)
base = tr[:, 0, 0].min()
print("timeline of workgroup 7: per unit u, waves 0 and 4: enter barrier, released, busy until next enter")
for u in range(0, 34):
    e0, r0 = tr[0, u] - base; e4, r4 = tr[4, u] - base
    print(f"u{u:2d}  w0 enter {e0:7d} rel {r0:7d} busy {tr[0, u + 1, 0] - base - r0:6d} | w4 enter {e4:7d} rel {r4:7d} busy {tr[4, u + 1, 0] - base - r4:6d}")

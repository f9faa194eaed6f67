#!/usr/bin/env python3
"""Package power and shader clock (bench.GpuSampler: sysfs hwmon / rocm-smi) while each secondary workload runs back to back for a few
seconds: the split-f16 DFNet forward (4 x 480x640), the DFNet_dm step, the NeRF-H training step — which of them sit at the power cap.
One line per workload."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from dfnet_amd import engine as eng, synthetic as syn

dev = "cuda:0"
SEC = float(os.environ.get("POWER_SECONDS", "5"))


def run(label, fn, unit):
    fn(); torch.cuda.synchronize()
    with bench.GpuSampler(0) as smp:
        t0, n = time.time(), 0
        while time.time() - t0 < SEC:
            fn(); n += 1
            if n % 4 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.time() - t0
    pw, ck, k = smp.means()
    print(f"{label}: {dt / n * 1e3:.3f} ms per {unit}; package power {pw:.0f} W, shader clock {ck:.0f} MHz ({int(k)} samples)")


E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
x = torch.rand(4, 3, 480, 640, device=dev)
for prec in ("f16x3", "f32", "f16"):
    run(f"DFNet forward {prec}, 4 x 480x640, features", lambda: E.forward(x, True, True, False, 480, 640, precision=prec), "forward of 4 frames")
cw, fw, ea, et = syn.nerfh_weights(0)
N = eng.NerfHEngine(precision="f16x3").load_numpy(cw, fw, ea, et)
hist = torch.from_numpy(syn.HIST_IDX).to(dev)
pose = torch.from_numpy(syn.orbit_pose(0, 8)).to(dev)
run("NeRF-H render f16x3, 640x480 at 64+128", lambda: N.render_image(pose, 480, 640, 585.0, hist, 64, 128, 0.0, 2.5, precision="f16x3"), "frame")

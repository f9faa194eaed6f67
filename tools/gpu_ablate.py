#!/usr/bin/env python3
"""Ablation timing of the fine MLP kernel: each libabl_*.so is the library with one cost removed (timing only)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import time, torch
    sys.path.insert(0, ROOT)
    from dfnet_amd import engine as eng, synthetic as syn
    dev = "cuda:0"
    cw, fw, ea, et = syn.nerfh_weights(0)
    E = eng.NerfHEngine(precision=os.environ.get("PREC", "f16")).load_numpy(cw, fw, ea, et)
    n = 61440
    o, d, v = eng.raygen(480, 640, 585.0, torch.from_numpy(syn.orbit_pose(0, 8)).to(dev))
    o, d, v = o.reshape(-1, 3)[:n].contiguous(), d.reshape(-1, 3)[:n].contiguous(), v.reshape(-1, 3)[:n].contiguous()
    z = torch.sort(torch.cat([torch.linspace(0, 2.5, 64).expand(n, 64), torch.rand(n, 128) * 2.5], -1), -1)[0].to(dev)
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    E.mlp_fine(o, d, v, hist, z); E.mlp_coarse(o, d, 64, 0., 2.5)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): E.mlp_fine(o, d, v, hist, z)
    e1.record(); torch.cuda.synchronize()
    print("fine %.3f ms" % (e0.elapsed_time(e1) / 3), end="  ")
    e0.record()
    for _ in range(3): E.mlp_coarse(o, d, 64, 0., 2.5)
    e1.record(); torch.cuda.synchronize()
    print("coarse %.3f ms" % (e0.elapsed_time(e1) / 3))
else:
    for var in ("0",):
        for lib in [os.path.join(ROOT, "dfnet_amd", "libdfnet_hip.so")] + sorted(glob.glob(os.path.join(ROOT, "dfnet_amd", "libabl_*.so"))):
            env = dict(os.environ, DFN_LIB_PATH=lib, DFN_MLP_VARIANT=var)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print("variant", var, os.path.basename(lib)[3:-3].ljust(28), r.stdout.strip() or r.stderr[-300:])

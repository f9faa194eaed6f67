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
    E = eng.NerfHEngine().load_numpy(cw, fw, ea, et)
    n = 61440
    g = torch.Generator().manual_seed(0)
    o = (torch.rand(n, 3, generator=g) - .5).to(dev); d = torch.randn(n, 3, generator=g).to(dev)
    v = d / d.norm(dim=-1, keepdim=True)
    z = torch.sort(torch.rand(n, 192, generator=g) * 2.5, -1)[0].to(dev)
    hist = torch.from_numpy(syn.HIST_IDX).to(dev)
    for var in (0, 2):
        os.environ["DFN_MLP_VARIANT"] = str(var)
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
    for var in ("0", "2"):
        for lib in sorted(glob.glob(os.path.join(ROOT, "dfnet_amd", "libabl_*.so"))):
            env = dict(os.environ, DFN_LIB_PATH=lib, DFN_MLP_VARIANT=var)
            r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
            print("variant", var, os.path.basename(lib)[7:-3].ljust(28), r.stdout.strip() or r.stderr[-300:])

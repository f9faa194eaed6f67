#!/usr/bin/env python3
"""Per-dispatch durations of the product kernels of ONE NeRF-H optimisation step, from a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --output-format csv -d DIR -o tr -- python tools/gpu_nerf_train_step.py 1536 128 6
   python tools/gpu_train_layers.py DIR"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "stratified_z" in r["Kernel_Name"]]
step = rows[idx[-2]:idx[-1]]
out, tot = [], {}
for r in step:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].replace("dfn::train::", "").replace("void ", "").split("(")[0]
    tot[n] = tot.get(n, 0.) + d
    if "gemm" in n:
        out.append("%s:%.0f" % (n.replace("gemm_", "").replace("_kernel", ""), d))
print(" ".join(out))
print({k: round(v) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:8]}, "sum %.0f us" % sum(tot.values()))

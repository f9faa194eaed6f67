#!/usr/bin/env python3
"""Convolution launches of ONE steady step out of a rocprofv3 kernel trace (DFNet_dm step / DFNet training step): workgroups, duration,
queue — the table that shows which launches leave the chip idle.   python tools/gpu_step_convs.py TRACE_DIR [marker-kernel-substring]"""
import csv, glob, os, sys
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
marker = sys.argv[2] if len(sys.argv) > 2 else "nerfh_coarse_kernel"
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
tot = {}
for r in rows[a:b]:
    n = r["Kernel_Name"].replace("void ", "").replace("dfn::", "").split("(")[0]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot[n] = tot.get(n, 0.) + d
    if "conv" in n and "finalize" not in n:
        g = [int(r[k]) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")]
        w = [int(r[k]) for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z")]
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:7.1f} us  wgs {(g[0] // w[0]) * (g[1] // w[1]) * (g[2] // w[2]):5d} q{r['Queue_Id']} {n[:64]}")
print("step span %.1f us; kernel time by name:" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {v:8.1f} us  {k[:80]}")

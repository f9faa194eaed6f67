#!/usr/bin/env python3
"""Per-kernel durations of ONE split-f16 DFNet forward (4 x 480x640, features only), from a rocprofv3 kernel trace.
   run:    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/layers -o l -- python tools/gpu_dfnet_layers.py run
   report: python tools/gpu_dfnet_layers.py report gpurun_out/layers"""
import csv, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, ROOT)
    from dfnet_amd import engine as eng, synthetic as syn
    E = eng.DfnetEngine(3, 12).load_numpy(syn.dfnet_weights(3))
    B = int(os.environ.get("LAYERS_B", "4"))
    LH, LW = int(os.environ.get("LAYERS_H", "480")), int(os.environ.get("LAYERS_W", "640"))
    x = torch.rand(B, 3, LH, LW, device="cuda:0")
    for _ in range(3):
        E.forward(x, True, True, False, LH, LW, precision=os.environ.get("LAYERS_PREC", "f16x3"))
    torch.cuda.synchronize()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = [r for r in csv.DictReader(open(f)) if "dfn::" in r["Kernel_Name"] or "_ZN3dfn" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    n = len(rows) // 3
    last = rows[-n:]
    tot = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
    span = max(int(r["End_Timestamp"]) for r in last) - int(last[0]["Start_Timestamp"])   # (the last kernel to START is not the last to end)
    print(f"{n} kernels, sum {tot / 1e3:.1f} us, span {span / 1e3:.1f} us")
    if tot > 1.05 * span:
        print("(the adaptation branches of pyramid levels 1-2 — 1x1, 5x5, upsample_rows — run on a side stream beside the encoder and the level-0 5x5:\n"
              " the durations of kernels that overlap are each inflated by the other; `span` is the forward's wall time, kernels listed by start time)")
    for r in last:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        g = "x".join(r.get(k, "?") for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        t0 = int(last[0]["Start_Timestamp"])
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} + {d / 1e3:7.1f} us  {100 * d / tot:5.1f} %  q{r.get('Queue_Id', '?')}  grid {g:>16s}  {r['Kernel_Name'][:86]}")

#!/usr/bin/env python3
"""Where do the device-to-device copies of a DFNet_dm step come from?  torch.profiler over two steps: memcpy events and the
CPU ops that launched them."""
import collections, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["DM_ONLY"] = "1"
import runpy
from torch.profiler import profile, ProfilerActivity
sys.argv = ["gpu_dm_step.py", "4", "1"]
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=False) as prof:
    try:
        runpy.run_path(os.path.join(ROOT, "tools", "gpu_dm_step.py"), run_name="__main__")
    except SystemExit:
        pass
ev = prof.events()
mem = [e for e in ev if "emcpy" in e.name or "copyBuffer" in e.name]
print("memcpy-like events:", len(mem))
c = collections.Counter()
for e in mem:
    p = e.cpu_parent
    names = []
    while p is not None and len(names) < 4:
        names.append(p.name); p = p.cpu_parent
    st = [s for s in (e.stack or []) if "dfnet_amd" in s or "tools/" in s][:2]
    c[(e.name[:30], " < ".join(names), " | ".join(st))] += 1
for k, v in c.most_common(30):
    print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))

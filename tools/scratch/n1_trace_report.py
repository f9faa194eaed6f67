import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# steps are delimited by the coarse forward chain kernel
idx = [i for i, r in enumerate(rows) if "train_fwd_chain_kernel<false" in r["Kernel_Name"] or "train_fwd_chain_kernelILb0" in r["Kernel_Name"]]
print("steps found", len(idx))
a, b = idx[-3], idx[-2]
# walk back from the coarse chain to the step's first kernel (viewdirs)
while a > 0 and "viewdirs" not in rows[a]["Kernel_Name"]: a -= 1
while b > 0 and "viewdirs" not in rows[b]["Kernel_Name"]: b -= 1
t0 = int(rows[a]["Start_Timestamp"])
print("step span %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3))
prev_end = t0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur = (e - s) / 1e3
    gap = (s - prev_end) / 1e3
    if dur >= 8 or gap > 5:
        print(f"{(s - t0) / 1e3:8.1f} + {dur:7.1f} us  gap {gap:6.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:80]}")
    prev_end = max(prev_end, e)

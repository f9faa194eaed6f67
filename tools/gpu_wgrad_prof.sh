#!/bin/bash
# kernel-trace durations of the fused step's kernels per tuning mode (DFN_WGRAD_MODE)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for m in 0; do
  rm -rf $R/gpurun_out/prof_w$m
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_w$m -o w -- python $R/tools/gpu_nerf_train_step.py 1536 128 10 > /dev/null 2>&1
  echo "== mode $m"
  python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/gpurun_out/prof_w$m/w_kernel_stats.csv")))
for r in rows[:12]:
    print(f"{r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:8.1f} us min {float(r['MinNs'])/1e3:8.1f} max {float(r['MaxNs'])/1e3:8.1f}")
PY
done

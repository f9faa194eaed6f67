#!/bin/bash
# Kernel table of one training step under rocprofv3 (per-step totals, every kernel above a threshold): tools/gpu_step_kernels.sh n2|dm|n1 [min_us_per_step]
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; export TMPDIR=/tmp
case "$1" in n2) CMD="python $R/tools/gpu_feature_train_step.py 4 20 240 320"; STEPS=21;; dm) CMD="python $R/tools/gpu_dm_step.py 4 24"; STEPS=25; export DM_ONLY=1;; n1) CMD="python $R/tools/gpu_nerf_train_step.py 1536 128 10"; STEPS=22;; *) echo "n2|dm|n1"; exit 2;; esac
rm -rf /tmp/prof_sk; ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_sk -o k -- $CMD > /tmp/prof_sk.json 2> /tmp/prof_sk.err ); tail -1 /tmp/prof_sk.json | cut -c1-400
python3 - $STEPS ${2:-20} <<'PY'
import csv, glob, sys
steps, thr = int(sys.argv[1]), float(sys.argv[2])
rows = list(csv.DictReader(open(glob.glob('/tmp/prof_sk/**/k_kernel_stats.csv', recursive=True)[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
print("kernel ms per step: %.3f, launches per step: %.1f" % (sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps))
for r in rows:
    us = float(r['TotalDurationNs']) / steps / 1e3
    if us >= thr: print(f"{int(r['Calls'])/steps:6.1f} x {float(r['AverageNs'])/1e3:8.1f} us = {us:8.1f} us/step  {r['Name'][:90]}")
PY

#!/bin/bash
# Within-run A/B of two builds of the library: tools/gpu_ab_lib.sh labelA:pathA labelB:pathB ... (paths relative to the repo)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo -n "$1: "; DFN_LIB_PATH=$R/$2 timeout 300 python bench.py --cpu-sample 0 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3fM rays/s  %.2f ms/frame  fine frac %.3f  fine %.3f ms coarse %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['coarse_kernel_avg_launch_ms']))"; }
for spec in "$@"; do run "${spec%%:*}" "${spec#*:}"; done

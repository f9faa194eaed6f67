#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { echo -n "variant=$1: "; DFN_MLP_VARIANT=$1 timeout 300 python bench.py --cpu-sample 0 --steps 6 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3fM rays/s  %.2f ms/frame  fine frac %.3f  fine %.3f ms coarse %.3f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['coarse_kernel_avg_launch_ms']))"; }
for v in 0 1 2 3; do DFN_MLP_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_nerfh.py -x -q -k "mlp or golden or full" 2>&1 | tail -1; done
run 0; run 3; run 1; run 2

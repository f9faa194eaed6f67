#!/bin/bash
# A/B of the weight-gradient stream's tuning knobs on one box: DFN_WGRAD_MODE (0 full, 1 stream only, 2 products only), DFN_WGRAD_NT, DFN_WGRAD_D
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "0 1 0" "0 0 0" "1 1 0" "2 1 0"; do
  set -- $cfg
  DFN_WGRAD_MODE=$1 DFN_WGRAD_NT=$2 DFN_WGRAD_D=$3 python tools/gpu_nerf_train_step.py 1536 128 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mode $1 nt $2 depth $3', {k: round(v,3) for k,v in d.items() if k.endswith('_ms') or k.endswith('adam')})"
done

#!/bin/bash
# Same-box A/B: the adaptation layers' weight gradients on the side stream (default) or on the chain's stream (DFN_ADAPT_WGRAD_SIDE=0)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DFN_ADAPT_WGRAD_SIDE=$v N2 loop: "; DFN_ADAPT_WGRAD_SIDE=$v FT_LOOP=1 timeout 300 python tools/gpu_feature_train_step.py 4 20 240 320 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['step_ms'])"
  echo -n "DFN_ADAPT_WGRAD_SIDE=$v DM all levels: "; DFN_ADAPT_WGRAD_SIDE=$v DM_ONLY=1 DM_ALL_LEVELS=1 timeout 300 python tools/gpu_dm_step.py 4 24 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['full_step_ms'])"
  echo -n "DFN_ADAPT_WGRAD_SIDE=$v DM: "; DFN_ADAPT_WGRAD_SIDE=$v DM_ONLY=1 timeout 300 python tools/gpu_dm_step.py 4 24 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['full_step_ms'])"
done; done

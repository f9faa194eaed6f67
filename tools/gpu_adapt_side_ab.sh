#!/bin/bash
# Same-box A/B of an environment switch on the DFNet training steps: tools/gpu_adapt_side_ab.sh VAR  (VAR=0 against VAR=1, three alternations;
# DFN_ADAPT_WGRAD_SIDE: the adaptation layers' weight gradients on the side stream; DFN_ADAPT_FWD_SIDE: their forward branches beside the encoder)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; VAR=${1:-DFN_ADAPT_WGRAD_SIDE}
for rep in 1 2 3; do for v in 0 1; do
  echo -n "$VAR=$v N2 loop: "; env $VAR=$v FT_LOOP=1 timeout 300 python tools/gpu_feature_train_step.py 4 20 240 320 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['step_ms'])"
  echo -n "$VAR=$v DM all levels: "; env $VAR=$v DM_ONLY=1 DM_ALL_LEVELS=1 timeout 300 python tools/gpu_dm_step.py 4 24 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['full_step_ms'])"
  echo -n "$VAR=$v DM: "; env $VAR=$v DM_ONLY=1 timeout 300 python tools/gpu_dm_step.py 4 24 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['full_step_ms'])"
done; done

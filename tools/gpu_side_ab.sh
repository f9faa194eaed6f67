#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 900 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_grad.py -x -q -k "wgrad or parameter_gradients or kept_forward or c5_size" 2>&1 | tail -2
for rep in 1 2 3; do
  echo -n "side   ft: "; python tools/gpu_feature_train_step.py 4 20 240 320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['step_ms'],3), {k: round(v,2) for k,v in d['breakdown_ms_with_syncs'].items()})"
  echo -n "noside ft: "; DFN_NO_SIDE=1 python tools/gpu_feature_train_step.py 4 20 240 320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['step_ms'],3), {k: round(v,2) for k,v in d['breakdown_ms_with_syncs'].items()})"
  echo -n "side   dm: "; python tools/gpu_dm_step.py 4 24 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['full_step_with_adam_and_device_repack_ms'],3), round(d['forward_backward_all_gradients_ms'],3))"
  echo -n "noside dm: "; DFN_NO_SIDE=1 python tools/gpu_dm_step.py 4 24 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['full_step_with_adam_and_device_repack_ms'],3), round(d['forward_backward_all_gradients_ms'],3))"
done

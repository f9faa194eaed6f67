#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_dfnet.py -q -x 2>&1 | tail -4
for i in 1 2 3; do
  for v in "" OLD2; do
    lib=""; [ -n "$v" ] && lib=$R/dfnet_amd/libvar_$v.so
    echo "=== '${v:-new}'"
    DFN_LIB_PATH=$lib FT_LOOP=1 timeout 600 python tools/gpu_feature_train_step.py 4 20 240 320 2>&1 | tail -1 | cut -c170-260
    DFN_LIB_PATH=$lib DM_ONLY=1 timeout 600 python tools/gpu_dm_step.py 4 24 2>&1 | tail -1
  done
done

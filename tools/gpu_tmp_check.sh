#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_gpu_train.py -q -x 2>&1 | tail -4
for i in 1 2 3; do
  for v in "" Y0; do
    lib=""; [ -n "$v" ] && lib=$R/dfnet_amd/libvar_$v.so
    echo "=== '${v:-new}'"; DFN_LIB_PATH=$lib timeout 600 python tools/gpu_nerf_train_step.py 2>&1 | tail -1 | cut -c90-260
  done
done

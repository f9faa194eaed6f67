#!/bin/bash
# DFNet training-path tests + step timings after the split-storage chain
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_grad.py tests/test_gpu_dfnet.py tests/test_gpu_cli.py -q > gpurun_out/pytest_chain.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_chain.log | tail -30
python $R/tools/gpu_feature_train_step.py 4 20 240 320 | cut -c1-700
python $R/tools/gpu_dm_step.py 4 24 | cut -c1-900

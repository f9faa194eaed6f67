#!/bin/bash
# Same-box A/B of the N1 backward's chain launch: DFN_TRAIN_BWD_PAIR=0 (the fine and the coarse data-gradient chains one after the
# other) against 1 (both as the halves of one grid, nerfh_fused_chain.hip: train_bwd_chain_pair_kernel), three alternations of the
# step loop with device-side phase intervals (HIP events), then the parity tests of the step and the schedule digests.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
{
for rep in 1 2 3; do for v in 0 1; do
  echo -n "DFN_TRAIN_BWD_PAIR="; env DFN_TRAIN_BWD_PAIR=$v timeout 300 python tools/gpu_n1_phases.py 2>/dev/null | tail -1
done; done
} 2>&1 | tee gpurun_out/r06_n1_bwd_pair_ab.log
timeout 1500 python -m pytest tests/test_gpu_streams.py tests/test_gpu_train.py tests/test_gpu_grad.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06_n1_bwd_pair_tests.log

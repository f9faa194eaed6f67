#!/bin/bash
# Round-5 "before" pass on one box: GPU parity tests, then rocprof kernel stats of the N2 step (DFNet training) and the DFNet_dm step.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 gpurun_out/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ft0 $R/gpurun_out/prof_dm0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ft0 -o ft -- python $R/tools/gpu_feature_train_step.py 4 20 240 320 > $R/gpurun_out/ft_step0.json 2> $R/gpurun_out/ft_step0.err; echo "ft rc=$?"
cat $R/gpurun_out/ft_step0.json | cut -c1-600
DM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dm0 -o dm -- python $R/tools/gpu_dm_step.py 4 24 > $R/gpurun_out/dm_step0.json 2> $R/gpurun_out/dm_step0.err; echo "dm rc=$?"
cat $R/gpurun_out/dm_step0.json | cut -c1-600
python $R/tools/gpu_feature_train_step.py 4 20 240 320 | cut -c1-400
python $R/tools/gpu_dm_step.py 4 24 | cut -c1-600
find $R/gpurun_out/prof_ft0 $R/gpurun_out/prof_dm0 -name "*kernel_stats.csv" | head

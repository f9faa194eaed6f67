#!/bin/bash
# rocprof kernel stats of the N2 step (DFNet training) and the DFNet_dm step; TAG = output suffix
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ft_$TAG $R/gpurun_out/prof_dm_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ft_$TAG -o ft -- python $R/tools/gpu_feature_train_step.py 4 20 240 320 > $R/gpurun_out/ft_step_$TAG.json 2> $R/gpurun_out/ft_step_$TAG.err; echo "ft rc=$?"
DM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dm_$TAG -o dm -- python $R/tools/gpu_dm_step.py 4 24 > $R/gpurun_out/dm_step_$TAG.json 2> $R/gpurun_out/dm_step_$TAG.err; echo "dm rc=$?"
python $R/tools/gpu_feature_train_step.py 4 20 240 320 | cut -c1-420
python $R/tools/gpu_dm_step.py 4 24 | cut -c1-420

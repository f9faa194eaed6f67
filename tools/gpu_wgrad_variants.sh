#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; timeout 600 python -m pytest tests/test_gpu_wgrad.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for V in 0 1 2 0 1; do
  rm -rf $R/gpurun_out/prof_wgl
  DFN_WGS_V=$V timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_wgl -o w -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
  echo "== variant $V"; python $R/tools/gpu_wgrad_layers.py report $R/gpurun_out/prof_wgl | tee $R/gpurun_out/wgrad_layers_v$V.txt | grep -v "^layer"
done

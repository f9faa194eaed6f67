#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  rm -rf $R/gpurun_out/prof_wgl
  env "${@:2}" timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_wgl -o w -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
  echo "== $1"; python $R/tools/gpu_wgrad_layers.py report $R/gpurun_out/prof_wgl | grep -E "conv1_2|conv2_2|conv3_2|conv4_2|conv5_1|total"
}
run base DFN_WGS_V=0
run interleaved DFN_WGS_V=3
run nomfma DFN_LIB_PATH=$R/dfnet_amd/libvar_NOMFMA.so
run nobarrier DFN_LIB_PATH=$R/dfnet_amd/libvar_NOBARRIER.so
run nodma DFN_LIB_PATH=$R/dfnet_amd/libvar_NODMA.so
run base DFN_WGS_V=0
run interleaved DFN_WGS_V=3
cd $R; DFN_WGS_V=3 timeout 600 python -m pytest tests/test_gpu_wgrad.py -x -q 2>&1 | tail -2

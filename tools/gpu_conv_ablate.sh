#!/bin/bash
# Time the split-f16 3x3 conv under compile-time ablations (scratch libs dfnet_amd/libdfnet_abl_*.so).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in ${ABL_VARIANTS:-BASE NOSTORE NOMFMA NODMA NOBREAD NOAREAD}; do
  if [ $v = BASE ]; then export DFN_LIB_PATH=$R/dfnet_amd/libdfnet_hip.so; else export DFN_LIB_PATH=$R/dfnet_amd/libdfnet_abl_$v.so; fi
  rm -rf $R/gpurun_out/abl_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$v -o a -- python $R/tools/gpu_dfnet_bench.py 4 > /dev/null 2>&1
  echo "== $v"; grep -E "conv_x3_kernel<(3|5), 16" $R/gpurun_out/abl_$v/a_kernel_stats.csv | cut -d, -f1-4 | cut -c1-110
done

#!/bin/bash
# A/B of library builds in one GPU-box call, three alternations: tools/gpu_ab_libs.sh "command" lib1.so lib2.so ... (libs under dfnet_amd/;
# the command's last line is printed per library).  Variant builds: tools/build_variant.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
CMD=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
  echo -n "$lib: "; DFN_LIB_PATH=$R/dfnet_amd/$lib timeout 300 $CMD 2>&1 | grep -v amdgpu.ids | tail -${LINES_KEPT:-1}; done; done

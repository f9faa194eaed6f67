#!/bin/bash
# A/B of library builds in one GPU-box call (fine / coarse MLP kernel ms per 61,440-ray pass): tools/gpu_ab_libs.sh PREC lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
PREC=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
  echo -n "$lib $PREC: "; DFN_LIB_PATH=$R/dfnet_amd/$lib PREC=$PREC timeout 120 python tools/gpu_ablate.py child 2>&1 | tail -1; done; done

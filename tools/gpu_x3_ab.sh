#!/bin/bash
# A/B of the split-f16 / f16 MLP kernel variants in one GPU-box call: fine / coarse kernel ms per 61,440-ray pass (tools/gpu_ablate.py child).
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do for v in ${VARIANTS:-0 3}; do for prec in ${PRECS:-f16x3 f16}; do
  echo -n "variant $v $prec: "; DFN_MLP_VARIANT=$v PREC=$prec timeout 120 python tools/gpu_ablate.py child 2>&1 | tail -1; done; done; done

#!/bin/bash
# A/B of MLP kernel knobs in one GPU-box call: fine / coarse kernel ms per 61,440-ray pass.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
for dw in 8 4; do
  echo -n "x3  dma_waves $dw: "; PREC=f16x3 DFN_DMA_WAVES=$dw python tools/gpu_ablate.py child
  echo -n "f16 dma_waves $dw: "; PREC=f16 DFN_DMA_WAVES=$dw python tools/gpu_ablate.py child
done; done

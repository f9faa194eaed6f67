#!/bin/bash
# One GPU-box pass for the split-f16 MLP kernel: per-unit timeline (timing build) and ablation timings.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
PREC=${PREC:-f16x3}
echo "== ablations ($PREC)"; PREC=$PREC timeout 900 python tools/gpu_ablate.py 2>&1 | tee gpurun_out/x3_ablate.txt
echo "== timeline ($PREC)"; PREC=$PREC timeout 300 python tools/gpu_timing.py 2>&1 | tee gpurun_out/x3_timing.txt | head -60
echo "== f16 reference"; PREC=f16 DFN_LIB_PATH=$R/dfnet_amd/libdfnet_hip.so timeout 300 python tools/gpu_ablate.py child

#!/bin/bash
# N1 fused step: stored operands of the fine / coarse network as ONE f16 plane or as hi | lo planes, same box —
#   shipped = fine 1 plane, coarse 2;  libvar_P22 = both 2 (round 4);  libvar_P11 = both 1
# per-tensor gradient distance from the exact-fp32 step (random weights and trained-like weights), then step time.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
for v in "" P22 P11; do
  lib=""; [ -n "$v" ] && lib=$R/dfnet_amd/libvar_$v.so
  [ -n "$v" ] && [ ! -f "$lib" ] && continue
  for args in "256 64 128" "1536 64 128" "1536 64 128 trained"; do
    echo "=== variant '${v:-shipped}' $args"
    DFN_LIB_PATH=$lib timeout 600 python tools/gpu_fused_debug.py $args 2>&1 | grep -v "^grad .*e-0[6-9]" | tail -40
  done
done > $O/n1_planes.txt 2>&1
for v in "" P22 P11; do
  lib=""; [ -n "$v" ] && lib=$R/dfnet_amd/libvar_$v.so
  [ -n "$v" ] && [ ! -f "$lib" ] && continue
  echo "=== train step '${v:-shipped}'"; DFN_LIB_PATH=$lib timeout 600 python tools/gpu_nerf_train_step.py 2>&1 | tail -3
done >> $O/n1_planes.txt 2>&1
echo "=== tests (shipped)" >> $O/n1_planes.txt
timeout 1500 python -m pytest tests/test_gpu_train.py -q 2>&1 | tail -15 >> $O/n1_planes.txt
grep -n "===\|worst\|step_ms\|passed\|failed\|FAILED" $O/n1_planes.txt | cut -c1-330

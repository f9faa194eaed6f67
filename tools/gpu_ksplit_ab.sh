#!/bin/bash
# Round 6: K split of the 3x3 split-storage convolutions on the small maps (DFN_CONV_KSPLIT=0 off | unset: heuristic | n forced).
# Parity first (DFNet forward / gradients / training steps), then the DFNet_dm and N2 steps A/B on this box, then one traced step each way.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_dfnet.py tests/test_gpu_grad.py -x -q -m gpu -k "dfnet or dm" 2>&1 | tail -2
for rep in 1 2; do
  for V in 0 "" 4; do
    export DFN_CONV_KSPLIT=$V; [ -z "$V" ] && unset DFN_CONV_KSPLIT
    dm=$(DM_ONLY=1 timeout 300 python tools/gpu_dm_step.py 4 40 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['full_step_ms'])")
    dma=$(DM_ONLY=1 DM_ALL_LEVELS=1 timeout 300 python tools/gpu_dm_step.py 4 40 2>/dev/null | python -c "import json,sys; print('%.3f' % json.loads(sys.stdin.read())['full_step_ms'])")
    ft=$(FT_LOOP=1 timeout 300 python tools/gpu_feature_train_step.py 4 30 240 320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d.get('step_ms', d.get('full_step_ms', -1)))")
    echo "KSPLIT='${V}' rep $rep: DFNet_dm step $dm ms, all levels $dma ms, N2 step $ft ms"
  done
done
unset DFN_CONV_KSPLIT
cd /tmp; export TMPDIR=/tmp
for V in 0 ""; do
  export DFN_CONV_KSPLIT=$V; [ -z "$V" ] && unset DFN_CONV_KSPLIT
  rm -rf /tmp/tr_$V; DM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$V -o dm -- python $R/tools/gpu_dm_step.py 4 12 > /dev/null 2>&1
  echo "==== DFNet_dm step, KSPLIT='${V}'"; python $R/tools/gpu_step_convs.py /tmp/tr_$V | grep -E "wgs +(64|96|128|192|256|384|512|576) |step span|conv_x3s" 
done

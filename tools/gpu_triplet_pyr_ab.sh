#!/bin/bash
# Round 6: pyramid triplet path — parity tests (G16, four mining cases vs the stack path) and the N2 step A/B (FT_STACKS=1 = materialised stacks)
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 1200 python -m pytest tests/test_gpu_triplet_pyr.py -x -q -m gpu -s 2>&1 | grep -E "G16|case|passed|failed|Error|assert" | tail -40
for rep in 1 2; do
for V in "stacks:FT_STACKS=1" "pyramid, two passes:FT_TWO_PASS=1" "pyramid, one encoder pass:X=1"; do
  env ${V#*:} FT_LOOP=1 timeout 300 python tools/gpu_feature_train_step.py 4 30 240 320 2>&1 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${V%%:*}', 'step %.3f ms loss %.6f peak %.2f GB' % (d['step_ms'], d['loss'], d['peak_mem_GB']))"
done; done

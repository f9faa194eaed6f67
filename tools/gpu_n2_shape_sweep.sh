#!/bin/bash
# The one-pass N2 step over batch sizes and frame shapes (including widths that are not multiples of 16 and odd totals): every
# combination must run (no "invalid argument" from a scratch plan) and give a finite loss.  tools/gpu_n2_shape_sweep.sh
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for shape in "240 427" "120 214" "64 96" "96 160" "180 320" "72 100" "32 48"; do for B in 1 2 3 4 5; do
  set -- $shape
  echo -n "B=$B ${1}x${2}: "; FT_LOOP=1 timeout 300 python tools/gpu_feature_train_step.py $B 2 $1 $2 2>&1 | python3 -c "
import sys, json
t = sys.stdin.read().strip().splitlines()
try: d = json.loads(t[-1]); print('ok loss %.6f step %.2f ms' % (d['loss'], d['step_ms']))
except Exception: print('FAILED', ' | '.join(l for l in t[-3:])[:400])"
done; done

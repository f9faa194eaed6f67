#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout 1500 python -m pytest tests/test_gpu_grad.py tests/test_gpu_dm_pieces.py -q -x 2>&1 | tail -15
for i in 1 2; do DM_ONLY=1 timeout 600 python tools/gpu_dm_step.py 4 24 2>&1 | tail -1; done
DM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_dm_b -o dm -- python tools/gpu_dm_step.py 4 24 > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_dm_b/dm_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'nerfh_coarse_kernel' in n]
print('launches per steady step', [idx[i+1]-idx[i] for i in range(len(idx)-1)][-5:])
a,b=idx[-3],idx[-2]
t0=int(rows[a]['Start_Timestamp'])
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b])
print('step span us', (int(rows[b]['Start_Timestamp'])-t0)/1e3, 'kernel time sum us', busy/1e3)
PY

#!/bin/bash
# kernel-trace durations of the render's stage kernels (one headline bench run of 3 frames under rocprofv3)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stage
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stage -o s -- python $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extras --precision ${PREC:-f16x3} > /dev/null 2>&1
python3 - <<PY
import csv
for r in csv.DictReader(open("$R/gpurun_out/prof_stage/s_kernel_stats.csv")):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY

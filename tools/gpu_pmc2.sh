#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc2
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2 -o pass$i -- $CMD > $R/gpurun_out/pmc2/pass$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc2/pass*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:30]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in agg.items():
        if "nerfh" in k: print("  ", k, {c: f"{v:.4g}" for c, v in d.items()})
PY

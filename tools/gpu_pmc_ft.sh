#!/bin/bash
# PMC pass over the DFNet feature-training step (tools/gpu_feature_train_step.py): per-kernel MFMA / VALU / wait counters.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_ft; mkdir -p $R/gpurun_out/pmc_ft
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_ft -o p1 -- python $R/tools/gpu_feature_train_step.py 4 1 > $R/gpurun_out/pmc_ft/p1.log 2>&1
echo "pmc rc=$?"
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc_ft/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", 0))[:8]:
        print(k, {c: f"{v:.3g}" for c, v in d.items()})
PY

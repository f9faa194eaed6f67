#!/bin/bash
# PMC passes over the fused training step's kernels: LDS activity / conflicts, instruction mix, waits, HBM bytes (per-dispatch averages;
# FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE counts half of a wide read stream on gfx950) -> gpurun_out/train_step_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
CMD="python $R/tools/gpu_nerf_train_step.py 1536 128 3"
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE"; do
  rm -rf /tmp/pl; timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pl -o p -- $CMD > /dev/null 2>&1
  python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in csv.DictReader(open("/tmp/pl/p_counter_collection.csv")):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
import json, os
path = "$R/gpurun_out/train_step_pmc.json"
out = json.load(open(path)) if os.path.exists(path) and os.environ.get("PMC_APPEND") else {}
for k, d in agg.items():
    if "fused" in k and ("chain" in k or "wgrad_stream" in k):
        print(k, len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in d.items()})
        out.setdefault(k, {"dispatches": len(n[k]), "per_dispatch_average": True}).update({c: v / len(n[k]) for c, v in d.items()})
json.dump(out, open(path, "w"), indent=1)
PY
  export PMC_APPEND=1
done

#!/bin/bash
# per-layer timing of the split weight-gradient stream (+ SQ counters): tools/gpu_wgrad_layers.py under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for PD in 2 1; do
  rm -rf $R/gpurun_out/prof_wgl
  DFN_WGS_PD=$PD timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_wgl -o w -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
  echo "== PD $PD"; python $R/tools/gpu_wgrad_layers.py report $R/gpurun_out/prof_wgl | tee $R/gpurun_out/wgrad_layers_pd$PD.txt
done
rm -rf $R/gpurun_out/pmc_wgl
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wgl -o p1 -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wgl -o p2 -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in sorted(glob.glob("$R/gpurun_out/pmc_wgl/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
        if "wgrad_s" not in k: continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
for k, d in agg.items():
    print(k, {c: f"{v:.4g}" for c, v in d.items()})
    if d.get("SQ_INSTS_MFMA"): print("   VALU/MFMA", d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"], " LDS conflict share", d.get("SQ_LDS_BANK_CONFLICT", 0) / max(d.get("SQ_LDS_IDX_ACTIVE", 1), 1),
          " mfma busy / wave cycles*4", d["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * d["SQ_WAVE_CYCLES"]))
PY

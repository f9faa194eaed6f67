#!/bin/bash
# SQ counters of the split-f16 DFNet forward (4 x 480x640), per kernel and grid: gpurun_out/dfnet_pmc.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d /tmp/pm -o p -- python $R/tools/gpu_dfnet_layers.py run > /dev/null 2>&1
python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open("/tmp/pm/p_counter_collection.csv")))
gk = [k for k in rows[0].keys() if "Grid" in k]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in rows:
    k = row["Kernel_Name"].replace("dfn::", "").replace("void ", "").split("(")[0] + " grid " + "x".join(row[g] for g in gk)
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
out = {}
for k, d in agg.items():
    w = d["SQ_WAVE_CYCLES"]
    if w <= 0: continue
    out[k] = {"dispatches": len(n[k]),
              # SQ_WAVE_CYCLES counts quad-cycles per wave, MFMA_BUSY cycles per SIMD: with W waves per SIMD, busy fraction = ratio * W / 4
              "mfma_busy_cycles_per_wave_quadcycle": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / w, 3),
              "wait_any": round(d["SQ_WAIT_ANY"] / w, 3), "wait_inst_any": round(d["SQ_WAIT_INST_ANY"] / w, 3),
              "active_inst_any": round(d["SQ_ACTIVE_INST_ANY"] / w, 3), "valu_per_mfma": round(d["SQ_INSTS_VALU"] / max(d["SQ_INSTS_MFMA"], 1), 2)}
json.dump(out, open("$R/gpurun_out/dfnet_pmc.json", "w"), indent=1)
for k, v in out.items():
    if "conv" in k: print(k[:64], v)
PY

#!/bin/bash
# SQ counters per kernel of an arbitrary workload: tools/gpu_pmc.sh NAME cmd...  -> gpurun_out/NAME_pmc.json
# (counters only with --kernel-trace: the node pool refuses PMC combined with API tracing)
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm_$NAME
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA \
  --kernel-trace --output-format csv -d /tmp/pm_$NAME -o p -- "$@" > /dev/null 2>&1
python3 - <<PY
import csv, collections, json
rows = list(csv.DictReader(open("/tmp/pm_$NAME/p_counter_collection.csv")))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in rows:
    k = row["Kernel_Name"].replace("dfn::", "").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
out = {}
for k, d in agg.items():
    w = d["SQ_WAVE_CYCLES"]
    if w <= 0 or d["SQ_INSTS_MFMA"] <= 0: continue
    out[k] = {"dispatches": len(n[k]),
              # SQ_WAVE_CYCLES counts quad-cycles per wave, MFMA_BUSY cycles per SIMD: with W waves per SIMD, busy fraction = ratio * W / 4
              "mfma_busy_cycles_per_wave_quadcycle": round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / w, 3),
              "wait_any": round(d["SQ_WAIT_ANY"] / w, 3), "wait_inst_any": round(d["SQ_WAIT_INST_ANY"] / w, 3),
              "active_inst_any": round(d["SQ_ACTIVE_INST_ANY"] / w, 3), "valu_per_mfma": round(d["SQ_INSTS_VALU"] / max(d["SQ_INSTS_MFMA"], 1), 2),
              "mfma_insts": d["SQ_INSTS_MFMA"]}
json.dump(out, open("$R/gpurun_out/${NAME}_pmc.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_insts"])[:14]: print(k[:70], v)
PY
# HBM traffic of the same command, every kernel (also the ones without matrix instructions): FETCH_SIZE and WRITE_SIZE in their own
# passes (TCC: 3 + 2 slots, never together), KiB per dispatch -> bytes per dispatch.  MI355X_MICROARCH.md "HBM": on gfx950 FETCH_SIZE
# reports half the bytes of a wide coalesced read stream -> `fetch_bytes_x2` is the corrected figure to hold against a byte model;
# WRITE_SIZE is uncalibrated (stored as counted).  PMC_NO_HBM=1 skips the two passes.
if [ -z "${PMC_NO_HBM:-}" ]; then
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pm_${NAME}_$C
    rocprofv3 --pmc GRBM_GUI_ACTIVE $C --kernel-trace --output-format csv -d /tmp/pm_${NAME}_$C -o p -- "$@" > /dev/null 2>&1
  done
  python3 - <<PY
import csv, collections, json, os
path = "$R/gpurun_out/${NAME}_pmc.json"
out = json.load(open(path)) if os.path.exists(path) else {}
for C, key in (("FETCH_SIZE", "fetch_bytes"), ("WRITE_SIZE", "write_bytes")):
    f = "/tmp/pm_${NAME}_%s/p_counter_collection.csv" % C
    if not os.path.exists(f):
        continue
    agg = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        if row["Counter_Name"] != C:
            continue
        k = row["Kernel_Name"].replace("dfn::", "").replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        agg[k] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, v in agg.items():
        rec = out.setdefault(k, {"dispatches": len(n[k])})
        rec[key + "_per_dispatch"] = v * 1024.0 / max(len(n[k]), 1)
        if C == "FETCH_SIZE":
            rec["fetch_bytes_x2_per_dispatch"] = 2.0 * rec[key + "_per_dispatch"]
json.dump(out, open(path, "w"), indent=1)
top = sorted(((k, v) for k, v in out.items() if "fetch_bytes_per_dispatch" in v), key=lambda kv: -kv[1]["fetch_bytes_per_dispatch"] * kv[1]["dispatches"])[:10]
for k, v in top: print(k[:70], "fetch x2 %.1f MB  write %.1f MB per dispatch" % (v["fetch_bytes_x2_per_dispatch"] / 1e6, v.get("write_bytes_per_dispatch", 0) / 1e6))
PY
fi

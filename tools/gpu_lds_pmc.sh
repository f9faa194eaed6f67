#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT_INST_[A-Z]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u | tr '\n' ' '; echo
CMD="python $R/bench.py --precision f16x3 --steps 2 --warmup 1 --cpu-sample 0 --no-extras"
rm -rf /tmp/pl; timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM --kernel-trace --output-format csv -d /tmp/pl -o p -- $CMD > /dev/null 2>&1
python3 - <<PY
import csv, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in csv.DictReader(open("/tmp/pl/p_counter_collection.csv")):
    k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
for k, d in agg.items():
    if "nerfh" in k: print(k, len(n[k]), {c: f"{v/len(n[k]):.4g}" for c, v in d.items()})
PY

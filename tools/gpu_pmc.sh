#!/bin/bash
# PMC passes over the bench (separate runs, --kernel-trace only alongside --pmc; never with sys/hip trace).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|GRBM_[A-Z0-9_]+|FETCH_SIZE|WRITE_SIZE|MfmaUtil|VALUBusy|LDSBankConflict)\b" | sort -u > $R/gpurun_out/pmc/counters.txt
wc -l $R/gpurun_out/pmc/counters.txt
CMD="python $R/bench.py --steps 2 --warmup 1 --cpu-sample 0"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc -o pass$i -- $CMD > $R/gpurun_out/pmc/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
ls $R/gpurun_out/pmc | head -30
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$R/gpurun_out/pmc/pass*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:40]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    print(f.split("/")[-1])
    for k, d in agg.items():
        print("  ", k, {c: f"{v:.4g}" for c, v in d.items()})
PY

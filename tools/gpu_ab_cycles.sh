#!/bin/bash
# Same-box A/B of library builds in CYCLES, not milliseconds: GRBM_GUI_ACTIVE (all 8 XCDs) and the kernel time per fine / coarse launch,
# so that a power-limited part (clock follows the load) does not hide or fake a change.  usage: tools/gpu_ab_cycles.sh PREC lib1.so lib2.so ...
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp
PREC=$1; shift
for lib in "$@"; do
  rm -rf /tmp/abc; DFN_LIB_PATH=$R/dfnet_amd/$lib PREC=$PREC timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/abc -o p -- python $R/tools/gpu_ablate.py child > /tmp/abc.log 2>&1
  python3 - "$lib" <<'PY'
import csv, collections, sys, glob
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in csv.DictReader(open("/tmp/abc/p_counter_collection.csv")):
    k = row["Kernel_Name"].split("<")[0].replace("void ", "")
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
dur = collections.defaultdict(list)
for row in csv.DictReader(open("/tmp/abc/p_kernel_trace.csv")):
    dur[row["Kernel_Name"].split("<")[0].replace("void ", "")].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for k in agg:
    if "nerfh_fine" in k or "nerfh_coarse" in k:
        c = agg[k]["GRBM_GUI_ACTIVE"] / len(n[k]) / 8
        ms = sum(dur[k]) / len(dur[k]) / 1e6
        print(f"{sys.argv[1]:24s} {k:28s} {c/1e6:7.3f} M cycles  {ms:7.3f} ms  -> {c/ms/1e6:.2f} GHz")
PY
done

#!/bin/bash
# Copy the summaries of the last tools/gpu_round.sh pass from gpurun_out/ (scratch) into profiles/ (tracked): usage tools/collect_profiles.sh r03
T=${1:-r06}; cd "$(dirname "$0")/.."; G=gpurun_out; P=profiles
for prec in f16x3 f32 f16; do
  [ -f $G/prof_$prec/${T}_kernel_stats.csv ] && cp $G/prof_$prec/${T}_kernel_stats.csv $P/${T}_kernel_stats_$prec.csv
  [ -f $G/pmc_$prec/summary.json ] && cp $G/pmc_$prec/summary.json $P/${T}_pmc_summary_$prec.json
  [ -f $G/prof_bench_$prec.json ] && cp $G/prof_bench_$prec.json $P/${T}_bench_under_rocprof_$prec.json
done
[ -f $G/bench.json ] && cp $G/bench.json $P/${T}_bench.json
[ -f $G/prof_dm/dm_kernel_stats.csv ] && cp $G/prof_dm/dm_kernel_stats.csv $P/${T}_dm_step_kernel_stats.csv
[ -f $G/prof_train/tr_kernel_stats.csv ] && cp $G/prof_train/tr_kernel_stats.csv $P/${T}_train_step_kernel_stats.csv
[ -f $G/dfnet_layers.txt ] && cp $G/dfnet_layers.txt $P/${T}_dfnet_layers.txt
[ -f $G/dm_step.json ] && python3 - $G/dm_step.json $G/prof_dm/dm_kernel_stats.csv $P/${T}_dm_step.json <<'PY'
# the step's JSON + launches per step and library-kernel census from the kernel trace of the same run
import csv, json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
try:
    rows = list(csv.DictReader(open(sys.argv[2])))
    it = int(d.get("iters_profiled", 25))
    d["launches_per_step"] = round(sum(int(r["Calls"]) for r in rows) / it, 1)
    d["launches_under_10us_per_step"] = round(sum(int(r["Calls"]) for r in rows if float(r["AverageNs"]) < 1e4) / it, 1)
    d["kernel_ms_per_step"] = round(sum(float(r["TotalDurationNs"]) for r in rows) / 1e6 / it, 3)
    try:   # a STEADY step (the average above carries the first step's weight uploads): launches and idle time between two coarse-kernel launches
        tr = list(csv.DictReader(open(sys.argv[2].replace("kernel_stats", "kernel_trace"))))
        tr.sort(key=lambda r: int(r["Start_Timestamp"]))
        idx = [i for i, r in enumerate(tr) if "nerfh_coarse_kernel" in r["Kernel_Name"]]
        a, b = idx[-3], idx[-2]
        d["launches_per_steady_step"] = b - a
        end, idle = int(tr[a]["Start_Timestamp"]), 0
        for r in tr[a:b + 1]:
            st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            if st - end > 6000:
                idle += st - end
            end = max(end, en)
        d["gpu_idle_ms_per_steady_step"] = round(idle / 1e6, 3)
        d["steady_step_span_ms"] = round((int(tr[b]["Start_Timestamp"]) - int(tr[a]["Start_Timestamp"])) / 1e6, 3)
    except Exception as e:
        d["steady_step_census_error"] = str(e)
    d["vendor_library_kernels"] = sorted({r["Name"][:60] for r in rows if any(k in r["Name"] for k in ("Cijk_", "rocsolver", "rocblas"))})
    d["source"] = "tools/gpu_dm_step.py 4 24 under rocprofv3 --kernel-trace --stats (profiles/%s_dm_step_kernel_stats.csv)" % sys.argv[3].split("/")[-1].split("_")[0]
except Exception as e:
    d["launch_census_error"] = str(e)
json.dump(d, open(sys.argv[3], "w"), indent=1)
PY
[ -f $G/train_step.json ] && cp $G/train_step.json $P/${T}_train_step.json
[ -f $G/train_step_pmc.json ] && cp $G/train_step_pmc.json $P/${T}_train_step_pmc.json
[ -f $G/prof_ft/ft_kernel_stats.csv ] && cp $G/prof_ft/ft_kernel_stats.csv $P/${T}_feature_train_kernel_stats.csv
[ -f $G/ft_step.json ] && cp $G/ft_step.json $P/${T}_feature_train.json
[ -f $G/ft_step_pmc.json ] && cp $G/ft_step_pmc.json $P/${T}_feature_train_pmc.json
[ -f $G/dm_step_pmc.json ] && cp $G/dm_step_pmc.json $P/${T}_dm_step_pmc.json
[ -f $G/wgrad_layers.txt ] && cp $G/wgrad_layers.txt $P/${T}_wgrad_layers.txt
[ -f $G/trained_parity.log ] && cp $G/trained_parity.log $P/${T}_trained_parity.log
[ -f $G/dm_step_convs.txt ] && cp $G/dm_step_convs.txt $P/${T}_dm_step_convs.txt
[ -f $G/dm_step_all_levels.json ] && cp $G/dm_step_all_levels.json $P/${T}_dm_step_all_levels.json
for f in r06_ks_abl2.log r06_triplet_pyr.log r06_n1_ab.log; do [ "$T" = r06 ] && [ -f $G/$f ] && cp $G/$f $P/$f; done
ls -la $P | grep $T

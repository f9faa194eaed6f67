#!/bin/bash
# One GPU-box pass: parity tests, smoke, bench, rocprof kernel stats, PMC passes.  Outputs under gpurun_out/.
# usage: tools/gpu_round.sh TAG [notest] [nopmc]
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06}
cd $R
mkdir -p gpurun_out
if [[ " $* " != *" notest "* ]]; then
  timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
  tail -6 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
  # the trained-weight parity figures DESIGN.md section 6 quotes (printed by the tests: mode by mode, stage by stage, create_nerf, CLI)
  timeout 900 python -m pytest tests/test_gpu_nerfh.py tests/test_gpu_cli.py -q -m gpu -s -k "trained" 2>&1 | grep -oE "(trained (stages|weights|checkpoint)|create_nerf engine).*" > gpurun_out/trained_parity.log
  wc -l gpurun_out/trained_parity.log
fi
timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-3000 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
cd /tmp && export TMPDIR=/tmp
# rocprofv3 kernel stats + three PMC passes of the SAME command per arithmetic mode (the headline's f16x3, exact f32, f16)
for PREC in f16x3 f32 f16; do
  rm -rf $R/gpurun_out/prof_$PREC $R/gpurun_out/pmc_$PREC
  CMD="python $R/bench.py --precision $PREC --steps 3 --warmup 1 --cpu-sample 0 --no-extras"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$PREC -o $TAG -- $CMD > $R/gpurun_out/prof_bench_$PREC.json 2> $R/gpurun_out/prof_$PREC.err; echo "rocprof $PREC rc=$?"
  head -6 $R/gpurun_out/prof_$PREC/${TAG}_kernel_stats.csv | cut -c1-170
  if [[ " $* " != *" nopmc "* ]]; then
    mkdir -p $R/gpurun_out/pmc_$PREC
    i=0
    for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
               "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
      i=$((i+1))
      timeout 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$PREC -o pass$i -- $CMD > $R/gpurun_out/pmc_$PREC/pass$i.log 2>&1
      echo "pmc $PREC pass $i rc=$?"
    done
    python3 - <<PY
import csv, glob, collections, json
out = {}
for f in sorted(glob.glob("$R/gpurun_out/pmc_$PREC/pass*_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")[:72]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); n[k].add(row["Dispatch_Id"])
    for k, d in agg.items():
        out.setdefault(k, {"dispatches": len(n[k])}).update({c: v for c, v in d.items()})
json.dump(out, open("$R/gpurun_out/pmc_$PREC/summary.json", "w"), indent=1)
for k, d in out.items():
    if "nerfh" in k: print(k, {c: (f"{v:.4g}" if isinstance(v, float) else v) for c, v in d.items()})
PY
  fi
done
# Secondary workloads: kernel stats of the DFNet_dm step and the NeRF-H training step, per-layer table of the DFNet forward.
if [[ " $* " != *" noextra "* ]]; then
  cd /tmp
  rm -rf $R/gpurun_out/prof_dm $R/gpurun_out/prof_train $R/gpurun_out/prof_layers
  DM_ONLY=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dm -o dm -- python $R/tools/gpu_dm_step.py 4 24 > $R/gpurun_out/dm_step.json 2> $R/gpurun_out/dm_step.err; echo "dm rc=$?"
  python $R/tools/gpu_step_convs.py $R/gpurun_out/prof_dm > $R/gpurun_out/dm_step_convs.txt 2>&1; tail -4 $R/gpurun_out/dm_step_convs.txt
  DM_ONLY=1 DM_ALL_LEVELS=1 timeout 300 python $R/tools/gpu_dm_step.py 4 24 > $R/gpurun_out/dm_step_all_levels.json 2>/dev/null; cat $R/gpurun_out/dm_step_all_levels.json
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_train -o tr -- python $R/tools/gpu_nerf_train_step.py 1536 128 10 > $R/gpurun_out/train_step.json 2> $R/gpurun_out/train_step.err; echo "train rc=$?"
  rm -f $R/gpurun_out/train_step_pmc.json; timeout -k 5 900 $R/tools/gpu_train_pmc.sh > $R/gpurun_out/train_step_pmc.log 2>&1; echo "train pmc rc=$?"
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_layers -o l -- python $R/tools/gpu_dfnet_layers.py run > /dev/null 2>&1
  python $R/tools/gpu_dfnet_layers.py report $R/gpurun_out/prof_layers > $R/gpurun_out/dfnet_layers.txt; head -3 $R/gpurun_out/dfnet_layers.txt
  rm -rf $R/gpurun_out/prof_layers
  # round 5: the N2 step (DFNet's own training) — kernel stats and SQ counters — the DFNet_dm step's counters, the split weight-gradient
  # stream layer by layer
  cd /tmp
  rm -rf $R/gpurun_out/prof_ft
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ft -o ft -- env FT_LOOP=1 python $R/tools/gpu_feature_train_step.py 4 20 240 320 > $R/gpurun_out/ft_step.json 2> $R/gpurun_out/ft_step.err; echo "ft rc=$?"
  timeout 600 $R/tools/gpu_pmc.sh ft_step env FT_LOOP=1 python $R/tools/gpu_feature_train_step.py 4 6 240 320 > $R/gpurun_out/ft_step_pmc.log 2>&1; echo "ft pmc rc=$?"
  DM_ONLY=1 timeout 600 $R/tools/gpu_pmc.sh dm_step python $R/tools/gpu_dm_step.py 4 8 > $R/gpurun_out/dm_step_pmc.log 2>&1; echo "dm pmc rc=$?"
  cd /tmp; rm -rf $R/gpurun_out/prof_wgl
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_wgl -o w -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
  python $R/tools/gpu_wgrad_layers.py report $R/gpurun_out/prof_wgl > $R/gpurun_out/wgrad_layers.txt; tail -1 $R/gpurun_out/wgrad_layers.txt
  rm -rf $R/gpurun_out/prof_wgl
fi

#!/bin/bash
# Copy the summaries of the last tools/gpu_round.sh pass from gpurun_out/ (scratch) into profiles/ (tracked): usage tools/collect_profiles.sh r03
T=${1:-r05}; cd "$(dirname "$0")/.."; G=gpurun_out; P=profiles
for prec in f16x3 f32 f16; do
  [ -f $G/prof_$prec/${T}_kernel_stats.csv ] && cp $G/prof_$prec/${T}_kernel_stats.csv $P/${T}_kernel_stats_$prec.csv
  [ -f $G/pmc_$prec/summary.json ] && cp $G/pmc_$prec/summary.json $P/${T}_pmc_summary_$prec.json
  [ -f $G/prof_bench_$prec.json ] && cp $G/prof_bench_$prec.json $P/${T}_bench_under_rocprof_$prec.json
done
[ -f $G/bench.json ] && cp $G/bench.json $P/${T}_bench.json
[ -f $G/prof_dm/dm_kernel_stats.csv ] && cp $G/prof_dm/dm_kernel_stats.csv $P/${T}_dm_step_kernel_stats.csv
[ -f $G/prof_train/tr_kernel_stats.csv ] && cp $G/prof_train/tr_kernel_stats.csv $P/${T}_train_step_kernel_stats.csv
[ -f $G/dfnet_layers.txt ] && cp $G/dfnet_layers.txt $P/${T}_dfnet_layers.txt
[ -f $G/dm_step.json ] && cp $G/dm_step.json $P/${T}_dm_step.json
[ -f $G/train_step.json ] && cp $G/train_step.json $P/${T}_train_step.json
[ -f $G/train_step_pmc.json ] && cp $G/train_step_pmc.json $P/${T}_train_step_pmc.json
[ -f $G/prof_ft/ft_kernel_stats.csv ] && cp $G/prof_ft/ft_kernel_stats.csv $P/${T}_feature_train_kernel_stats.csv
[ -f $G/ft_step.json ] && cp $G/ft_step.json $P/${T}_feature_train.json
[ -f $G/ft_step_pmc.json ] && cp $G/ft_step_pmc.json $P/${T}_feature_train_pmc.json
[ -f $G/dm_step_pmc.json ] && cp $G/dm_step_pmc.json $P/${T}_dm_step_pmc.json
[ -f $G/wgrad_layers.txt ] && cp $G/wgrad_layers.txt $P/${T}_wgrad_layers.txt
ls -la $P | grep $T

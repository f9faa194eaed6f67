R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_dfnet.py tests/test_gpu_grad.py -q -k "wgrad or triplet or all_parameter or training_step or kept_forward or conv0 or parameter_gradients" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp; rm -rf $R/gpurun_out/prof_wgl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_wgl -o w -- python $R/tools/gpu_wgrad_layers.py run > /dev/null 2>&1
python $R/tools/gpu_wgrad_layers.py report $R/gpurun_out/prof_wgl | tail -4
cd $R
for i in 1 2; do python tools/gpu_feature_train_step.py 4 20 240 320 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ft', round(d['step_ms'],3), {k: round(v,2) for k,v in d['breakdown_ms_with_syncs'].items()})"; done

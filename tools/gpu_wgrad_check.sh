cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wgrad.py tests/test_gpu_pose.py tests/test_gpu_train.py -x -q -k "wgrad or pose or recovers or mode_switch" -s 2>&1 | tail -40
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_wgl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_wgl -o w -- python $GRAFT_REPO_ROOT/tools/gpu_wgrad_layers.py run > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/wgl.err; echo "wgl rc=$?"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/wgl.err
python $GRAFT_REPO_ROOT/tools/gpu_wgrad_layers.py report $GRAFT_REPO_ROOT/gpurun_out/prof_wgl | tee $GRAFT_REPO_ROOT/gpurun_out/wgrad_layers.txt

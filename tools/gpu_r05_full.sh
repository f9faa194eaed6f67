#!/bin/bash
# full GPU parity suite, smoke, and the default bench
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 2700 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"
grep -E "^(FAILED|ERROR)|passed|failed" gpurun_out/pytest_gpu.log | tail -20
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cut -c1-2500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err

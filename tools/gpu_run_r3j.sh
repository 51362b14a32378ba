#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 200 tools/tile_bench 50 > gpurun_out/r3j_tile_bench.txt 2>&1; echo "tile_bench rc=$?"; tail -1 gpurun_out/r3j_tile_bench.txt
timeout 900 python -m pytest tests/test_gpu_f16.py -q -x --timeout 400 -p no:cacheprovider > gpurun_out/r3j_pytest_f16.log 2>&1
echo "pytest f16 rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -30 gpurun_out/r3j_pytest_f16.log

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING APRIL_BACKTRACE=1
timeout 400 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "not 60s and not larger and not 2048 and not torch and not churn and not above_max and not config3" > gpurun_out/h_lm_test.log 2>&1
echo rc=$?; grep -n "libapril\|aprilx\|libamdhip\|passed\|failed\|^tests" gpurun_out/h_lm_test.log | head -60; head -3 gpurun_out/h_lm_test.log

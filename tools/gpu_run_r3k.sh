#!/bin/bash
# long feed: bitwise tests, then 60 s in one call
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_recur_kernels.py -m gpu -q -x --timeout 400 -p no:cacheprovider -k "layer_major or 60s or recur or stream_kernels" > gpurun_out/r3k_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r3k_pytest.log
for k in 1 2; do
  LM_PROBE_REPS=4 timeout 300 python tools/lm_probe.py v0 60 2>&1 | grep -E "feed ok|mismatch"
done

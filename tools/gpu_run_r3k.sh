#!/bin/bash
# recurrent weight-stream kernels: bitwise tests that cross the kernel boundary, then the long feed (60 s in one call) with and without them
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -p no:cacheprovider -k "layer_major or batch_invariant or many_sessions or matches_oracle or 60s or odd_dimensions or torch_fp32 or transcript" > gpurun_out/r3k_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r3k_pytest.log
for k in 0 1 0 1; do
  echo "== APRIL_RECUR_KERNELS=$k"
  APRIL_RECUR_KERNELS=$k LM_PROBE_REPS=3 timeout 300 python tools/lm_probe.py v0 60 2>&1 | grep -E "feed ok|mismatch"
done

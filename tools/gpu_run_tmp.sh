#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
timeout 1200 python -m pytest tests/test_gpu_recur_kernels.py tests/test_gpu_parity.py -m gpu -q -x --timeout 400 -p no:cacheprovider > gpurun_out/tmp_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/tmp_pytest.log
for k in 0 1; do
  echo "== APRIL_RECUR_KERNELS=$k"
  for n in 1 16; do
  APRIL_RECUR_KERNELS=$k timeout 300 python bench.py --sessions $n --no-sweep --no-config5 --no-cpu-baseline --steady-steps 60 --steps 20 --warmup 5 2>gpurun_out/tmp_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('sessions', d['config']['sessions_per_gpu'] if 'sessions_per_gpu' in d['config'] else '?', d['ms_per_step'], d['steady']['ms_per_step'], d['replay_mismatch'])" || tail -3 gpurun_out/tmp_err.txt
  done
  APRIL_RECUR_KERNELS=$k LM_PROBE_REPS=3 timeout 300 python tools/lm_probe.py v0 60 2>&1 | grep -E "feed ok" | tail -2
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
t0=$(date +%s)
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli.py tests/test_gpu_reference_suite.py -m gpu -q -x --timeout 200 -p no:cacheprovider \
  -k "not 60s and not larger and not torch and not config3" > gpurun_out/i_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -4 gpurun_out/i_pytest.log
for v in "APRIL_HOST_THREADS=3" "APRIL_HOST_THREADS=8"; do
  for b in 256 2048 2304; do
    env $v timeout 200 python bench.py --sessions $b --steps 16 --warmup 4 --no-sweep --no-cpu-baseline --profile-steps 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', $b, 'ms/step', d['ms_per_step'], 'rtf', d['rtf'], 'host', d['host_phase_ms_total'])"
  done
done > gpurun_out/i_host.txt 2>&1
cat gpurun_out/i_host.txt

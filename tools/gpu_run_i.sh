#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING APRIL_BACKTRACE=1
for b in 256 1 64 1024 256; do
  echo "== B=$b: $(timeout 200 python bench.py --sessions $b --steps 50 --warmup 10 --no-cpu-baseline --no-sweep --profile-steps 0 2>gpurun_out/i_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_latency_ms']['p50'], d['replay_mismatch'], d['callbacks'], d['tokens_in_callbacks'])")"
done
timeout 400 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_suite.py tests/test_gpu_cli.py -m gpu -q -x --timeout 200 -p no:cacheprovider -k "not 60s" > gpurun_out/i_parity.log 2>&1; echo rc=$?; grep -E "passed|failed|error" gpurun_out/i_parity.log | tail -3

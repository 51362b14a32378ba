#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING APRIL_BACKTRACE=1
for v in "X=1" "APRIL_WAVE_MAX_CHUNKS=2" "APRIL_WAVE_MIN_CHUNKS=0"; do
 for b in 2048 1792 2304; do
  echo "== $v B=$b: $(env $v timeout 200 python bench.py --sessions $b --steps 30 --warmup 8 --no-cpu-baseline --no-sweep --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_latency_ms']['p50'], d['step_latency_ms']['max'], d['replay_mismatch'], d['host_phase_ms_total'])")"
 done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING APRIL_BACKTRACE=1
for b in 256 2048 64 256 2048; do
  echo "== B=$b: $(timeout 200 python bench.py --sessions $b --steps 40 --warmup 10 --no-cpu-baseline --no-sweep --profile-steps 0 2>gpurun_out/i_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); n=d['steps']+d['warmup']; print(d['ms_per_step'], d['step_latency_ms']['p50'], d['replay_mismatch'], [round(x/n,3) for x in d['host_phase_ms_total']])")"
done
python __graft_entry__.py smoke 2>&1 | tail -3

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
for gt in 0 1 0 1; do
  APRIL_GATES_TILE=$gt timeout 300 python bench.py --no-cpu-baseline --no-config5 --steady-steps 60 > gpurun_out/r3q_gt$gt.json 2>> gpurun_out/r3q.err; python -c "
import json; d=json.load(open('gpurun_out/r3q_gt$gt.json'))
print('gates_tile=$gt', d['ms_per_step'], d['steady']['ms_per_step'], d['rtf_by_sessions_per_gpu'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['replay_mismatch'])"
done
APRIL_GATES_TILE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --timeout 400 -p no:cacheprovider -k "invariant or many_sessions or config3 or wavefront" 2>&1 | tail -3

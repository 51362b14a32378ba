#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
for i in 1 2 3; do
timeout 300 python bench.py --no-cpu-baseline --no-sweep --profile-steps 0 > gpurun_out/i_bench$i.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/i_bench$i.json')); print(d['ms_per_step'], d['step_latency_ms']['series'], d['host_phase_ms_total'])"
done

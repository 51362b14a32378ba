#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
tag=r03
bash tools/trace_pass.sh ${tag}_b2048 --sessions 2048 --steps 8 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0 > /dev/null
bash tools/trace_pass.sh ${tag}_b256 --steps 10 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0 > /dev/null
for b in b256 b2048; do f=$(ls /tmp/trace/${tag}_$b/*kernel_trace.csv | head -1); python tools/gap_summary.py "$f" > gpurun_out/${tag}_${b}_gap_summary.txt; cat gpurun_out/${tag}_${b}_gap_summary.txt; done

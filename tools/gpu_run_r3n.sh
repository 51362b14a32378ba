#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o c5 -- python bench.py --config5-only --profile-steps 0 > gpurun_out/r3n_config5.json 2> gpurun_out/r3n.err
f=$(ls /tmp/prof5/*kernel_stats.csv | head -1); cp $f gpurun_out/r3n_config5_kernel_stats.csv; head -30 $f | cut -c1-230
t=$(ls /tmp/prof5/*kernel_trace.csv | head -1); python tools/gap_summary.py $t > gpurun_out/r3n_config5_gap_summary.txt; cat gpurun_out/r3n_config5_gap_summary.txt

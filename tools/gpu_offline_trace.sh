#!/bin/bash
# kernel trace of the long feed (eager launches: rocprofv3 crashes on the replayed search graphs)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export APRIL_LOG_LEVEL=WARNING APRIL_NO_GRAPHS=1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lmprof -o lm -- python $R/tools/lm_probe.py v0 60 > $R/gpurun_out/offline_probe.log 2>&1
echo "rc=$?"; tail -5 $R/gpurun_out/offline_probe.log
f=$(find /tmp/lmprof -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/offline_lm_kernel_stats.csv
t=$(find /tmp/lmprof -name "*kernel_trace.csv" | head -1)
python $R/tools/concurrency_summary.py $t > $R/gpurun_out/offline_lm_trace_summary.txt 2>&1
head -40 $R/gpurun_out/offline_lm_trace_summary.txt

#!/bin/bash
# usage: tools/trace_pass.sh <tag> <bench args...>  -- rocprofv3 --kernel-trace --stats, keeps only the per-kernel stats CSV + bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out /tmp/trace
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace/$tag -o t -- python bench.py "$@" > /tmp/trace/$tag.log 2>&1 || echo "trace failed"
f=$(ls /tmp/trace/$tag/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/${tag}_kernel_stats.csv
grep '^{' /tmp/trace/$tag.log | tail -1 > gpurun_out/${tag}_bench.json
ls /tmp/trace/$tag | head

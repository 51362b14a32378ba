#!/bin/bash
# usage: tools/pmc_cache_pass.sh <tag> <bench args...>  -- where the operands come from: L2 (TCC) hits / misses and the L1 -> L2 read
# requests of every kernel, separate rocprofv3 --pmc passes (kernel-trace only), per-kernel summary kept
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out /tmp/pmc
rocprofv3 --list-avail 2>/dev/null | grep -o -E "\b(TCC_HIT_sum|TCC_MISS_sum|TCC_REQ_sum|TCC_READ_sum|TCP_TCC_READ_REQ_sum|TCC_EA0_RDREQ_sum|TCP_TOTAL_CACHE_ACCESSES_sum|TCC_TAG_STALL_sum|TCC_BUBBLE_sum)\b" | sort -u > gpurun_out/pmc_${tag}_avail.txt
i=0
for ctrs in "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc/${tag}_c$i -o p -- python bench.py "$@" > /tmp/pmc/${tag}_c$i.log 2>&1 || echo "pass $i failed/timeout"
done
python tools/pmc_summary.py /tmp/pmc/${tag}_c* > gpurun_out/pmc_${tag}_cache_summary.txt
wc -c gpurun_out/pmc_${tag}_cache_summary.txt gpurun_out/pmc_${tag}_avail.txt

#!/bin/bash
# usage: tools/pmc_pass.sh <tag> <bench args...>
# Separate rocprofv3 --pmc passes (kernel-trace only), each bounded by `timeout`; only the per-kernel summary is kept.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
i=0
mkdir -p gpurun_out /tmp/pmc
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES" \
            "GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc/${tag}_$i -o p -- python bench.py "$@" > /tmp/pmc/${tag}_$i.log 2>&1 || echo "pass $i failed/timeout"
done
python tools/pmc_summary.py /tmp/pmc/${tag}_* > gpurun_out/pmc_${tag}_summary.txt
grep '^{' /tmp/pmc/${tag}_1.log | tail -1 > gpurun_out/pmc_${tag}_bench.json
wc -c gpurun_out/pmc_${tag}_summary.txt

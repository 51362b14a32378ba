#!/bin/bash
# PMC pass on tools/tile_bench: effective clock (GRBM_GUI_ACTIVE / duration) and SQ cycle breakdown of the GM_TILE kernel,
# shape 4 (ffdn 256x1, one workgroup per CU) and shape 13 (ffdn 2048x2, two workgroups per CU), fused 64-row tiles
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out /tmp/pmc
for sh in 4 13; do
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_WAVES" \
              "GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc/t${sh}_$i -o p -- tools/tile_bench 30 $sh 4 8 > /tmp/pmc/t${sh}_$i.log 2>&1 || echo "pass $sh/$i failed"
  done
  python tools/pmc_summary.py /tmp/pmc/t${sh}_* > gpurun_out/r3e_pmc_shape${sh}.txt
  tail -3 /tmp/pmc/t${sh}_1.log
done
cat gpurun_out/r3e_pmc_shape4.txt gpurun_out/r3e_pmc_shape13.txt | cut -c1-600

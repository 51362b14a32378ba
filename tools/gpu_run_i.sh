#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "APRIL_GEMM_ASM=2 APRIL_GEMM_SKEW=0" "APRIL_GEMM_ASM=0 APRIL_GEMM_TUNE=2 APRIL_GEMM_SKEW=0" "APRIL_GEMM_ASM=0 APRIL_GEMM_TUNE=2 APRIL_GEMM_SKEW=2" "APRIL_GEMM_ASM=0 APRIL_GEMM_TUNE=1 APRIL_GEMM_SKEW=0" "APRIL_GEMM_ASM=0 APRIL_GEMM_SKEW=0"; do
  echo "== $v"
  for shape in "2048 4096 1024 1 1" "1024 4096 1024 1 1" "512 4096 1024 1 1" "2048 2048 512 2 1" "512 2048 512 2 1"; do env $v timeout 60 tools/gemm_bench $shape 100 12; done
done > gpurun_out/i_gates2.txt 2>&1
cat gpurun_out/i_gates2.txt

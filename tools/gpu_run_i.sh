#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for shape in "256 512 2048 4 4" "256 512 1024 3 4" "256 2048 512 2 1" "256 512 2048 4 8" "256 512 1024 3 8"; do GEMM_TRACE=1 GEMM_TRACE_CU=1 timeout 60 tools/gemm_bench_trace $shape 200 12; done > gpurun_out/i_trace.txt 2>&1
cat gpurun_out/i_trace.txt

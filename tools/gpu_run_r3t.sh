#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# shape 4 = ffdn 256x1 (fused mt=4: one workgroup per CU), shape 16/17 don't exist; single-problem shapes only (trace needs n == 1)
for args in "4 4 8" "4 2 8" "4 4 1" "7 4 2"; do TB_TRACE=1 timeout 100 tools/tile_bench_trace 50 $args 2>&1 | grep -E "^[a-z]|tile mt|trace"; done > gpurun_out/r3t_trace.txt 2>&1
for d in 4 5; do APRIL_GEMM_DEBUG=$d TB_TRACE=1 timeout 100 tools/tile_bench_trace 50 4 4 8 2>&1 | grep -E "tile mt|trace" | sed "s/^/debug$d /"; done >> gpurun_out/r3t_trace.txt 2>&1
cat gpurun_out/r3t_trace.txt

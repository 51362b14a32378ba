#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "4 4 8" "4 2 8"; do TB_TRACE=1 timeout 100 tools/tile_bench_trace 50 $args 2>&1 | grep -E "tile mt|trace"; done
for d in 4; do APRIL_GEMM_DEBUG=$d TB_TRACE=1 timeout 100 tools/tile_bench_trace 50 4 4 8 2>&1 | grep -E "tile mt|trace" | sed "s/^/debug$d /"; done

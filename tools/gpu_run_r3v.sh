#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for args in "4 4 8" "4 2 8" "4 4 1" "4 2 2"; do TB_TRACE=1 timeout 100 tools/tile_bench_trace 50 $args 2>&1 | grep -E "tile mt|trace"; done > gpurun_out/r3v_trace.txt 2>&1
for args in "4 4 8" "4 2 8" "4 4 1" "4 2 2" "2 4 1" "2 2 2" "0 4 1" "0 2 1"; do timeout 100 tools/tile_bench 200 $args 2>&1 | grep -E "^[a-z]|tile mt"; done >> gpurun_out/r3v_trace.txt 2>&1
cat gpurun_out/r3v_trace.txt

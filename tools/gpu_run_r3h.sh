#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for ns in 3 4; do for sh in 2 10 11 12 13 14 15; do timeout 100 tools/tile_bench_ns$ns 200 $sh 2>&1 | grep -E "^[a-z]|mt=2 zs=8|mt=2 zs=4 -> fused|mt=4 zs=8|mt=4 zs=4 -> fused|mt=2 zs=2 " | sed "s/^/ns$ns /"; done; done > gpurun_out/r3h_stages.txt 2>&1
cat gpurun_out/r3h_stages.txt

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 300 tools/tile_bench 200 > gpurun_out/r3u_tile_bench.txt 2>&1; echo "rc=$?"; tail -1 gpurun_out/r3u_tile_bench.txt
grep -E "^[a-z]|planner|mt=4 zs=8|mt=4 zs=4 -> fused|mt=2 zs=8|mt=2 zs=4 -> fused|mt=4 zs=1 |mt=2 zs=2 " gpurun_out/r3u_tile_bench.txt
for ns in 3 4; do for sh in 4 11 13; do timeout 100 tools/tile_bench_ns$ns 200 $sh 2>&1 | grep -E "^[a-z]|mt=4 zs=8|mt=2 zs=8" | sed "s/^/ns$ns /"; done; done

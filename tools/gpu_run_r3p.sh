#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for pad in 0 32 64 0 32; do for sh in 2 4 11 13; do TB_APAD=$pad timeout 100 tools/tile_bench 200 $sh 2>&1 | grep -E "^[a-z]|mt=4 zs=8|mt=4 zs=1 |mt=2 zs=8|mt=2 zs=2 " | sed "s/^/pad$pad /"; done; done > gpurun_out/r3p_apad.txt 2>&1
cat gpurun_out/r3p_apad.txt

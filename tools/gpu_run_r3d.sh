mkdir -p gpurun_out
for d in 0 4 5; do echo "== APRIL_GEMM_DEBUG=$d"; APRIL_GEMM_DEBUG=$d timeout 200 tools/tile_bench 100 2>&1 | grep -E "^ffdn  256x1|^ffdn 2048x2|^proj  256x2|mt=4 zs=8 -> fused|mt=2 zs=8 -> fused|mt=4 zs=1 |mt=4 zs=4 -> fused" ; done > gpurun_out/r3d_ablation.txt 2>&1
cat gpurun_out/r3d_ablation.txt

# usage (GPU box): tools/ab_skew.sh   -- first-round start skew (APRIL_GEMM_SKEW, x 4096 cycles): the launch times of the large-row GEMMs
# under each value (tools/gemm_bench), then the 2048-session bench under the best few
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
out=gpurun_out/skew_ab.txt; : > $out
tools/cu_pair_probe 2048 >> $out 2>&1
for sk in 0 3 5 7 9 12; do
  echo "== APRIL_GEMM_SKEW=$sk" >> $out
  # gates, GM_TILE (the engine's form from 2048 rows) and the K-split form; FFN up (K-split form)
  APRIL_GEMM_SKEW=$sk GB_TILE_OK=2 tools/gemm_bench 4608 4096 1024 1 1 100 12 >> $out 2>&1
  APRIL_GEMM_SKEW=$sk tools/gemm_bench 4608 4096 1024 1 1 100 12 >> $out 2>&1
  APRIL_GEMM_SKEW=$sk tools/gemm_bench 4608 2048 512 2 1 100 12 >> $out 2>&1
  APRIL_GEMM_SKEW=$sk GB_TILE_OK=2 tools/gemm_bench 2304 4096 1024 1 1 100 12 >> $out 2>&1
done
cat $out

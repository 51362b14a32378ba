GEMM_TRACE=1 GEMM_TRACE_CU=1 APRIL_GEMM_PERSIST=1 timeout 60 tools/gemm_bench_nb2 1024 4096 1024 2 1 96 24
for shape in "1024 4096 1024 1 1" "1024 4096 1024 2 1" "2048 4096 1024 1 1" "1024 2048 512 2 1" "2048 2048 512 2 1" "2048 512 2048 0 8" "2048 512 1024 0 8" "256 4096 1024 1 1"; do
  for p in 0 1; do echo -n "PERSIST=$p "; APRIL_GEMM_PERSIST=$p timeout 60 tools/gemm_bench_nb2 $shape 96 24; done
done

for pad in 0 100; do for shape in "512 4096 16384" "1024 4096 1024" "2048 4096 1024"; do echo -n "LDSPAD=$pad "; APRIL_GEMM_LDSPAD=$pad timeout 60 tools/gemm_bench $shape 2 1 100; done; done
for skew in 0 4; do echo -n "SKEW=$skew "; APRIL_GEMM_SKEW=$skew timeout 60 tools/gemm_bench 1024 4096 1024 2 1 100; done

for shape in "256 4096 1024 1 1" "1024 4096 1024 1 1" "1024 4096 1024 2 1" "2048 4096 1024 1 1" "1024 2048 512 2 1"; do
  for p in 0 1; do for a in 0 1; do echo -n "PRIO=$p ASM=$a "; GEMM_TRACE=1 APRIL_GEMM_PRIO=$p APRIL_GEMM_ASM=$a timeout 60 tools/gemm_bench_nb2 $shape 96 24 | tr '\n' ' ' | sed 's/trace: [0-9]* workgroups, span [0-9]* ticks; mean ticks://; s/| start after first [0-9]*//'; echo; done; done
done

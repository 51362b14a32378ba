export APRIL_LOG_LEVEL=WARNING
timeout 250 tools/pp_bench 100 large 2>&1 | cut -c1-250 | head -8 > gpurun_out/pp_bench_10.txt
PPB_TRACE=1 timeout 100 tools/pp_bench_trace 30 large 2>/dev/null | grep -E -A4 "large gates  .*x 2" | cut -c1-420 >> gpurun_out/pp_bench_10.txt
cat gpurun_out/pp_bench_10.txt

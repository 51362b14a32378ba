#!/bin/bash
# Round-2 GPU pass B: quick parity subset (fail fast), default bench line, kernel trace at 256 sessions, GEMM timings.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py -m gpu -q -x --timeout 200 -p no:cacheprovider \
  -k "not 60s and not larger and not 2048 and not torch and not churn and not above_max" > gpurun_out/b_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -15 gpurun_out/b_pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/b_bench.json 2> gpurun_out/b_bench.err; echo "bench rc=$?"; tail -c 2600 gpurun_out/b_bench.json; tail -3 gpurun_out/b_bench.err
bash tools/trace_pass.sh b_b256 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0
f=$(ls /tmp/trace/b_b256/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/gap_summary.py "$f" > gpurun_out/b_b256_gap_summary.txt; cat gpurun_out/b_b256_gap_summary.txt
head -24 gpurun_out/b_b256_kernel_stats.csv
for shape in "256 512 1024 3 8" "256 512 2048 4 8" "256 4096 1024 1 1" "256 2048 512 2 1" "512 512 2048 4 8" "1024 512 2048 4 8" "1024 4096 1024 1 1"; do
  timeout 60 tools/gemm_bench $shape 200 12
done > gpurun_out/b_gemm_bench.txt 2>&1; cat gpurun_out/b_gemm_bench.txt
echo "total $(( $(date +%s) - t0 )) s"

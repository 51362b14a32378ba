#!/bin/bash
# Round-2 GPU pass F: quick parity subset, GEMM timings with per-phase stamps, bench line, trace.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 420 python -m pytest tests/test_gpu_parity.py tests/test_gpu_f16.py -m gpu -q -x --timeout 200 -p no:cacheprovider \
  -k "not 60s and not larger and not 2048 and not torch and not churn and not abovf_max" > gpurun_out/f_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -6 gpurun_out/f_pytest.log
for shape in "256 512 1024 3 8" "256 512 2048 4 8" "256 4096 1024 1 1" "256 2048 512 2 1" "512 512 2048 4 8" "1024 512 2048 4 8" "1024 4096 1024 1 1"; do
  timeout 60 tools/gemm_bench $shape 200 12
done > gpurun_out/f_gemm_bench.txt 2>&1; cat gpurun_out/f_gemm_bench.txt
for shape in "256 512 1024 3 8" "256 512 2048 4 8" "256 4096 1024 1 1" "256 2048 512 2 1"; do
  GEMM_TRACE=1 timeout 60 tools/gemm_bench_trace $shape 50 12
done > gpurun_out/f_gemm_trace.txt 2>&1; cat gpurun_out/f_gemm_trace.txt
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/f_bench.json 2> gpurun_out/f_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/f_bench.json'))
print({k:d[k] for k in ('ms_per_step','rtf','step_latency_ms','max_sessions_per_gpu_rtf_lf_0.1_tested','rtf_by_sessions_per_gpu','host_phasf_ms_total')})
print(d['roofline']['avg_launch_us'], d['roofline']['class_ms'])
PY
bash tools/tracf_pass.sh f_b256 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0 > /dev/null
f=$(ls /tmp/trace/f_b256/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/gap_summary.py "$f" > gpurun_out/f_b256_gap_summary.txt; cat gpurun_out/f_b256_gap_summary.txt
head -12 gpurun_out/f_b256_kernel_stats.csv | cut -c1-150
echo "total $(( $(date +%s) - t0 )) s"

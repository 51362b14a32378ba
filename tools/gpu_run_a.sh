#!/bin/bash
# Round-2 GPU pass A: parity suite (+ isolating variants if it fails), default bench line, kernel trace at 256 sessions,
# standalone GEMM timings.  Run through gpurun from the repository root; everything lands in gpurun_out/.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 1100 python -m pytest tests -m gpu -q --timeout 400 -p no:cacheprovider > gpurun_out/a_pytest.log 2>&1; rc=$?
echo "pytest rc=$rc  ($(( $(date +%s) - t0 )) s)"; tail -25 gpurun_out/a_pytest.log
if [ $rc -ne 0 ]; then
  for v in "APRIL_FULLK=0" "APRIL_NO_GRAPHS=1" "APRIL_GEMM_ASM=0"; do
    env $v timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q --timeout 200 -p no:cacheprovider -k "tiny or encoder or decoder or many or odd" > gpurun_out/a_pytest_$v.log 2>&1
    echo "variant $v rc=$?"; tail -4 gpurun_out/a_pytest_$v.log
  done
fi
timeout 400 python bench.py --no-cpu-baseline > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err; echo "bench rc=$?"; tail -c 2500 gpurun_out/a_bench.json; tail -5 gpurun_out/a_bench.err
bash tools/trace_pass.sh a_b256 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0
f=$(ls /tmp/trace/a_b256/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/gap_summary.py "$f" > gpurun_out/a_b256_gap_summary.txt; cat gpurun_out/a_b256_gap_summary.txt
head -30 gpurun_out/a_b256_kernel_stats.csv
for shape in "256 512 1024 3 8" "256 512 2048 4 8" "256 512 2304 4 4" "256 512 512 5 8" "256 4096 1024 1 1" "256 2048 512 2 1" "1024 512 2048 4 8" "1024 4096 1024 1 1" "2048 512 2048 4 8"; do
  timeout 60 tools/gemm_bench $shape 200 12
done > gpurun_out/a_gemm_bench.txt 2>&1; cat gpurun_out/a_gemm_bench.txt
echo "total $(( $(date +%s) - t0 )) s"

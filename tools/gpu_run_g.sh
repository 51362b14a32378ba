#!/bin/bash
# Round-2 GPU pass G: full parity suite (fail fast), bench line (sweep + 60 s one-feed case), trace.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 700 python -m pytest tests -m gpu -q -x --timeout 300 -p no:cacheprovider --durations=8 > gpurun_out/g_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -30 gpurun_out/g_pytest.log
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/g_bench.json 2> gpurun_out/g_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/g_bench.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/g_bench.json'))
for k in ('ms_per_step','rtf','step_latency_ms','max_sessions_per_gpu_rtf_le_0.1_tested','rtf_by_sessions_per_gpu','host_phase_ms_total','offline_single_session_60s','flights','replay_mismatch'):
    print(k, d.get(k))
print(d['roofline']['avg_launch_us'], d['roofline']['class_ms'])
PY
bash tools/trace_pass.sh g_b256 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0 > /dev/null
f=$(ls /tmp/trace/g_b256/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/gap_summary.py "$f" > gpurun_out/g_b256_gap_summary.txt; cat gpurun_out/g_b256_gap_summary.txt
head -8 gpurun_out/g_b256_kernel_stats.csv | cut -c1-150
echo "total $(( $(date +%s) - t0 )) s"

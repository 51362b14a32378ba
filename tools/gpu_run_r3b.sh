#!/bin/bash
# Round-3 GPU pass B: the full -m gpu suite, the default bench line (GM_TILE on), the same bench with the round-2 schedules
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider --durations=8 > gpurun_out/r3b_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -14 gpurun_out/r3b_pytest.log
timeout 600 python bench.py > gpurun_out/r3b_bench.json 2> gpurun_out/r3b_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3b_bench.err
APRIL_GM_TILE=0 timeout 300 python bench.py --no-cpu-baseline --no-config5 > gpurun_out/r3b_bench_notile.json 2> gpurun_out/r3b_bench_notile.err; echo "bench(no tile) rc=$?"
python - <<'PY'
import json
for f in ('gpurun_out/r3b_bench.json', 'gpurun_out/r3b_bench_notile.json'):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, 'unreadable', e); continue
    print(f)
    for k in ('ms_per_step', 'rtf', 'steady', 'max_sessions_per_gpu_rtf_le_0.1_tested', 'rtf_by_sessions_per_gpu', 'host_phase_ms_total', 'offline_single_session_60s', 'replay_mismatch', 'config5_f16'):
        print('  ', k, d.get(k))
    if d.get('roofline'):
        print('  ', d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['class_ms'])
PY
echo "total $(( $(date +%s) - t0 )) s"

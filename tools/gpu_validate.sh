#!/bin/bash
# GPU-box validation pass (run through gpurun): GM_TILE bitwise check + timing (tools/tile_bench), the -m gpu suite, the default bench line
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
t0=$(date +%s)
timeout 300 tools/tile_bench 200 > gpurun_out/validate_tile_bench.txt 2>&1; echo "tile_bench rc=$?"; grep -E "^[a-z]|planner" gpurun_out/validate_tile_bench.txt
timeout 900 python -m pytest tests -m gpu -q -x --timeout 400 -p no:cacheprovider > gpurun_out/validate_pytest.log 2>&1
echo "pytest rc=$?  ($(( $(date +%s) - t0 )) s)"; tail -3 gpurun_out/validate_pytest.log
timeout 600 python bench.py > gpurun_out/validate_bench.json 2> gpurun_out/validate_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/validate_bench.json'))
for k in ('ms_per_step', 'steady', 'rtf_by_sessions_per_gpu', 'max_sessions_per_gpu_rtf_le_0.1_tested', 'replay_mismatch'):
    print(k, d.get(k))
c = d['config5_f16']; print('config5', c['f16']['ms_per_step'], c['f32']['ms_per_step'], c['f16_speedup_vs_f32'])
print(d['roofline']['avg_launch_us'], d['roofline']['frac'], d['roofline']['class_ms'])
PY
echo "total $(( $(date +%s) - t0 )) s"

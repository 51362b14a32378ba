cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
timeout 600 python -m pytest tests/test_gpu_pipeline.py -q -x --timeout 500 -p no:cacheprovider > gpurun_out/r4b_pipe_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/r4b_pipe_pytest.log
timeout 300 python bench.py --no-sweep --no-cpu-baseline --no-config5 --steady-steps 60 > gpurun_out/r4b_bench.json 2> gpurun_out/r4b_bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r4b_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4b_bench.json'))
print(d['ms_per_step'], d['steady']['ms_per_step'], d['other_ingest'], d['host_phase_ms_total'], d['replay_mismatch'], d['step_latency_ms']['series'][:20])
PY

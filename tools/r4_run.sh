cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
tag=${1:-r4c}
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_parity.py -q -x --timeout 500 -p no:cacheprovider > gpurun_out/${tag}_pytest.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/${tag}_pytest.log
for sp in 2 1 0; do
APRIL_SPLIT_STREAMS=$sp timeout 300 python bench.py --no-sweep --no-cpu-baseline --no-config5 --steady-steps 60 > gpurun_out/${tag}_bench_sp$sp.json 2> gpurun_out/${tag}_bench_sp$sp.err; echo "bench split=$sp rc=$?"; tail -2 gpurun_out/${tag}_bench_sp$sp.err
python - <<PY
import json
d=json.load(open('gpurun_out/${tag}_bench_sp$sp.json'))
print($sp, d['ms_per_step'], d['steady']['ms_per_step'], d['other_ingest']['ms_per_step'], d['host_phase_ms_total'], d['replay_mismatch'], d['step_latency_ms']['series'][4:16])
PY
done

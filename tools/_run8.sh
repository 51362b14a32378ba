export APRIL_LOG_LEVEL=WARNING
timeout 250 tools/pp_bench 100 large 2>&1 | cut -c1-250 > gpurun_out/pp_bench_8.txt
PPB_TRACE=1 timeout 100 tools/pp_bench_trace 30 large 2>/dev/null | grep -E -A4 "large gates  .*x 2" | cut -c1-420 >> gpurun_out/pp_bench_8.txt
bash tools/trace_pass.sh r06a_config5 --config5-only --profile-steps 0 >/dev/null 2>&1
python - <<'PY' >> gpurun_out/pp_bench_8.txt
import csv
rows=list(csv.DictReader(open('gpurun_out/r06a_config5_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:28]:
    print("%-100s calls %5s total %8.2f ms avg %8.2f us  %5.2f%%" % (r['Name'][:100], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3, float(r['Percentage'])))
PY
cat gpurun_out/r06a_config5_bench.json | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('f16 ms/step', d['f16']['ms_per_step'], 'f32', d['f32']['ms_per_step'], d['f16']['gates_gemm'])" >> gpurun_out/pp_bench_8.txt
cat gpurun_out/pp_bench_8.txt

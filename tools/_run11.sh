export APRIL_LOG_LEVEL=WARNING
timeout 250 tools/pp_bench 100 both 2>&1 | cut -c1-250 > gpurun_out/pp_bench_11.txt
for envs in "APRIL_PP_SPLIT=0" "APRIL_PP_SPLIT=1"; do
  env $envs timeout 400 python bench.py --config5-only --profile-steps 4 > gpurun_out/c5_$envs.json 2> gpurun_out/c5_$envs.err || tail -3 gpurun_out/c5_$envs.err
  python - "$envs" gpurun_out/c5_$envs.json <<'PY' >> gpurun_out/pp_bench_11.txt
import json, sys
d=json.load(open(sys.argv[2]))
g=d['f16'].get('gates_gemm') or {}
print("%-30s f16 %.3f f32 %.3f | f16 gates %.1f us (%.3f of peak) mism %d | class_ms %s" % (sys.argv[1], d['f16']['ms_per_step'], d['f32']['ms_per_step'], g.get('avg_launch_us', 0), g.get('frac_of_mfma_peak', 0), d['f16']['replay_mismatch'], g.get('class_ms')))
PY
done
timeout 900 python -m pytest tests/test_gpu_f16.py tests/test_gpu_gemm_pp.py -x -q 2>&1 | tail -4 >> gpurun_out/pp_bench_11.txt
cat gpurun_out/pp_bench_11.txt

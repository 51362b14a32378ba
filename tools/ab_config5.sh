# usage: tools/ab_config5.sh tag "ENV.." ...  -- the configs[4] leg (512 sessions, larger encoder, fp16 + fp32) under each environment
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
tag=$1; shift
for envs in "$@"; do
  env $envs timeout 400 python bench.py --config5-only --profile-steps 4 > gpurun_out/${tag}_c5.json 2> gpurun_out/${tag}_c5.err || tail -3 gpurun_out/${tag}_c5.err
  python - "$envs" gpurun_out/${tag}_c5.json <<'PY'
import json, sys
d=json.load(open(sys.argv[2]))
g=d['f16'].get('gates_gemm') or {}
print("%-50s f16 %.3f f32 %.3f | f16 gates %.1f us (%.3f of peak) mism %d" % (sys.argv[1], d['f16']['ms_per_step'], d.get('f32', {}).get('ms_per_step', 0.0), g.get('avg_launch_us', 0), g.get('frac_of_mfma_peak', 0), d['f16']['replay_mismatch']))
PY
done

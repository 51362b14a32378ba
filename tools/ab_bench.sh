# usage: tools/ab_bench.sh tag "ENV1=.. ENV2=.." "ENV.." ...   -- the pipelined 256-session bench under each environment, alternating twice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=WARNING
tag=$1; shift
for rep in 1 2; do
for envs in "$@"; do
  env $envs timeout 300 python bench.py --no-sweep --no-cpu-baseline --no-config5 --steady-steps 100 --profile-steps 0 > gpurun_out/${tag}_ab.json 2> gpurun_out/${tag}_ab.err || tail -3 gpurun_out/${tag}_ab.err
  python - "$envs" gpurun_out/${tag}_ab.json <<'PY'
import json, sys
d=json.load(open(sys.argv[2]))
print("%-40s pipelined %.3f steady %.3f p50 %.3f | lockstep %.3f | mism %d" % (sys.argv[1], d['ms_per_step'], d['steady']['ms_per_step'], d['steady']['p50'], d['other_ingest']['ms_per_step'], d['replay_mismatch']))
PY
done
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
nproc
for ht in default 3 16 32; do
  if [ $ht = default ]; then unset APRIL_HOST_THREADS; else export APRIL_HOST_THREADS=$ht; fi
  timeout 300 python bench.py --sessions 2048 --steps 30 --warmup 8 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0 > gpurun_out/r3g_b2048_ht$ht.json 2>> gpurun_out/r3g.err; echo "rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3g_b2048_ht*.json')):
    d = json.load(open(f))
    n = d['engine_steps']
    print(f, 'ms/step', d['ms_per_step'], 'engine steps', n, 'host ms per step', [round(x / n, 3) for x in d['host_phase_ms_total']], 'p50', d['step_latency_ms']['p50'])
PY

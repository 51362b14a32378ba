#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
timeout 300 tools/tile_bench 200 > gpurun_out/r3f_tile_bench.txt 2>&1; echo "tile_bench rc=$?"; grep -E "^[a-z]|planner" gpurun_out/r3f_tile_bench.txt
for mode in 1 0 1 0; do
  APRIL_GM_TILE=$mode timeout 300 python bench.py --no-cpu-baseline --no-config5 --steady-steps 100 > gpurun_out/r3f_bench_tile${mode}_$RANDOM.json 2>> gpurun_out/r3f_bench.err; echo "bench tile=$mode rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3f_bench_tile*.json')):
    d = json.load(open(f))
    print(f, d['ms_per_step'], d['steady']['ms_per_step'], d['rtf_by_sessions_per_gpu'], d['roofline']['class_ms'])
PY

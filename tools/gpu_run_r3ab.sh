#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
for cfg in "APRIL_FF1_TILE_ROWS=256 APRIL_TILE_F16_MT=4" "APRIL_FF1_TILE_ROWS=256 APRIL_TILE_F16_MT=2"; do
  env $cfg bash tools/trace_pass.sh y --steps 10 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0 > /dev/null
  echo "== $cfg"; python -c "
import json; d=json.load(open('gpurun_out/y_bench.json')); print('ms_per_step', d['ms_per_step'])"
  grep -E "tile_zkernel<., 2, 0" gpurun_out/y_kernel_stats.csv | cut -c1-175
done

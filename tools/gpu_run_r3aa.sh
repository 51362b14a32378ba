#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
for cfg in "A=0" "APRIL_FF1_TILE_ROWS=256" "APRIL_GATES_TILE_ROWS=256" "APRIL_TILE_SPLIT_TILES=64"; do
  export $cfg
  bash tools/trace_pass.sh x_$cfg --steps 10 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0 > /dev/null
  echo "== $cfg"; python -c "
import json; d=json.load(open('gpurun_out/x_${cfg}_bench.json')); print('ms_per_step', d['ms_per_step'])"
  head -8 gpurun_out/x_${cfg}_kernel_stats.csv | cut -c1-175
  unset ${cfg%%=*}
done

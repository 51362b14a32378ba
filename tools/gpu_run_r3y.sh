#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-config5 --steady-steps 0 > gpurun_out/r3y_tmp.json 2>> gpurun_out/r3y.err; python -c "
import json; d=json.load(open('gpurun_out/r3y_tmp.json'))
print('$*', d['ms_per_step'], d['rtf_by_sessions_per_gpu'], d['roofline']['avg_launch_us'], d['replay_mismatch'])"; }
run A=default
run APRIL_GATES_TILE_ROWS=1024
run APRIL_GATES_TILE_ROWS=512
run APRIL_TILE_SPLIT_TILES=128
run APRIL_FF1_TILE_ROWS=2048
run APRIL_FF1_TILE_ROWS=1024
run A=default

#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
run() { env "$@" timeout 300 python bench.py --no-cpu-baseline --no-config5 --steady-steps 0 > gpurun_out/r3s_tmp.json 2>> gpurun_out/r3s.err; python -c "
import json; d=json.load(open('gpurun_out/r3s_tmp.json'))
print('$*', d['ms_per_step'], d['rtf_by_sessions_per_gpu'], d['roofline']['avg_launch_us'], d['roofline']['frac'], d['replay_mismatch'])"; }
run APRIL_GATES_TILE=1 APRIL_TILE_BIG_F32=1
run APRIL_GATES_TILE=1 APRIL_TILE_BIG_F32=0
run APRIL_GATES_TILE=0
run APRIL_GATES_TILE=1 APRIL_TILE_BIG_F32=1
APRIL_GATES_TILE=1 APRIL_TILE_BIG_F32=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_gates_tile.py -q -x --timeout 400 -p no:cacheprovider -k "invariant or many_sessions or config3 or gates" 2>&1 | tail -3

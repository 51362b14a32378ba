#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
run() { env "$@" python bench.py --config5-only > gpurun_out/r3m_tmp.json 2> gpurun_out/r3m.err; python -c "
import json; d=json.load(open('gpurun_out/r3m_tmp.json'))
print('$*', 'f16', d['f16']['ms_per_step'], d['f16']['gates_gemm']['avg_launch_us'], d['f16']['gates_gemm']['class_ms'], 'f32', d['f32']['ms_per_step'], 'x', d['f16_speedup_vs_f32'])"; }
run A=1
run APRIL_TILE_F16_MT=2
run APRIL_TILE_BIG_MIN_N=4000
run APRIL_TILE_BIG_MIN_N=4000 APRIL_TILE_F16_MT=2

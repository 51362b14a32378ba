#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
timeout 600 python -m pytest tests/test_gpu_f16.py -q -x --timeout 400 -p no:cacheprovider 2>&1 | tail -3
run() { env "$@" python bench.py --config5-only > gpurun_out/r3ac_tmp.json 2> gpurun_out/r3ac.err; python -c "
import json; d=json.load(open('gpurun_out/r3ac_tmp.json'))
print('$*', 'f16', d['f16']['ms_per_step'], d['f16']['gates_gemm']['class_ms'], 'f32', d['f32']['ms_per_step'], 'x', d['f16_speedup_vs_f32'])"; }
run APRIL_TILE_WIDE=1
run APRIL_TILE_WIDE=0
run APRIL_TILE_WIDE=1
run APRIL_TILE_WIDE=0

#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): per-kernel stats at B=256 and B=1, PMC passes at B=256,
# the default bench line.  Outputs land in gpurun_out/ and are copied into profiles/ by hand.
tag=${1:-r01}
bash tools/trace_pass.sh ${tag}_b256 --steps 10 --warmup 3 --no-sweep --no-cpu-baseline --profile-steps 0
bash tools/trace_pass.sh ${tag}_b1 --sessions 1 --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --profile-steps 0
bash tools/pmc_pass.sh ${tag}_b256 --steps 4 --warmup 2 --no-sweep --no-cpu-baseline --profile-steps 1
timeout 600 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
tail -c 3000 gpurun_out/${tag}_bench_default.json
ls -la gpurun_out | head -30

#!/bin/bash
# Round-end measurement set (run on the GPU box through gpurun): per-kernel stats + inter-kernel gap summaries at 256, 1 and
# 2048 sessions (graph replay: --profile-steps 0 keeps the captured graphs in use), the configs[4] leg (fp16 tile path vs fp32),
# PMC passes at 256 sessions (separate passes, kernel-trace only), the default bench line.  Outputs land in gpurun_out/ and are
# copied into profiles/ by hand.  (rocprofv3's kernel trace serialises the queues, so the traced runs use the lock-step ingest: one
# stream's worth of kernels per feed; the overlap of the three streams is shown by APRIL_STREAM_TRACE instead.)
tag=${1:-r06}
cd $GRAFT_REPO_ROOT
export APRIL_LOG_LEVEL=${APRIL_LOG_LEVEL:-WARNING}
bash tools/trace_pass.sh ${tag}_b256 --ingest lockstep --steps 10 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0
bash tools/trace_pass.sh ${tag}_b1 --ingest lockstep --sessions 1 --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0
bash tools/trace_pass.sh ${tag}_b2048 --ingest lockstep --sessions 2048 --steps 8 --warmup 4 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 0
bash tools/trace_pass.sh ${tag}_config5 --config5-only --profile-steps 0
# the default (pipelined) invocation under the kernel trace: the per-kernel durations bench.py's gates clock is checked against (same run's bench line kept)
bash tools/trace_pass.sh ${tag}_b256_pipelined --steps 20 --warmup 5 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0
for b in b256 b1 b2048 config5; do
  f=$(ls /tmp/trace/${tag}_$b/*kernel_trace.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python tools/gap_summary.py "$f" > gpurun_out/${tag}_${b}_gap_summary.txt
done
bash tools/pmc_pass.sh ${tag}_b256 --ingest lockstep --steps 4 --warmup 2 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 --profile-steps 1
# the three streams of the default (pipelined) invocation: hipEvent stamps around the front end / layer / search parts of every split feed
APRIL_STREAM_TRACE=/tmp/${tag}_stream.txt timeout 200 python bench.py --steps 30 --warmup 6 --no-sweep --no-cpu-baseline --no-config5 --steady-steps 0 > /dev/null 2>&1
[ -f /tmp/${tag}_stream.txt ] && cp /tmp/${tag}_stream.txt gpurun_out/${tag}_stream_overlap_trace.txt
timeout 120 tools/kw_bench 200 > gpurun_out/${tag}_kw_bench.txt 2>&1
# configs[4]: PMC passes of the fp16 leg alone; GM_PP against GM_TILE (bitwise + launch times), cold weights, the phase trace of the ping-pong tiles
APRIL_BENCH_C5_PRECS=f16 bash tools/pmc_pass.sh ${tag}_config5 --config5-only --profile-steps 1
( timeout 300 tools/pp_bench 200 both; timeout 200 tools/pp_bench 200 cold ) > gpurun_out/${tag}_pp_bench.txt 2>&1
PPB_TRACE=1 timeout 300 tools/pp_bench_trace 50 large > gpurun_out/${tag}_pp_phase_trace.txt 2>&1
timeout 120 tools/mfma_peak > gpurun_out/${tag}_mfma_peak.txt 2>&1
timeout 120 tools/vmem_mfma_probe > gpurun_out/${tag}_vmem_mfma_probe.txt 2>&1
timeout 600 python bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
tail -c 2500 gpurun_out/${tag}_bench_default.json
cat gpurun_out/${tag}_b256_gap_summary.txt gpurun_out/${tag}_b2048_gap_summary.txt gpurun_out/${tag}_config5_gap_summary.txt
head -14 gpurun_out/${tag}_b256_kernel_stats.csv | cut -c1-150
head -10 gpurun_out/${tag}_b2048_kernel_stats.csv | cut -c1-150
grep -E "gemm_f32_zkernel(_walk)?<4, 4, 1" gpurun_out/pmc_${tag}_b256_summary.txt | cut -c1-400
ls -la gpurun_out | wc -l

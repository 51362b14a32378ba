#!/bin/bash
# builds tools/pp_bench (GM_PP vs GM_TILE: bitwise comparison + launch times) against the product objects, from the repository root
set -e
cd "$(dirname "$0")/.."
C=april_asr_amd/csrc
make -C $C -j8 >/dev/null
HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O2 -std=c++17 -I$C -c tools/pp_bench.hip -o /tmp/pp_bench.o
$HIPCC --offload-arch=gfx950 /tmp/pp_bench.o $C/build/kernels_gemm.o $C/build/kernels_gemm_tile.o $C/build/kernels_gemm_pp.o $C/build/kernels_gemm_pw.o $C/build/kernels_gemm_kw.o $C/build/kernels_recur.o $C/build/kernels_misc.o -o tools/pp_bench
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DAPRIL_GEMM_TRACE -I$C -c $C/kernels_gemm_pp.hip -o /tmp/pp_trace.o
$HIPCC --offload-arch=gfx950 /tmp/pp_bench.o $C/build/kernels_gemm.o $C/build/kernels_gemm_tile.o /tmp/pp_trace.o $C/build/kernels_gemm_pw.o $C/build/kernels_gemm_kw.o $C/build/kernels_recur.o $C/build/kernels_misc.o -o tools/pp_bench_trace
echo built tools/pp_bench tools/pp_bench_trace

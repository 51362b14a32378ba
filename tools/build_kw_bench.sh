#!/bin/bash
# builds tools/kw_bench (product objects) and tools/kw_bench_trace (kernels_gemm_kw.hip with -DAPRIL_GEMM_TRACE: s_memtime stamps + the
# measurement-only debug modes) from the repository root
set -e
cd "$(dirname "$0")/.."
C=april_asr_amd/csrc
make -C $C -j8 >/dev/null
HIPCC=/opt/rocm/bin/hipcc
$HIPCC --offload-arch=gfx950 -O2 -std=c++17 -I$C -c tools/kw_bench.hip -o /tmp/kw_bench.o
OBJS="$C/build/kernels_gemm.o $C/build/kernels_gemm_tile.o $C/build/kernels_gemm_pp.o $C/build/kernels_gemm_pw.o $C/build/kernels_recur.o $C/build/kernels_misc.o"
$HIPCC --offload-arch=gfx950 /tmp/kw_bench.o $OBJS $C/build/kernels_gemm_kw.o -o tools/kw_bench
$HIPCC -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -DAPRIL_GEMM_TRACE -I$C -c $C/kernels_gemm_kw.hip -o /tmp/kw_trace.o
$HIPCC --offload-arch=gfx950 /tmp/kw_bench.o $OBJS /tmp/kw_trace.o -o tools/kw_bench_trace
echo built

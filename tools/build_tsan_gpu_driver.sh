#!/bin/bash
# builds tools/tsan_gpu_driver (and tools/asan_gpu_driver, the same under -fsanitize=address,undefined): the scenario driver of tests/sched_harness (driver.cc) linked against the REAL engine -- host code
# (april_api.cc, engine.cc, session.cc, loader) compiled with -fsanitize=thread, the device objects of the product build -- for
# tests/test_gpu_tsan_engine.py (runs on the GPU box with tools/tsan_gpu.supp suppressing the uninstrumented HIP / HSA runtimes)
set -e
cd "$(dirname "$0")/.."
C=april_asr_amd/csrc
make -C $C -j8 >/dev/null
HIPCC=/opt/rocm/bin/hipcc
O=${TMPDIR:-/tmp}/tsan_objs
mkdir -p $O
for f in april_api engine session model_loader onnx_reader fbank_tables; do
  $HIPCC -std=c++17 -fPIC -ffp-contract=off -O1 -g -fno-omit-frame-pointer -fsanitize=thread -Wno-unused-result -c $C/$f.cc -o $O/$f.o
done
$HIPCC -std=c++17 -O1 -g -fsanitize=thread -I$C -c tests/sched_harness/driver.cc -o $O/driver.o
$HIPCC --offload-arch=gfx950 -fsanitize=thread $O/april_api.o $O/engine.o $O/session.o $O/model_loader.o $O/onnx_reader.o $O/fbank_tables.o $O/driver.o \
  $C/build/kernels_gemm.o $C/build/kernels_gemm_tile.o $C/build/kernels_gemm_pp.o $C/build/kernels_gemm_pw.o $C/build/kernels_gemm_kw.o $C/build/kernels_recur.o $C/build/kernels_misc.o $C/build/kernels_fbank.o \
  -L/opt/rocm/lib -lrccl -lpthread -o tools/tsan_gpu_driver 2>&1 | grep -v "not currently supported" || true
# the same under AddressSanitizer + UBSan
A=${TMPDIR:-/tmp}/asan_objs
mkdir -p $A
for f in april_api engine session model_loader onnx_reader fbank_tables; do
  $HIPCC -std=c++17 -fPIC -ffp-contract=off -O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined -Wno-unused-result -c $C/$f.cc -o $A/$f.o
done
$HIPCC -std=c++17 -O1 -g -fsanitize=address,undefined -I$C -c tests/sched_harness/driver.cc -o $A/driver.o
$HIPCC --offload-arch=gfx950 -fsanitize=address,undefined $A/april_api.o $A/engine.o $A/session.o $A/model_loader.o $A/onnx_reader.o $A/fbank_tables.o $A/driver.o \
  $C/build/kernels_gemm.o $C/build/kernels_gemm_tile.o $C/build/kernels_gemm_pp.o $C/build/kernels_gemm_pw.o $C/build/kernels_gemm_kw.o $C/build/kernels_recur.o $C/build/kernels_misc.o $C/build/kernels_fbank.o \
  -L/opt/rocm/lib -lrccl -lpthread -o tools/asan_gpu_driver 2>&1 | grep -v "not currently supported" || true
test -x tools/tsan_gpu_driver && test -x tools/asan_gpu_driver && echo built

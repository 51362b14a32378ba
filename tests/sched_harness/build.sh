#!/bin/bash
# build.sh <sanitizer flags> <output>: the real host runtime + the fake engine + the driver, host-only (no device code is compiled)
set -e
cd "$(dirname "$0")/../.."
CXX=${CXX:-/opt/rocm/lib/llvm/bin/clang++}
C=april_asr_amd/csrc
OUT=$2
mkdir -p "$(dirname "$OUT")"
$CXX -std=c++17 -O1 -g -fno-omit-frame-pointer $1 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -I$C -Wno-unused-result \
    $C/session.cc $C/april_api.cc $C/model_loader.cc $C/onnx_reader.cc $C/fbank_tables.cc tests/sched_harness/fake_engine.cc tests/sched_harness/driver.cc \
    -lpthread -o "$OUT"

#!/bin/bash
# Builds the library with the K7 matrix-mode phase clock compiled in (its own .so, not the product's)
# -> tools/diag/_build/libecoflap_hip_k7clk.so; then on the GPU: python tools/diag/k7_clock.py
set -e
cd "$(dirname "$0")/../../ecoflap_amd/csrc"
make -s build/zo_perturb.hip.o build/reduce.hip.o build/sparsegpt.hip.o build/syrk.hip.o build/global_prune.hip.o \
     build/shape_ops.hip.o build/attention.hip.o build/gemm_f32.hip.o build/allocator.cpp.o build/api_misc.cpp.o
mkdir -p ../../tools/diag/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -std=c++17 -DECO_K7_CLOCK -w -c wanda.hip \
    -o ../../tools/diag/_build/wanda_clk.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../tools/diag/_build/libecoflap_hip_k7clk.so \
    build/zo_perturb.hip.o build/reduce.hip.o ../../tools/diag/_build/wanda_clk.o build/sparsegpt.hip.o build/syrk.hip.o \
    build/global_prune.hip.o build/shape_ops.hip.o build/attention.hip.o build/gemm_f32.hip.o build/allocator.cpp.o \
    build/api_misc.cpp.o
echo built tools/diag/_build/libecoflap_hip_k7clk.so

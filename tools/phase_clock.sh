#!/bin/bash
# Diagnostic build of the library with per-phase wave-cycle counters in the two gather kernels
# (csrc/common.hpp: DGNN_PHASE_CLOCK) -> tools/bin/libdgnn_hip_clk.so; run on the GPU box with
#   DGNN_HIP_LIB=$PWD/tools/bin/libdgnn_hip_clk.so python tools/phase_clock.py 8x8 1x16 4x64
set -e
cd "$(dirname "$0")/../dgnn_amd/csrc"
mkdir -p ../../tools/bin
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable \
  -ffp-contract=fast -fno-fast-math -DDGNN_PHASE_CLOCK -c gather.hip -o /tmp/gather_clk.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-variable \
  -ffp-contract=fast -fno-fast-math -DDGNN_PHASE_CLOCK -c gather_bwd.hip -o /tmp/gather_bwd_clk.o
make ops.o convert.o knn.o layer.o >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/bin/libdgnn_hip_clk.so /tmp/gather_clk.o /tmp/gather_bwd_clk.o ops.o convert.o knn.o layer.o
echo built tools/bin/libdgnn_hip_clk.so

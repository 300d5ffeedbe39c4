#!/bin/bash
# Builds the library and the C++ probes (phase stamps: tools/knn_probe) from the repo root.
set -e
cd "$(dirname "$0")/.."
make -C flux3d.jl_amd/csrc 2>&1 | grep -E "error|Error" -A5 || true
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form -I include -I flux3d.jl_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS -DFX3D_PROBE tools/knn_probe.hip flux3d.jl_amd/csrc/runtime.hip -o tools/knn_probe 2>&1 | grep -E "error" -A3 || true
ls -la flux3d.jl_amd/lib/libflux3d_hip.so tools/knn_probe

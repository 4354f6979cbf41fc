#!/bin/bash
# build an A/B variant of libbhgeo.so: scripts/build_variant.sh <name> [extra hipcc flags]
name=$1; shift
mkdir -p build/variants
cd blackhole_geodesic_calculator_amd/csrc
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-math-errno -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -disable-machine-licm -Wno-unused-function "$@" -shared -o ../../build/variants/libbhgeo_$name.so geodesic_kernels.hip frame_kernels.hip bhgeo_capi.hip

#!/bin/bash
# build an A/B variant of libbhgeo.so: scripts/build_variant.sh <name> [extra hipcc flags]
name=$1; shift
mkdir -p build/variants
cd blackhole_geodesic_calculator_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-math-errno -mllvm -amdgpu-atomic-optimizer-strategy=None -Wno-unused-function"
T=$(mktemp -d)
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c geodesic_kernels.hip -o $T/a.o &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c geodesic_kernels_kerr.hip -o $T/b.o &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c frame_kernels.hip -o $T/c.o &
/opt/rocm/bin/hipcc $F -mllvm -disable-machine-licm "$@" -c geodesic_kernels_timelike.hip -o $T/f.o &
/opt/rocm/bin/hipcc $F "$@" -c probe_kernels.hip -o $T/g.o &
/opt/rocm/bin/hipcc $F "$@" -c bhgeo_capi.hip -o $T/d.o &
/opt/rocm/bin/hipcc $F "$@" -c bhgeo_frame.hip -o $T/e.o &
wait
/opt/rocm/bin/hipcc $F -shared -o ../../build/variants/libbhgeo_$name.so $T/a.o $T/b.o $T/c.o $T/d.o $T/e.o $T/f.o $T/g.o -ldl
rm -rf $T

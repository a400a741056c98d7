#!/bin/bash
# Developer: build the current working tree's fp64 kernels into gpsat_amd/csrc/libgpsat_hip_<TAG>.so (the other objects
# are taken as they are), for A/B runs in ONE gpurun call:  scripts/build_variant_f64.sh TAG [extra hipcc flags]
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT/gpsat_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=on --offload-arch=gfx950 -I../../include -I. -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c gpsat_kernels_f64.hip -o /tmp/v_${TAG}_f64.o &
/opt/rocm/bin/hipcc $F "$@" -DGPSAT_F64_W4 -c gpsat_kernels_f64.hip -o /tmp/v_${TAG}_f64w4.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgpsat_hip_${TAG}.so gpsat_kernels.o gpsat_kernels_w8.o /tmp/v_${TAG}_f64.o /tmp/v_${TAG}_f64w4.o gpsat_select.o gpsat_post.o gpsat_capi.o
ls -la libgpsat_hip_${TAG}.so

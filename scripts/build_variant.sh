#!/bin/bash
# Developer: build the current working tree's fp32 kernels into gpsat_amd/csrc/libgpsat_hip_<TAG>.so (the other objects
# are taken as they are), for A/B runs in ONE gpurun call (boxes differ by a few per cent):
#   scripts/build_variant.sh TAG [--patch scripts/experiments/X.patch] [extra hipcc flags]
# --patch applies a developer patch (scripts/experiments/) to the kernel sources for this build only and reverts it.
set -e
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
PATCH=""
if [ "$1" = "--patch" ]; then PATCH="$(realpath "$2")"; shift 2; fi
cd "$ROOT"
if [ -n "$PATCH" ]; then patch -p1 < "$PATCH"; trap 'cd "$ROOT" && patch -R -p1 < "$PATCH"' EXIT; fi
cd "$ROOT/gpsat_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=on --offload-arch=gfx950 -I../../include -I. -Wno-unused-function"
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -c gpsat_kernels.hip -o /tmp/v_${TAG}_k.o &
/opt/rocm/bin/hipcc $F -fno-slp-vectorize "$@" -DGPSAT_W8 -c gpsat_kernels.hip -o /tmp/v_${TAG}_k8.o &
/opt/rocm/bin/hipcc $F "$@" -x hip -c gpsat_capi.cpp -o /tmp/v_${TAG}_capi.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libgpsat_hip_${TAG}.so /tmp/v_${TAG}_k.o /tmp/v_${TAG}_k8.o gpsat_kernels_f64.o gpsat_kernels_f64_w4.o gpsat_select.o gpsat_post.o /tmp/v_${TAG}_capi.o
ls -la libgpsat_hip_${TAG}.so

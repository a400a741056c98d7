#!/bin/bash
# Developer (E48): one GPU-box call -- hazard microbenchmark, then factor-dump comparisons of three diagnostic builds.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
export TMPDIR=/tmp
L=$PWD/gpsat_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 scripts/bench_valu_mfma_hazard.hip -o /tmp/bench_vmh || exit 1
timeout -k 10 240 /tmp/bench_vmh 20000 > gpurun_out/e48/vmh.txt 2>&1 || { echo "vmh failed/timeout"; tail -3 gpurun_out/e48/vmh.txt; exit 1; }
echo "vmh done"; tail -4 gpurun_out/e48/vmh.txt
GPSAT_LIB=$L/libgpsat_hip_dirty.so timeout -k 10 400 python3 scripts/e48_dump_compare.py 5 4096 500 > gpurun_out/e48/dump_dirty.txt 2>&1 || { echo "dirty failed"; tail -5 gpurun_out/e48/dump_dirty.txt; exit 1; }
echo "dirty done"; tail -12 gpurun_out/e48/dump_dirty.txt
GPSAT_LIB=$L/libgpsat_hip_pad.so E48_SHOW=30 timeout -k 10 500 python3 scripts/e48_dump_compare.py 100 4096 500 > gpurun_out/e48/dump_pad.txt 2>&1 || { echo "pad failed"; tail -5 gpurun_out/e48/dump_pad.txt; exit 1; }
echo "pad done"; grep -v "differs 0, whose objective differs 0" gpurun_out/e48/dump_pad.txt | tail -30
GPSAT_LIB=$L/libgpsat_hip_shipdump.so timeout -k 10 300 python3 scripts/e48_dump_compare.py 30 4096 500 > gpurun_out/e48/dump_ship.txt 2>&1 || { echo "ship failed"; tail -5 gpurun_out/e48/dump_ship.txt; exit 1; }
echo "ship done"; grep -v "differs 0, whose objective differs 0" gpurun_out/e48/dump_ship.txt | tail -8

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
GPSAT_LIB=$L/libgpsat_hip_dirty.so E48_SHOW=16 timeout -k 10 400 python3 scripts/e48_dump_compare.py 3 4096 500 > gpurun_out/e48/dump_dirty2.txt 2>&1 || { echo "dirty failed"; tail -5 gpurun_out/e48/dump_dirty2.txt; exit 1; }
grep -v "^  tile.*only z" gpurun_out/e48/dump_dirty2.txt | head -120

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
GPSAT_LIB=$L/libgpsat_hip_dirty.so E48_SAVE=40 E48_SHOW=0 timeout -k 10 300 python3 scripts/e48_dump_compare.py 4 4096 500 > gpurun_out/e48/dump_dirty_ev.txt 2>&1 || { echo "failed"; tail -5 gpurun_out/e48/dump_dirty_ev.txt; exit 1; }
grep "^launch" gpurun_out/e48/dump_dirty_ev.txt; ls gpurun_out/e48/events | wc -l; du -sh gpurun_out/e48/events

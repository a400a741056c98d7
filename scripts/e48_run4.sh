#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
for v in dirtysc dirty dirtyplain dirtysleep; do
  SC=""; [ $v = dirtysc ] && SC=1
  E48_SELFCHECK=$SC GPSAT_LIB=$L/libgpsat_hip_$v.so E48_SHOW=2 timeout -k 10 300 python3 scripts/e48_dump_compare.py 12 4096 500 > gpurun_out/e48/dump_$v.txt 2>&1 || { echo "$v failed"; tail -5 gpurun_out/e48/dump_$v.txt; exit 1; }
  echo "== $v"; grep "^launch\|^lib\|by lane\|packed" gpurun_out/e48/dump_$v.txt | head -60
done

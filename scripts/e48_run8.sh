#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
for v in $VARIANTS; do
  GPSAT_LIB=$L/libgpsat_hip_$v.so E48_SHOW=0 E48_SAVE=${SAVE:-0} timeout -k 10 500 python3 scripts/e48_dump_compare.py ${LAUNCHES:-12} 4096 500 > gpurun_out/e48/dump_$v.txt 2>&1 || { echo "$v failed"; tail -5 gpurun_out/e48/dump_$v.txt; exit 1; }
  echo "== $v: $(grep '^launch' gpurun_out/e48/dump_$v.txt | awk '{print $7}' | tr -d ',' | tr '\n' ' ')"
done

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/bench_chain_beside_bf16.hip -o /tmp/bench_cb || exit 1
timeout -k 10 300 /tmp/bench_cb 400 > gpurun_out/e48/chain_beside.txt 2>&1 || { echo "bench_cb failed"; tail -3 gpurun_out/e48/chain_beside.txt; exit 1; }
cat gpurun_out/e48/chain_beside.txt
for v in dirtynoprio dirtyscalar dirty; do
  GPSAT_LIB=$L/libgpsat_hip_$v.so E48_SHOW=2 timeout -k 10 300 python3 scripts/e48_dump_compare.py 12 4096 500 > gpurun_out/e48/dump_$v.txt 2>&1 || { echo "$v failed"; tail -5 gpurun_out/e48/dump_$v.txt; exit 1; }
  echo "== $v"; grep "^launch\|^lib" gpurun_out/e48/dump_$v.txt | awk '{print $1,$2,$8}' | tr '\n' ';'; echo
done

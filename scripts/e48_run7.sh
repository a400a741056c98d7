#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/bench_pkfma_waw.hip -o /tmp/bench_waw || exit 1
timeout -k 10 400 /tmp/bench_waw 100000 > gpurun_out/e48/pkfma_waw.txt 2>&1 || { echo "bench failed"; tail -3 gpurun_out/e48/pkfma_waw.txt; exit 1; }
cat gpurun_out/e48/pkfma_waw.txt

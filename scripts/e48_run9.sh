#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
BENCH_ARGS="--no-quality" bash scripts/ab_bench.sh base noslp cand > gpurun_out/e48/ab_cand.txt 2>&1; cat gpurun_out/e48/ab_cand.txt
GPSAT_LIB=$L/libgpsat_hip_cand.so timeout -k 10 600 python3 scripts/nll_values.py 150 > gpurun_out/e48/soak_cand.txt 2>&1; tail -4 gpurun_out/e48/soak_cand.txt
GPSAT_LIB=$L/libgpsat_hip_dirtynoslp.so timeout -k 10 300 python3 scripts/nll_values.py 60 > gpurun_out/e48/soak_dirtynoslp.txt 2>&1; tail -4 gpurun_out/e48/soak_dirtynoslp.txt

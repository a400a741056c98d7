#!/bin/bash
# Developer: parity tests + A/B bench of library variants in one GPU-box call:  VARIANTS="a b" [TESTS="tests/x.py ..."] scripts/ab_run.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
L=$PWD/gpsat_amd/csrc
for v in $VARIANTS; do
  if [ -n "$TESTS" ]; then
    GPSAT_LIB=$L/libgpsat_hip_$v.so python -m pytest $TESTS -m gpu -q -x > gpurun_out/ab/test_$v.txt 2>&1; echo "== tests $v: $(tail -1 gpurun_out/ab/test_$v.txt)"
    grep -m5 "^FAILED\|^E  " gpurun_out/ab/test_$v.txt
  fi
done
BENCH_ARGS="--no-quality $BENCH_EXTRA" bash scripts/ab_bench.sh base $VARIANTS 2>&1 | tee gpurun_out/ab/bench_${TAG:-ab}.txt
if [ -n "$CONFIGS2" ]; then BENCH_ARGS="--no-quality --workload configs2 $BENCH_EXTRA" bash scripts/ab_bench.sh base $VARIANTS 2>&1 | tee gpurun_out/ab/bench2_${TAG:-ab}.txt; fi

#!/bin/bash
# Developer: A/B of library variants on the bench workload inside one GPU-box call: scripts/ab_bench.sh TAG... ("base" = libgpsat_hip.so)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for rep in 1 2; do
  for t in "$@"; do
    lib=$ROOT/gpsat_amd/csrc/libgpsat_hip_$t.so; [ "$t" = base ] && lib=$ROOT/gpsat_amd/csrc/libgpsat_hip.so
    GPSAT_LIB=$lib python $ROOT/bench.py --steps 5 --warmup 1 --cpu-tiles 0 --no-host-leg --no-other-workloads $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', d['value'], d['config']['evals_per_tile'], d['roofline']['frac'], d['roofline']['kernel_ms'])"
  done
done

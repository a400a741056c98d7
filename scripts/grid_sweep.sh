#!/bin/bash
# Developer: per-workgroup tile rate against the number of resident workgroups (GPSAT_DEBUG_GRID) -- the working set of
# factors is grid x tile bytes: 512 MB at 512 workgroups of configs[1] (HBM), 32 MB at 32 (L2-resident)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for wl in configs1 f64fit; do
  for g in 512 256 128 64 32; do
    GPSAT_DEVELOPER=1 GPSAT_DEBUG_GRID=$g python $ROOT/bench.py --workload $wl --tiles $((g * 8)) --steps 3 --warmup 1 --cpu-tiles 0 --no-host-leg --no-quality --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl grid $g', 'tiles/s', d['value'], 'per-wg', round(d['value']/$g, 2), 'frac', d['roofline']['frac'], 'ms', d['roofline']['kernel_ms'])"
  done
done

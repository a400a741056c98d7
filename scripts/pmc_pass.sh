#!/bin/bash
# One extra PMC pass over the default bench workload: scripts/pmc_pass.sh TAG COUNTER...
set -e
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/raw -- python3 $ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-other-workloads $BENCH_ARGS --steps 1 --warmup 0 > $OUT/log.txt 2>&1
cd $ROOT
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
agg = {}
for f in glob.glob(os.path.join(sys.argv[1], "raw", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "gp_tile_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] = agg.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    print(f"{k}: {v:.6g}")
PY
rm -rf $OUT/raw

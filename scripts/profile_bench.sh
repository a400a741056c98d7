#!/bin/bash
# Profile the default bench.py workload on the GPU box: kernel trace + stats, then PMC passes in SEPARATE runs
# (never combined with other trace domains).  Output under gpurun_out/$1; summaries by scripts/collect_profile.py.
set -e
TAG=${1:-prof}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT/summary
cd /tmp && export TMPDIR=/tmp
B="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg $BENCH_ARGS"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $B --steps 3 --warmup 1 > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
echo stats done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -- python3 $B --steps 1 --warmup 0 > $OUT/fetch.log 2>&1
echo fetch done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -- python3 $B --steps 1 --warmup 0 > $OUT/write.log 2>&1
echo write done
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_F32 GRBM_GUI_ACTIVE -d $OUT/mfma -- python3 $B --steps 1 --warmup 0 > $OUT/mfma.log 2>&1
echo mfma done
cd $ROOT
python3 scripts/collect_profile.py $OUT/summary stats=$OUT/stats fetch=$OUT/fetch write=$OUT/write mfma=$OUT/mfma | tee $OUT/summary/derived.txt
cp $OUT/bench_under_rocprof.json $OUT/summary/
# keep only the summaries (the raw traces are large)
rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma

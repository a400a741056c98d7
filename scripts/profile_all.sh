#!/bin/bash
# Profile every shape / kernel of the path on the GPU box.  For each workload: one `rocprofv3 --kernel-trace --stats` run
# and SEPARATE `--pmc` passes (never combined with other trace domains); raw traces are summarised by
# scripts/collect_profile.py into gpurun_out/<TAG>/<workload>/ and deleted.  Copy what is to be judged into profiles/.
#   scripts/profile_all.sh TAG [workload ...]      workloads: configs1 configs2 configs2_1024 configs4 f64fit select post
set -e
TAG=${1:-prof}; shift || true
WLS=${@:-configs1 configs2 configs4 f64fit select post}
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
for WL in $WLS; do
  OUT=$ROOT/gpurun_out/$TAG/$WL; mkdir -p $OUT
  MFMA=SQ_INSTS_VALU_MFMA_F32; PAT=gp_tile_kernel; ST="--steps 3 --warmup 1"; ONE="--steps 1 --warmup 0"
  case $WL in
    configs1) CMD="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-quality --no-other-workloads";;
    configs2) CMD="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-quality --workload configs2";;
    configs2_1024) CMD="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-quality --workload configs2 --tiles 1024";;
    configs4) CMD="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-quality --workload configs4"; MFMA=SQ_INSTS_VALU_MFMA_F64;;
    f64fit)   CMD="$ROOT/bench.py --cpu-tiles 0 --workers 1 --no-host-leg --no-quality --workload f64fit"; MFMA=SQ_INSTS_VALU_MFMA_F64;;
    select)   CMD="$ROOT/scripts/select_bench.py"; PAT=select; ST=""; ONE="";;
    post)     CMD="$ROOT/scripts/post_bench.py"; PAT="smooth|glue"; ST=""; ONE="";;
  esac
  echo "== $WL"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $CMD $ST > $OUT/run_under_rocprof.txt 2> $OUT/stats.log
  P() { n=$1; shift; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv --pmc "$@" -d $OUT/$n -- python3 $CMD $ONE > $OUT/$n.log 2>&1 || echo "pass $n failed"; }
  P fetch FETCH_SIZE
  P write WRITE_SIZE
  if [ $WL != select ] && [ $WL != post ]; then
    P mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES $MFMA GRBM_GUI_ACTIVE
    P sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES
    P sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS_F32
  else
    P sq1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
  fi
  (cd $ROOT && python3 scripts/collect_profile.py $OUT "$PAT" stats=$OUT/stats fetch=$OUT/fetch write=$OUT/write mfma=$OUT/mfma sq1=$OUT/sq1 sq2=$OUT/sq2 > $OUT/derived.txt; cat $OUT/derived.txt)
  rm -rf $OUT/stats $OUT/fetch $OUT/write $OUT/mfma $OUT/sq1 $OUT/sq2 $OUT/*.log
done

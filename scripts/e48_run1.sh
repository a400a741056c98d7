#!/bin/bash
# Developer (EXPERIMENTS.md E48): the GPU-box calls of the root-causing, for the record.  Build the diagnostic variants first:
#   scripts/build_variant.sh slp -fslp-vectorize -DGPSAT_DUMP        (the SLP vectoriser back on: 2-5 events per launch)
# then e.g.   gpurun -- 'bash scripts/e48_run1.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/e48
L=$PWD/gpsat_amd/csrc
for b in bench_valu_mfma_hazard bench_sc1_handoff bench_chain_beside_bf16 bench_pkfma_waw; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -Wno-unused-result scripts/$b.hip -o /tmp/$b || exit 1
  timeout -k 10 400 /tmp/$b > gpurun_out/e48/$b.txt 2>&1 || { echo "$b failed"; exit 1; }
  tail -3 gpurun_out/e48/$b.txt
done
for v in ${VARIANTS:-slp}; do
  [ -f $L/libgpsat_hip_$v.so ] || continue
  GPSAT_LIB=$L/libgpsat_hip_$v.so E48_SHOW=8 E48_SAVE=${SAVE:-0} timeout -k 10 500 python3 scripts/e48_dump_compare.py ${LAUNCHES:-12} 4096 500 > gpurun_out/e48/dump_$v.txt 2>&1 || { echo "$v failed"; exit 1; }
  echo "== $v: $(grep '^launch' gpurun_out/e48/dump_$v.txt | awk '{print $7}' | tr -d ',' | tr '\n' ' ')"
done

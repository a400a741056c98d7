#!/bin/bash
# Developer: device assembly of gpsat_kernels.hip and a table of its MFMA-bearing basic blocks (kernel <3,0> by default).
# usage: scripts/asm_blocks.sh [extra hipcc flags]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p /tmp/dis
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I$ROOT/include -I$ROOT/gpsat_amd/csrc --cuda-device-only -S $ROOT/gpsat_amd/csrc/gpsat_kernels.hip -o /tmp/dis/k2.s -Rpass-analysis=kernel-resource-usage "$@" 2> /tmp/dis/res2.txt
grep -A9 "${KSEL:-ILi3ELi0}" /tmp/dis/res2.txt | grep "VGPRs\|Scratch\|Spill" | sed 's/.*remark: *//'
awk -v pat="^_ZN5gpsat2(w4|w8|v2)14gp_tile_kernel${KSEL:-ILi3ELi0}" '$0 ~ pat {f=1} f{print} /\.end_amdhsa_kernel/{if(f){exit}}' /tmp/dis/k2.s > /tmp/dis/k30b.s
python3 - <<'PY'
import re
lines=open('/tmp/dis/k30b.s').read().split('\n')
blocks=[]; cur=None
tot=dict(mfma=0,valu=0,mov=0,lane=0,scratch=0)
for i,l in enumerate(lines):
    m=re.match(r'^(\.LBB\d+_\d+):',l)
    if m:
        cur=[m.group(1),i,0,0,0,0,0,0]; blocks.append(cur)
    elif cur is not None:
        s=l.strip()
        if s.startswith('v_mfma'): cur[2]+=1; tot['mfma']+=1
        elif s.startswith('v_mov'): cur[7]+=1; tot['mov']+=1
        elif s.startswith('v_'): cur[3]+=1; tot['valu']+=1
        if s.startswith('global_load'): cur[4]+=1
        if 'v_readlane' in s or 'v_writelane' in s: cur[5]+=1; tot['lane']+=1
        if s.startswith('scratch_'): cur[6]+=1; tot['scratch']+=1
print("static totals", tot, "lines", len(lines))
print("label line mfma valu gload laneops scratch vmov")
for b in blocks:
    if b[2]>=32: print(b)
PY

#!/usr/bin/env python3
"""Developer (EXPERIMENTS.md E48): static scan of a disassembled gfx950 kernel (llvm-objdump -d) for the distances, in wait
states, between (a) a VALU instruction that writes a VGPR and an MFMA that reads it as A / B / C, and (b) an MFMA that writes
VGPRs and the first non-MFMA instruction that reads or overwrites one of them.  Straight-line distances only (a label or a
branch resets the window), s_nop N counts N + 1.

    scripts/asm_hazard_scan.py dis.s _ZN5gpsat2w414gp_tile_kernelILi3ELi0EEEvNS_10KernelArgsE
"""
import re
import sys
from collections import Counter

path, sym = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.endswith(f"<{sym}>:"))
end = next((i for i in range(start + 1, len(lines)) if re.match(r"^[0-9a-f]+ <", lines[i])), len(lines))
body = lines[start + 1:end]


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"^(v|a)\[(\d+):(\d+)\]$", tok)
    if m:
        return [(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)]
    m = re.match(r"^(v|a)(\d+)$", tok)
    if m:
        return [(m.group(1), int(m.group(2)))]
    return []


ins = []
for l in body:
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*?)\s*//", l)
    if not m:
        if re.match(r"^[0-9a-f]+ <", l):
            ins.append(("label", []))
        continue
    op, rest = m.group(1), m.group(2)
    ops = [t for t in re.split(r",\s*(?![^\[]*\])", rest) if t]
    ins.append((op, ops))

last_w = {}          # reg -> (position in wait states, op)
mf_w = {}            # reg -> (position, op) written by MFMA
pos = 0
valu_to_mfma = Counter()
mfma_to_other = Counter()
examples = {}
for op, ops in ins:
    if op == "label" or op.startswith("s_cbranch") or op == "s_branch" or op.startswith("s_setpc") or op.startswith("s_swappc"):
        last_w.clear(); mf_w.clear(); pos += 64
        continue
    if op == "s_nop":
        pos += int(ops[0]) + 1 if ops else 1
        continue
    is_mfma = op.startswith("v_mfma")
    is_valu = op.startswith("v_") and not is_mfma
    dst = regs(ops[0]) if ops and (op.startswith("v_") or op.startswith("ds_read") or "load" in op) else []
    srcs = [r for t in ops[1:] for r in regs(t)]
    if op.startswith("v_") and not dst:
        srcs = [r for t in ops for r in regs(t)]
    if op.startswith("buffer_store") or op.startswith("global_store") or op.startswith("ds_write") or op.startswith("scratch_store") or op.startswith("v_cmp") or op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
        srcs = [r for t in ops for r in regs(t)]
        dst = []
    if is_mfma:
        names = ["A", "B", "C"]
        for k, t in enumerate(ops[1:4]):
            for r in regs(t):
                if r in last_w:
                    d = pos - last_w[r][0] - 1
                    if d <= 8:
                        key = (last_w[r][1].split("_e")[0], op, names[k], d)
                        valu_to_mfma[key] += 1
        for r in dst:
            mf_w[r] = (pos, op)
            last_w.pop(r, None)
    else:
        for r in srcs + dst:
            if r in mf_w:
                d = pos - mf_w[r][0] - 1
                key = (mf_w[r][1], "read" if r in srcs else "overwrite", op.split("_e")[0] if is_valu else op, d)
                if d <= 24:
                    mfma_to_other[key] += 1
                mf_w.pop(r, None)
        if is_valu:
            for r in dst:
                last_w[r] = (pos, op)
        else:
            for r in dst:
                last_w.pop(r, None)
    pos += 1

print("VALU write -> MFMA read (writer, mfma, operand, wait states between): count")
for k, v in sorted(valu_to_mfma.items(), key=lambda kv: (kv[0][1], kv[0][3])):
    print("   ", k, v)
print("MFMA write -> first other access (mfma, kind, instruction, wait states between): count [<= 24 only]")
agg = Counter()
for (m, kind, o, d), v in mfma_to_other.items():
    agg[(m, kind, d)] += v
for k, v in sorted(agg.items()):
    print("   ", k, v)

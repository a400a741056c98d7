#!/usr/bin/env python3
"""Developer (EXPERIMENTS.md E48): offline analysis of the event dumps saved by e48_dump_compare.py (E48_SAVE=n).

For every event (a tile whose forward-solve sums differ between two launches while its factor is bit-identical) the per-lane
difference of the chain's partial sum tp (stage 0 / stage 1) is explained, if possible, by ONE term A[q][lane] * z_k[rho(q, h)]
of the sum: lost (H1), taken with another z (H2), or taken with another A (H3)."""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rho(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


def blk_regs(d, NB, r, c):
    """block (r, c) as [reg q][lane l]"""
    b = d[(r * NB + c) * 1024:(r * NB + c + 1) * 1024]
    out = np.empty((16, 64), np.float32)
    for q in range(16):
        out[q] = b[(q >> 2) * 256 + 4 * np.arange(64) + (q & 3)]
    return out


for f in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", "e48", "events", "*.npz")))[:int(sys.argv[1]) if len(sys.argv) > 1 else 100]:
    E = np.load(f)
    a, b, NB = E["a"], E["b"], int(E["NB"])
    NBLK, NPAD = NB * NB + NB, NB * 32
    DBG0 = NBLK * 1024 + 2 * NPAD + 16
    ga, gb = a[DBG0:DBG0 + 8192].reshape(4, 32, 64), b[DBG0:DBG0 + 8192].reshape(4, 32, 64)
    za, zb = a[NBLK * 1024:NBLK * 1024 + NPAD], b[NBLK * 1024:NBLK * 1024 + NPAD]
    gm = ga.view(np.int32) != gb.view(np.int32)
    gm[:, NB:, :] = False
    rows_bad = np.nonzero(gm.any(axis=(0, 2)))[0]
    if not len(rows_bad):
        print(os.path.basename(f), "no staged difference"); continue
    jr = int(rows_bad.min())
    st = int(np.nonzero(gm[:, jr].any(axis=1))[0].min())
    lanes = np.nonzero(gm[st, jr])[0]
    delta = (gb[st, jr].astype(np.float64) - ga[st, jr].astype(np.float64))
    j0 = jr & ~1
    print(f"{os.path.basename(f)}: first row {jr} (j0 {j0}), first stage {st}, lanes {lanes.min()}..{lanes.max()} ({len(lanes)}), |delta| max {np.abs(delta).max():.3e},"
          f" z equal up to row {jr}: {np.array_equal(za[:32 * jr], zb[:32 * jr])}")
    L = np.arange(48, 64)
    dl = delta[L]
    best = []
    # the k-steps of the stage: stage 0 = chain_kloop(0, j0 - 2): operands U_{k,jr} from memory; stage 1 adds rows j0-2, j0-1 (held path, registers)
    krange = range(0, max(0, j0 - 2)) if st == 0 else range(max(0, j0 - 2), j0)
    for k in krange:
        A = blk_regs(a, NB, k, jr).astype(np.float64)          # [q][lane]
        Aprev = blk_regs(a, NB, k - 1, jr).astype(np.float64) if k > 0 else None
        for q in range(16):
            zq = float(za[32 * k + rho(q, 1)])
            term = A[q, L] * zq
            for sign in (+1, -1):
                res = np.linalg.norm(dl - sign * term) / (np.linalg.norm(dl) + 1e-300)
                best.append((res, "H1 lost/duplicated term", k, q, sign, None))
            # H2: another z
            den = float(A[q, L] @ A[q, L])
            if den > 0:
                cst = float(dl @ A[q, L]) / den
                res = np.linalg.norm(dl - cst * A[q, L]) / (np.linalg.norm(dl) + 1e-300)
                best.append((res, "H2 other z", k, q, 0, zq + cst))
            # H3: another A: A' = A + delta / z
            if abs(zq) > 0:
                Ap = A[q, L] + dl / zq
                best.append((9.0, "H3 other A (candidate values)", k, q, 0, Ap))
    best.sort(key=lambda t: t[0])
    for res, what, k, q, sign, extra in best[:3]:
        msg = f"    {what}: k {k} reg {q} (row {rho(q, 1)}) residual {res:.2e}"
        if what.startswith("H1"):
            msg += f" sign {sign:+d}"
        if what.startswith("H2"):
            zq = float(za[32 * k + rho(q, 1)])
            # does the implied z' equal another entry of z?
            near = np.argsort(np.abs(za.astype(np.float64) - extra))[:2]
            msg += f"  z used {zq!r} implied z' {extra!r}; nearest entries of z: {[(int(i), float(za[i])) for i in near]}"
        print(msg)

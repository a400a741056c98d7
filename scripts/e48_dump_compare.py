#!/usr/bin/env python3
"""Developer (DESIGN.md / EXPERIMENTS.md E48): WHERE does a tile's factorisation differ between two launches of the same batch?

Needs a -DGPSAT_DUMP build (GPSAT_LIB=...): every tile's factor square (U upper / M lower, NB x NB blocks in the accumulator
layout), DinvT, z, alpha and log-determinant of its one evaluation are left in a device buffer.  Launch 0 is the reference;
for every later launch the tiles whose dump differs are listed with the FIRST panel of the sweep in which a block differs,
which blocks of that panel differ, and what the difference looks like (elements, lanes, registers, size in ulps).

    GPSAT_LIB=gpsat_amd/csrc/libgpsat_hip_dirty.so python3 scripts/e48_dump_compare.py [launches] [T] [N]
"""
import ctypes as C
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gpsat_amd import synthetic as syn   # noqa: E402
from gpsat_amd.engine import Engine      # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N = int(sys.argv[3]) if len(sys.argv) > 3 else 500
MAXSHOW = int(os.environ.get("E48_SHOW", "12"))
P, D, kid = 8, 3, 0
NB = (N + 31) // 32
NPAD = NB * 32
NBLK = NB * NB + NB
DBG0 = NBLK * 1024 + 2 * NPAD + 16
STRIDE = DBG0 + 8 * 1024


def rho(r, h):
    return (r & 3) + 8 * (r >> 2) + 4 * h


# float offset inside a block -> (row, column) of the 32 x 32 block
_o = np.arange(1024)
_reg = (_o // 256) * 4 + (_o % 4)
_lane = (_o % 256) // 4
ROW = rho(_reg, _lane // 32)
COL = _lane % 32


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


def main():
    with threadpool_limits(1):
        with ThreadPoolExecutor(16) as pool:
            tiles = list(pool.map(lambda t: syn.make_tile(t, N, P, D, kid), range(T)))
    X = np.concatenate([t[0] for t in tiles]).astype(np.float32)
    y = np.concatenate([t[1] for t in tiles]).astype(np.float32)
    Xs = np.concatenate([t[2] for t in tiles]).astype(np.float32)
    th = np.exp(np.random.default_rng(5).normal(0.0, 0.5, (T, D + 2)))
    kw = dict(D=D, obs_off=np.arange(T + 1, dtype=np.int64) * N, pred_off=np.arange(T + 1, dtype=np.int64) * P,
              theta0=th, kernel="RBF", optimiser="none", want_grad=True)
    eng = Engine(0, workgroups_per_cu=int(os.environ.get("GD_WG", "0")))
    lib = eng._lib
    lib.gpsat_debug_set_dump.restype = C.c_int
    lib.gpsat_debug_set_dump.argtypes = [C.c_void_p, C.c_void_p, C.c_ulonglong]
    bufs = [torch.zeros((T, STRIDE), dtype=torch.float32, device="cuda") for _ in range(2)]
    print(f"lib {os.path.basename(os.environ.get('GPSAT_LIB', 'default'))}  T {T} N {N} NB {NB}  dump {T * STRIDE * 4 / 2**30:.2f} GiB x 2", flush=True)
    ref_nll = None
    first_panel_hist = {}
    kinds_hist = {}
    shown = 0
    nsaved = [0]
    for rep in range(REPS):
        buf = bufs[0] if rep == 0 else bufs[1]
        assert lib.gpsat_debug_set_dump(eng._h, C.c_void_p(buf.data_ptr()), STRIDE) == 0
        r = eng.fit_predict_batch(X=X, y=y, Xs=Xs, **kw)
        torch.cuda.synchronize()
        if os.environ.get("E48_SELFCHECK"):
            # -DGPSAT_SELFCHECK builds: the chain's partial sums after its first k-loop (stage 0, rows 4..15) against the same
            # sums accumulated a second time by scalar v_fma_f32 from the same registers (rows 20..31), within THIS launch
            g = buf[:, DBG0:DBG0 + 8192].view(T, 4, 32, 64).view(torch.int32)
            mm = (g[:, 0, 4:16, :] != g[:, 0, 20:32, :])
            nbad = int(mm.sum().item())
            print(f"launch {rep}: self-check: {nbad} (tile, row, lane) entries where the packed and the scalar sum differ", flush=True)
            if nbad:
                idx = mm.nonzero().cpu().numpy()
                lanes = np.bincount(idx[:, 2], minlength=64)
                rows_ = np.bincount(idx[:, 1] + 4, minlength=16)
                print("      by lane quarter:", [int(lanes[16 * q:16 * q + 16].sum()) for q in range(4)], " by row:", rows_.tolist())
                gf = buf[:, DBG0:DBG0 + 8192].view(T, 4, 32, 64)
                for (tt, rr, ll) in idx[:6]:
                    print(f"      tile {tt} row {rr + 4} lane {ll}: packed {gf[tt, 0, rr + 4, ll].item()!r} scalar {gf[tt, 0, rr + 20, ll].item()!r}")
        if rep == 0:
            ref_nll = r.nll.copy()
            continue
        # bitwise comparison on the device
        a = bufs[0].view(torch.int32)
        b = bufs[1].view(torch.int32)
        ne = (a != b)
        tile_bad = ne.any(dim=1).nonzero().flatten().cpu().numpy()
        nll_bad = np.nonzero(r.nll != ref_nll)[0]
        print(f"launch {rep}: tiles whose dump differs {len(tile_bad)}, whose objective differs {len(nll_bad)}"
              f" (dump differs but not the objective: {len(set(tile_bad) - set(nll_bad))})", flush=True)
        for t in tile_bad:
            da = bufs[0][t].cpu().numpy()
            db = bufs[1][t].cpu().numpy()
            blk_a = da[:NBLK * 1024].reshape(NBLK, 1024)
            blk_b = db[:NBLK * 1024].reshape(NBLK, 1024)
            dm = (blk_a.view(np.int32) != blk_b.view(np.int32))
            bad_blocks = np.nonzero(dm.any(axis=1))[0]
            zdiff = np.nonzero(da[NBLK * 1024:NBLK * 1024 + NPAD].view(np.int32) != db[NBLK * 1024:NBLK * 1024 + NPAD].view(np.int32))[0]
            if len(bad_blocks) == 0 and os.environ.get("E48_SAVE") and nsaved[0] < int(os.environ["E48_SAVE"]):
                # the whole dump of the tile from both launches, for offline analysis (scripts/e48_event_analysis.py)
                os.makedirs(os.path.join(ROOT, "gpurun_out", "e48", "events"), exist_ok=True)
                np.savez_compressed(os.path.join(ROOT, "gpurun_out", "e48", "events", f"ev_{rep}_{t}.npz"), a=da, b=db, NB=NB, N=N, tile=t,
                                    theta=th[t], nll_a=ref_nll[t], nll_b=r.nll[t])
                nsaved[0] += 1
            if len(bad_blocks) == 0:
                ga = da[DBG0:DBG0 + 8192].reshape(4, 32, 64)
                gb = db[DBG0:DBG0 + 8192].reshape(4, 32, 64)
                gm = ga.view(np.int32) != gb.view(np.int32)
                gm[:, NB:, :] = False
                rows_bad = np.nonzero(gm.any(axis=(0, 2)))[0]
                jr0 = int(rows_bad.min()) if len(rows_bad) else -1
                zkey = ((int(zdiff[0]) // 32) if len(zdiff) else -1, "z-only")
                kinds_hist[zkey] = kinds_hist.get(zkey, 0) + 1
                if shown < MAXSHOW:
                    shown += 1
                    za, zb = da[NBLK * 1024:NBLK * 1024 + NPAD], db[NBLK * 1024:NBLK * 1024 + NPAD]
                    if not len(zdiff):
                        print(f"  tile {t}: neither blocks nor z differ (alpha / logdet / staged sums only); rows with staged differences {rows_bad.tolist()}")
                        continue
                    zr = int(zdiff[0]) // 32
                    zu = ulps(za[32 * zr:32 * zr + 32], zb[32 * zr:32 * zr + 32])
                    print(f"  tile {t}: only z/alpha/logdet differ; nll {ref_nll[t]:.10f} -> {r.nll[t]:.10f}; first z index {zdiff[:1]} (row {zr});"
                          f" ulps of z in that row: {zu.tolist()}")
                    print(f"      staged sums: first row with a difference {jr0}; rows with differences {rows_bad.tolist()}")
                    for st in range(4):
                        for jr in rows_bad[:2]:
                            lanes = np.nonzero(gm[st, jr])[0]
                            if len(lanes):
                                u = ulps(ga[st, jr][lanes], gb[st, jr][lanes])
                                rel = np.abs(ga[st, jr][lanes] - gb[st, jr][lanes]) / (np.abs(ga[st, jr]).max() + 1e-30)
                                print(f"      stage {st} row {jr}: lanes {lanes.tolist()} ulps max {int(u.max())} rel-to-max {rel.max():.2e}"
                                      f"  e.g. lane {lanes[0]}: {ga[st, jr][lanes[0]]!r} vs {gb[st, jr][lanes[0]]!r}")
                continue
            # panel in which a block is written: its ROW index // 2 (square blocks r*NB+c; DinvT[j] at NB*NB + j)
            rows = np.where(bad_blocks < NB * NB, bad_blocks // NB, bad_blocks - NB * NB)
            cols = np.where(bad_blocks < NB * NB, bad_blocks % NB, bad_blocks - NB * NB)
            isd = bad_blocks >= NB * NB
            pan = rows // 2
            p0 = pan.min()
            first_panel_hist[p0] = first_panel_hist.get(p0, 0) + 1
            sel = np.nonzero(pan == p0)[0]
            desc = []
            for i in sel:
                bi = bad_blocks[i]
                r_, c_ = int(rows[i]), int(cols[i])
                kind = "DinvT" if isd[i] else ("diag" if r_ == c_ else ("U" if r_ < c_ else "M"))
                m = dm[bi]
                u = ulps(blk_a[bi][m], blk_b[bi][m])
                rr, cc = ROW[m], COL[m]
                desc.append((kind, r_, c_, int(m.sum()), int(u.max()), sorted(set(rr.tolist()))[:6], sorted(set(cc.tolist()))[:6],
                             float(np.abs(blk_a[bi][m] - blk_b[bi][m]).max()), float(np.abs(blk_a[bi]).max())))
            # the row (j0 or j1) and kinds that differ in the first panel
            key = tuple(sorted(set((d[0], d[1] - 2 * p0) for d in desc)))
            kinds_hist[key] = kinds_hist.get(key, 0) + 1
            if shown < MAXSHOW:
                shown += 1
                print(f"  tile {t}: nll {ref_nll[t]:.10f} -> {r.nll[t]:.10f}; {len(bad_blocks)} blocks differ; first panel {p0}"
                      f" (rows {2 * p0},{2 * p0 + 1}); z differs from index {zdiff[:1]}")
                for d in desc[:10]:
                    print(f"      {d[0]:5s} block ({d[1]},{d[2]}): {d[3]} elements differ, max {d[4]} ulps, |diff| max {d[7]:.3e} of |block| max {d[8]:.3e};"
                          f" rows {d[5]}{'...' if d[3] > 6 else ''} cols {d[6]}")
    eng.close()
    print("first differing panel -> tiles:", dict(sorted(first_panel_hist.items())))
    print("what differs in the first panel (kind, row within panel) -> tiles:")
    for k, v in sorted(kinds_hist.items(), key=lambda kv: -kv[1])[:20]:
        print("   ", v, k)


if __name__ == "__main__":
    main()

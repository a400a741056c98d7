"""Timing of the post-processing kernels on the GPU box (no oracle): smoothing of T expert locations and gluing of R
overlapping prediction rows; prints kernel ms from gpsat_last_timing and the NumPy host equivalent beside it."""
import os
import sys
import time
import ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gpsat_amd.engine import default_engine

eng = default_engine()


def kernel_ms():
    k, t = C.c_double(), C.c_double()
    eng._lib.gpsat_last_timing(eng._h, C.byref(k), C.byref(t))
    return k.value


rng = np.random.default_rng(0)
for T in (4096, 32768, 131072):
    x, y, v = rng.uniform(-3e6, 3e6, T), rng.uniform(-3e6, 3e6, T), rng.uniform(0, 1, T)
    eng.smooth_batch(x, y, v, 2e5, 2e5)
    t0 = time.perf_counter()
    out = eng.smooth_batch(x, y, v, 2e5, 2e5)
    wall = time.perf_counter() - t0
    km = kernel_ms()
    n = min(T, 512)
    t0 = time.perf_counter()
    d2 = ((x[None, :] - x[:n, None]) / 2e5) ** 2 + ((y[None, :] - y[:n, None]) / 2e5) ** 2
    w = np.exp(-d2 / 2)
    ref = (w * v).sum(1) / w.sum(1)
    host = (time.perf_counter() - t0) * T / n
    print(f"smooth T={T}: kernel {km:.2f} ms ({T * T / km / 1e6:.1f} G pairs/s), wall {wall * 1e3:.1f} ms, "
          f"numpy host (extrapolated) {host * 1e3:.0f} ms, max rel err {np.max(np.abs(out[:n] - ref) / np.abs(ref)):.2e}")
for L in (100_000, 2_000_000):
    k = rng.integers(1, 8, L)
    rep = np.repeat(np.arange(L), k)
    R = len(rep)
    seg = np.zeros(L + 1, np.int64)
    np.cumsum(k, out=seg[1:])
    pred = rng.uniform(0, 4000, (2, L))[:, rep]
    xprt = pred + rng.uniform(-300, 300, (2, R))
    vals = rng.standard_normal((3, R))
    eng.glue_batch(seg, pred, xprt, vals, 100.0)
    t0 = time.perf_counter()
    eng.glue_batch(seg, pred, xprt, vals, 100.0)
    wall = time.perf_counter() - t0
    km = kernel_ms()
    byts = (2 * 2 + 3) * 8 * R + 3 * 8 * L + 8 * L
    print(f"glue R={R} G={L}: kernel {km:.3f} ms ({byts / km / 1e6:.0f} GB/s), wall incl. PCIe {wall * 1e3:.1f} ms")

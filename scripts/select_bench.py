"""Developer: throughput of the batched tile selection (T experts x M rows), device vs host (KD-tree + compares)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, pandas as pd
from gpsat_amd.engine import Engine
from gpsat_amd.local_experts import DeviceSelector, LocalSelector
rng = np.random.default_rng(0)
M, T = int(os.environ.get("M", 1_000_000)), int(os.environ.get("T", 4096))
df = pd.DataFrame({"x": rng.uniform(-3e6, 3e6, M), "y": rng.uniform(-3e6, 3e6, M), "t": rng.integers(0, 30, M).astype(float)})
xl = pd.DataFrame({"x": rng.uniform(-2.5e6, 2.5e6, T), "y": rng.uniform(-2.5e6, 2.5e6, T), "t": rng.integers(4, 26, T).astype(float)})
ls = [{"col": "t", "comp": "<=", "val": 4}, {"col": "t", "comp": ">=", "val": -4}, {"col": ["x", "y"], "comp": "<", "val": 3e5}]
eng = Engine(0)
ds = DeviceSelector(df, ls, eng)
ds.select(xl.iloc[:8])
t0 = time.perf_counter(); off, idx = ds.select(xl); dt = time.perf_counter() - t0
import ctypes as C
km, tm = C.c_double(), C.c_double(); eng._lib.gpsat_last_timing(eng._h, C.byref(km), C.byref(tm))
print(f"device: count+fill kernels {km.value:.1f} ms, with copies {tm.value:.1f} ms")
print(f"device: T={T} M={M}: {dt*1e3:.1f} ms wall ({T/dt:.0f} tiles/s), mean N/tile {off[-1]/T:.0f}, {T*M/dt/1e9:.1f} G predicate-rows/s")
# the same table ordered by day (how GPSat's tables come): sub-chunks outside the +-4 day window are skipped by their boxes
dfs = df.sort_values("t", kind="stable").reset_index(drop=True)
dss = DeviceSelector(dfs, ls, eng)
dss.select(xl.iloc[:8])
t0 = time.perf_counter(); off2, idx2 = dss.select(xl); dt2 = time.perf_counter() - t0
eng._lib.gpsat_last_timing(eng._h, C.byref(km), C.byref(tm))
print(f"device, rows ordered by day: kernels {km.value:.1f} ms, {dt2*1e3:.1f} ms wall ({T/dt2:.0f} tiles/s), same selection size: {off2[-1] == off[-1]}")
# GPSat's usual sweep: all expert locations of one run share the day (a grid at t = const)
xl1 = xl.assign(t=15.0)
dss.select(xl1.iloc[:8])
t0 = time.perf_counter(); off3, idx3 = dss.select(xl1); dt3 = time.perf_counter() - t0
eng._lib.gpsat_last_timing(eng._h, C.byref(km), C.byref(tm))
print(f"device, rows ordered by day, experts of one day: kernels {km.value:.1f} ms, {dt3*1e3:.1f} ms wall ({T/dt3:.0f} tiles/s), mean N/tile {off3[-1]/T:.0f}")
ds.select(xl1.iloc[:8])
t0 = time.perf_counter(); off4, idx4 = ds.select(xl1); dt4 = time.perf_counter() - t0
eng._lib.gpsat_last_timing(eng._h, C.byref(km), C.byref(tm))
print(f"device, rows in random order, experts of one day: kernels {km.value:.1f} ms, {dt4*1e3:.1f} ms wall")
if os.environ.get("BIG"):
    # 100 000 experts x 10 M rows (was ~15 s without binning)
    Mb, Tb = 10_000_000, 100_000
    dfb = pd.DataFrame({"x": rng.uniform(-3e6, 3e6, Mb), "y": rng.uniform(-3e6, 3e6, Mb), "t": rng.integers(0, 30, Mb).astype(float)})
    xlb = pd.DataFrame({"x": rng.uniform(-2.5e6, 2.5e6, Tb), "y": rng.uniform(-2.5e6, 2.5e6, Tb), "t": rng.integers(4, 26, Tb).astype(float)})
    lsb = [{"col": "t", "comp": "<=", "val": 4}, {"col": "t", "comp": ">=", "val": -4}, {"col": ["x", "y"], "comp": "<", "val": 2.2e4}]
    dsb = DeviceSelector(dfb, lsb, eng)
    dsb.select(xlb.iloc[:8])
    t0 = time.perf_counter(); offb, idxb = dsb.select(xlb); dtb = time.perf_counter() - t0
    eng._lib.gpsat_last_timing(eng._h, C.byref(km), C.byref(tm))
    print(f"device, binned: T={Tb} M={Mb}: {dtb*1e3:.0f} ms wall (device part {tm.value:.0f} ms), mean N/tile {offb[-1]/Tb:.0f}")
hs = LocalSelector(df, ls)
t0 = time.perf_counter()
for t in range(64):
    hs.mask({c: xl.iloc[t][c] for c in xl.columns})
dh = (time.perf_counter() - t0) / 64
print(f"host (reference semantics, KD-tree built once): {dh*1e3:.2f} ms/tile -> {1/dh:.0f} tiles/s")

"""Developer: the same sweep (SIDE^2 experts, as scripts/orchestrator_bench.py) run three ways -- default (two engines, calls of 1024
tiles), one engine with calls of 768 tiles, two engines with calls of 4096 -- must return identical tables, bit for bit."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DEVICE_ONLY"] = "1"
import numpy as np, pandas as pd
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orchestrator_bench.py")).read()
src = src.split("eng = Engine(0)")[0]                      # the configuration part only
exec(compile(src, "orchestrator_bench.py", "exec"))
from gpsat_amd.engine import Engine
from gpsat_amd.local_experts import BatchedLocalExpertOI
eng = Engine(0)
runs = []
for workers, chunk in ((2, None), (1, 768), (2, 4096)):
    with tempfile.TemporaryDirectory() as d:
        oi = BatchedLocalExpertOI(engine=eng, device_select=True, **cfg)
        oi.engine_workers = workers
        tabs = oi.run(store_path=os.path.join(d, "s"), engine_chunk=chunk)
        runs.append(tabs)
        print(f"engines {workers}, engine_chunk {chunk}: {len(tabs['run_details'])} experts in {oi.run_seconds:.2f} s", flush=True)
ignore = ("run_time",)
for k in runs[0]:
    for other in runs[1:]:
        a = runs[0][k].drop(columns=[c for c in ignore if c in runs[0][k].columns])
        b = other[k].drop(columns=[c for c in ignore if c in other[k].columns])
        pd.testing.assert_frame_equal(a, b, check_exact=True)
print("all tables identical:", sorted(runs[0]))

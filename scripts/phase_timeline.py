"""Developer: event timeline of one objective evaluation of workgroup 0 (diagnostic build, make -C gpsat_amd/csrc prof).

Prints, per wave, what it did when (cycles from the start of the evaluation) so that the dependent path of the
Cholesky / inverse sweep can be read off: chain of panel s, run-ahead k-loops, groups, waits."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPSAT_LIB", os.path.join(ROOT, "gpsat_amd", "csrc", "libgpsat_hip_prof.so"))
import ctypes as C
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn, _lib

T, N, D = int(os.environ.get("T", 2048)), int(os.environ.get("N", 500)), 3
b = syn.make_batch(32, N, 500, D, 0, base_seed=1)
rep = T // 32
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
obs_off = np.arange(T + 1) * N
eng = Engine(0, workgroups_per_cu=int(os.environ.get("WG", 2)))
lib = _lib.get_lib()
args = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=np.zeros(T + 1, dtype=np.int64), Xs=dXs[:0].contiguous(),
            theta0=np.tile(b["truth"], (rep, 1)), kernel="RBF", optimiser="none", want_grad=True)
eng.fit_predict_batch(**args)
eng.fit_predict_batch(**args)
buf = (C.c_ulonglong * 8192)()
lib.gpsat_debug_trace.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
lib.gpsat_debug_trace(eng._h, buf)
tr = np.array(buf[:], dtype=np.uint64).reshape(8, 1024)
NAMES = {1: "chain start", 2: "chain: columns there", 3: "chain: early k-steps done", 4: "chain: parked k-loop there", 5: "chain: rows of group 0 + last k-steps done",
         6: "chain: factor copies free", 7: "chain: factor 0 done", 8: "chain: factor 1 done", 9: "chain END (ready)",
         20: "group start", 21: "group: columns there", 22: "group: k-loop done", 23: "group END", 24: "group: factors there",
         30: "run-ahead: inputs there", 31: "run-ahead END (parked)", 40: "column wave: wait ready", 41: "column wave: ready",
         50: "pulled group of panel", 70: "K^-1 group start, a0 =", 71: "K^-1 group END, a0 =", 60: "sweep done", 61: "after barrier"}
ev = []
for w in range(8):
    for v in tr[w]:
        v = int(v)
        if v:
            ev.append((v >> 16, w, v & 255, (v >> 8) & 255))
ev.sort()
t0 = ev[0][0]
print(f"{len(ev)} events; cycles from the first")
last = {}
for t, w, code, arg in ev:
    a = f"panel {arg}" if code < 20 or code >= 30 else f"panel {arg >> 4} group {arg & 15}"
    dt = t - last.get(w, t)
    last[w] = t
    print(f"{t - t0:9d}  w{w}  (+{dt:7d})  {NAMES.get(code, code)}  {a}")

"""Developer: in-kernel cycle split of the fp64 4-wave tile kernel per code segment (diagnostic build: make -C gpsat_amd/csrc prof)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPSAT_LIB", os.path.join(ROOT, "gpsat_amd", "csrc", "libgpsat_hip_prof.so"))
import ctypes as C
import numpy as np, torch
from gpsat_amd.engine import Engine
from gpsat_amd import synthetic as syn, _lib

T, N, D = int(os.environ.get("T", 2048)), int(os.environ.get("N", 500)), 3
b = syn.make_batch(32, N, 500, D, 0, base_seed=1, dtype=np.float64)
rep = T // 32
X, y, Xs = np.tile(b["X"], (rep, 1)), np.tile(b["y"], rep), np.tile(b["Xs"], (rep, 1))
dev = torch.device("cuda", 0)
dX, dy, dXs = (torch.from_numpy(v).to(dev) for v in (X, y, Xs))
obs_off = np.arange(T + 1) * N
th0 = np.tile(b["truth"], (rep, 1))
eng = Engine(0)
lib = _lib.get_lib()
names = ["(A) diagonal-region items", "barrier after (A)", "(B) block triangle (wave 0)", "look-ahead beside (B)", "barrier after (B)",
         "(C) U column items", "(C) inverse: triangle", "(C) inverse: pairs", "look-ahead tail", "barrier after (C)",
         "gradient: k-loops", "gradient: barrier + sum", "evaluate total", "gradient: contraction", "(C) K blocks of U items", "(B) 16x16 factorisations"]
for name, kw in [("factorisation only", dict(optimiser="none")), ("objective + gradient", dict(optimiser="none", want_grad=True))]:
    args = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=np.zeros(T + 1, dtype=np.int64), Xs=dXs[:0].contiguous(),
                theta0=th0, kernel="RBF", dtype="f64")
    args.update(kw)
    eng.fit_predict_batch(**args)
    r = eng.fit_predict_batch(**args)
    buf = (C.c_ulonglong * 64)()
    lib.gpsat_debug_profile(eng._h, buf)
    prof = np.array(buf[:], dtype=np.float64).reshape(4, 16) / T
    print(f"== {name}: kernel {r.kernel_ms:.2f} ms for {T} tiles; s_memtime ticks per evaluation and wave")
    tot = prof[:, 12]
    for i, nme in enumerate(names):
        print(f"   {nme:30s} " + " ".join(f"w{w}:{prof[w, i]:9.0f} ({100 * prof[w, i] / tot[w]:4.1f}%)" for w in range(4)))

"""Developer: in-kernel cycle split per code segment using the diagnostic build (make -C gpsat_amd/csrc prof)."""
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
th0 = np.tile(b["truth"], (rep, 1))
eng = Engine(0, workgroups_per_cu=int(os.environ.get("WG", 2)))
lib = _lib.get_lib()
names = ["w0 diag k-loop", "diag_factor", "w0 serial rest", "barrier wait (PT)", "group k-loop", "group rows",
         "grad c-loop", "grad contraction", "grad final barrier", "evaluate total", "chain flag wait", "park diag partials", "early-start k-loop"]
for name, kw in [("potrf only", dict(optimiser="none")), ("potrf+trtri+grad", dict(optimiser="none", want_grad=True))]:
    args = dict(D=D, obs_off=obs_off, X=dX, y=dy, pred_off=np.zeros(T + 1, dtype=np.int64), Xs=dXs[:0].contiguous(),
                theta0=th0, kernel="RBF")
    args.update(kw)
    eng.fit_predict_batch(**args)
    r = eng.fit_predict_batch(**args)
    buf = (C.c_ulonglong * 64)()
    lib.gpsat_debug_profile(eng._h, buf)
    prof = np.array(buf[:], dtype=np.float64).reshape(4, 16)
    print(f"== {name}: kernel {r.kernel_ms:.2f} ms; cycles per evaluation per wave (s_memtime ticks, 100MHz?)")
    print(f"   workgroup 0: {prof[0, 14]:.0f} s_memtime ticks in {prof[0, 15]:.0f} ticks of the 100 MHz clock -> s_memtime runs at "
          f"{prof[0, 14] / max(prof[0, 15], 1) * 100:.0f} MHz; events say {r.kernel_ms:.3f} ms")
    per_eval = prof / T
    for i, nme in enumerate(names):
        print(f"   {nme:22s} " + " ".join(f"w{w}:{per_eval[w, i]:10.0f}" for w in range(4)))

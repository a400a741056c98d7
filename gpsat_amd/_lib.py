"""ctypes binding of libgpsat_hip.so (C ABI declared in include/gpsat_hip.h).

The product path has no CPU fallback: if the shared library is missing or a symbol is
absent, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# GPSAT_LIB: developer override to load the diagnostic build (scripts/phase_profile.py); never a fallback
LIB_PATH = os.environ.get("GPSAT_LIB") or os.path.join(_HERE, "csrc", "libgpsat_hip.so")

# constants mirrored from include/gpsat_hip.h
ABI_VERSION = 3
F32, F64 = 0, 1
KERNEL_IDS = {"RBF": 0, "SquaredExponential": 0, "Matern12": 1, "Exponential": 1, "Matern32": 2, "Matern52": 3}
OPT_NONE, OPT_LBFGS, OPT_ADAM = 0, 1, 2
OPT_IDS = {"none": OPT_NONE, None: OPT_NONE, "lbfgs": OPT_LBFGS, "L-BFGS-B": OPT_LBFGS, "adam": OPT_ADAM}
MEM_HOST, MEM_DEVICE = 0, 1
STATUS = {0: "converged", 1: "max_iter", 2: "not_pd", 3: "nan", 4: "skipped", 5: "not_optimised", 6: "ls_failed"}

EXPORTS = ["gpsat_version", "gpsat_last_error", "gpsat_device_count", "gpsat_create", "gpsat_device_name",
           "gpsat_destroy", "gpsat_fit_predict_batch", "gpsat_last_timing", "gpsat_select_batch",
           "gpsat_smooth_batch", "gpsat_glue_batch", "gpsat_max_tile_obs"]


class GpsatOpts(C.Structure):
    _fields_ = [("workgroups_per_cu", C.c_int32), ("reserved", C.c_int32 * 7)]


class GpsatBatch(C.Structure):
    _fields_ = [
        ("T", C.c_int32), ("D", C.c_int32), ("dtype", C.c_int32), ("kernel", C.c_int32),
        ("memory", C.c_int32), ("optimiser", C.c_int32), ("max_iter", C.c_int32), ("max_ls", C.c_int32),
        ("ftol", C.c_double), ("gtol", C.c_double), ("adam_lr", C.c_double),
        ("obs_off", C.c_void_p), ("pred_off", C.c_void_p), ("theta0", C.c_void_p), ("lo", C.c_void_p),
        ("hi", C.c_void_p), ("trainable", C.c_void_p),
        ("X", C.c_void_p), ("y", C.c_void_p), ("Xs", C.c_void_p),
        ("theta", C.c_void_p), ("nll", C.c_void_p), ("grad", C.c_void_p), ("status", C.c_void_p),
        ("n_eval", C.c_void_p), ("f_mean", C.c_void_p), ("f_var", C.c_void_p), ("y_var", C.c_void_p),
        ("cov_off", C.c_void_p), ("f_cov", C.c_void_p), ("n_iter", C.c_void_p),
    ]


SEL_MAXCRIT = 4
COMP_IDS = {">=": 0, ">": 1, "==": 2, "<": 3, "<=": 4}


class GpsatSelectSpec(C.Structure):
    _fields_ = [("n_crit", C.c_int32), ("kind", C.c_int32 * SEL_MAXCRIT), ("comp", C.c_int32 * SEL_MAXCRIT),
                ("ncols", C.c_int32 * SEL_MAXCRIT), ("cols", (C.c_int32 * 3) * SEL_MAXCRIT),
                ("val", C.c_double * SEL_MAXCRIT)]


class LibraryMissing(ImportError):
    pass


def _one_hip_runtime():
    """A process must use ONE HIP runtime.  PyTorch wheels bundle their own libamdhip64 (same SONAME as the system ROCm's);
    whichever is loaded first serves both.  If this library came first with the system runtime, a later `import torch` +
    CUDA initialisation finds "No HIP GPUs"; the other way round works.  So when PyTorch is installed, its runtime is
    loaded first -- without importing torch."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError as e:
            import warnings
            warnings.warn(f"gpsat_amd: could not preload PyTorch's HIP runtime {cand}: {e}; if torch is imported later in this "
                          f"process it may not see the GPU")


def _check_one_hip_runtime():
    """After libgpsat_hip.so is loaded: the de-duplication above only works when PyTorch's bundled runtime has the SONAME
    this library was linked against (same ROCm major).  Two different libamdhip64 files mapped into the process = two
    runtimes: say so instead of failing later with 'No HIP GPUs'."""
    try:
        paths = set()
        for line in open("/proc/self/maps"):
            if "libamdhip64.so" in line:
                paths.add(os.path.realpath(line.split()[-1]))
    except OSError:
        return
    if len(paths) > 1:
        import warnings
        warnings.warn("gpsat_amd: two HIP runtimes are loaded (" + ", ".join(sorted(paths)) + "): libgpsat_hip.so was built "
                      "against a ROCm whose libamdhip64 SONAME differs from the one PyTorch bundles.  Build the library with "
                      "the ROCm release PyTorch was built for (INTEGRATION.md, 'One HIP runtime per process').")


def load():
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C gpsat_amd/csrc` (there is no CPU fallback)")
    _one_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    _check_one_hip_runtime()
    for name in EXPORTS:
        if not hasattr(lib, name):
            raise LibraryMissing(f"{LIB_PATH} does not export {name}")
    lib.gpsat_version.restype = C.c_int
    lib.gpsat_last_error.restype = C.c_char_p
    lib.gpsat_device_count.restype = C.c_int
    lib.gpsat_create.restype = C.c_int
    lib.gpsat_create.argtypes = [C.c_int, C.POINTER(GpsatOpts), C.POINTER(C.c_void_p)]
    lib.gpsat_device_name.restype = C.c_int
    lib.gpsat_device_name.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
    lib.gpsat_destroy.restype = C.c_int
    lib.gpsat_destroy.argtypes = [C.c_void_p]
    lib.gpsat_fit_predict_batch.restype = C.c_int
    lib.gpsat_fit_predict_batch.argtypes = [C.c_void_p, C.POINTER(GpsatBatch)]
    lib.gpsat_select_batch.restype = C.c_int
    lib.gpsat_select_batch.argtypes = [C.c_void_p, C.POINTER(GpsatSelectSpec), C.c_int64, C.c_int32, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
    lib.gpsat_smooth_batch.restype = C.c_int
    lib.gpsat_smooth_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_void_p]
    lib.gpsat_glue_batch.restype = C.c_int
    lib.gpsat_glue_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    lib.gpsat_max_tile_obs.restype = C.c_int
    lib.gpsat_max_tile_obs.argtypes = [C.c_int, C.c_int]
    lib.gpsat_last_timing.restype = C.c_int
    lib.gpsat_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    if lib.gpsat_version() != ABI_VERSION:
        raise LibraryMissing(f"ABI version mismatch: library {lib.gpsat_version()} != binding {ABI_VERSION}")
    return lib


def max_tile_obs(dtype: str, D: int) -> int:
    """Largest tile (observations) the kernels take for this dtype / input dimension (gpsat_max_tile_obs)."""
    return int(get_lib().gpsat_max_tile_obs(F32 if dtype == "f32" else F64, int(D)))


_lib = None


def get_lib():
    global _lib
    if _lib is None:
        _lib = load()
    return _lib

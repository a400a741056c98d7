"""gpsat_amd -- MI355X-native local-expert exact-GP backend (drop-in for GPSat's model backend).

Only what the hot path needs lives here:
  csrc/       gfx950 HIP kernels + the C ABI (libgpsat_hip.so, declared in include/gpsat_hip.h)
  _lib.py     ctypes binding
  engine.py   packed-ragged batch API (fit + objective + predict for thousands of tiles per launch)
  models.py   HipGPRModel: per-tile class with the reference's BaseGPRModel interface
"""
__all__ = ["Engine", "default_engine", "HipGPRModel", "get_model"]


def __getattr__(name):
    if name in ("Engine", "default_engine", "BatchResult", "GpsatError"):
        from . import engine
        return getattr(engine, name)
    if name in ("HipGPRModel", "get_model"):
        from . import models
        return getattr(models, name)
    raise AttributeError(name)

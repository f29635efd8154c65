"""MI355X-native ray-trace hot path for par_raytracer.

The package holds only what the path needs: ``csrc/`` (HIP kernels + the C ABI of include/prt.h),
``host/`` (C++ mirror of the reference driver: OBJ loader, sphere hierarchy, Render, tone map + PNG),
``capi``/``api`` (ctypes bindings used by tests and bench.py) and ``scenes`` (synthetic OBJ generators).
Importing the package does not load the libraries; ``capi.hip_lib()`` does, and raises if they are not built.
"""
from . import scenes  # noqa: F401

__all__ = ["scenes", "capi", "api"]

"""amber_amd -- MI355X-native path-tracing integrator for amber (etheriqa/amber's `pt` hot path).

The product is the shared library ``amber_amd/lib/libamber_hip.so`` (gfx950 kernels, the C ABI of
``include/amber_hip.h`` and the C++ host object model of ``amber_amd/csrc/amber``); the tests load
``libamber_hip_lab.so``, the same sources plus the known-answer entry points (``include/amber_hip_lab.h``).  This Python
package is a thin ctypes binding used by the tests, ``bench.py`` and the multi-GPU driver; it
contains no rendering code of its own and has no CPU fallback.
"""
from . import api  # noqa: F401
from .api import (  # noqa: F401
    ENGINE_AUTO,
    ENGINE_BVH,
    ENGINE_LIST,
    ENGINE_REFERENCE_BVH,
    ENGINE_TWO_PHASE,
    ENGINE_WAVEFRONT,
    PT_FLAG_BVH_ITEMS,
    PT_FLAG_BVH_POOL,
    AmberError,
    FlatMaterial,
    FlatObject,
    FlatThinLens,
    HostScene,
    PathTracer,
    PtParams,
    Sensor,
    build_library,
    device_count,
    is_lab,
    export,
    MATH_GLIBC,
    MATH_PORTABLE,
    kat_math,
    math_mode,
    library_path,
    load_library,
    tonemap,
)

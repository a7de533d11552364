"""pathtracer_amd -- MI355X (gfx950) wavefront path tracer behind the PathTracer scene API.

Python here is a thin ctypes layer over libpathtracer_amd.so (include/pt_api.h); the product is
the HIP library.  Importing the package does not load the library; the first call does, and
raises if it has not been built -- there is no CPU fallback.
"""
from . import api, scenes  # noqa: F401
from .api import Context, PtError, SceneDesc  # noqa: F401

__all__ = ["api", "scenes", "Context", "PtError", "SceneDesc"]

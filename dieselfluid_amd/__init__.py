"""dieselfluid_amd -- MI355X (gfx950) SPH particle-step engine behind dieselfluid's
solver / compute-gpu API.

The product is the C-ABI library ``libdslsph.so`` (hand-written HIP, include/dslsph.h).
This Python package is only the ctypes binding the tests and ``bench.py`` drive it with,
plus the multi-GPU slab driver that uses ``torch.distributed`` as plumbing.  There is no
CPU fallback: importing works without a GPU, creating an engine does not.
"""
from ._lib import Params, Stats, load_library, library_path, build_library, DslError  # noqa: F401
from .engine import SPHEngine, BUF, KERNEL_IDS  # noqa: F401
from . import scenes  # noqa: F401

__all__ = ["SPHEngine", "Params", "Stats", "BUF", "KERNEL_IDS", "load_library", "library_path", "build_library",
           "DslError", "scenes"]

"""rpt_amd — MI355X-native drop-in for the path-tracing hot path of neevparikh/rpt.

The package holds only what that path needs: `csrc/` (HIP kernels + the C ABI of
include/rpt_hip.h), the host-side mirror of rpt's builder API (`api.py`) and the benchmark
scene definitions (`scenes.py`).
"""
from .api import *  # noqa: F401,F403
from .api import __all__ as _api_all
from . import scenes  # noqa: F401
from .io import load_obj, load_obj_with_mtl, load_stl  # noqa: F401

__all__ = list(_api_all) + ["scenes", "load_obj", "load_obj_with_mtl", "load_stl"]

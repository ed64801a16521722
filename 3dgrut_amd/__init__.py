"""MI355X-native 3DGUT / 3DGRT renderer plugin (drop-in for threedgut_tracer / threedgrt_tracer).

Python host code above a C-ABI HIP library (include/grut_amd.h).  Import via
`importlib.import_module("3dgrut_amd")` or through the shim packages in `shims/`.
"""
from . import _abi  # noqa: F401

__all__ = ["_abi"]

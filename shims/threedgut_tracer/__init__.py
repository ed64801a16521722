"""Shadow of the reference's `threedgut_tracer` package: put `<repo>/shims` ahead of the reference checkout on
PYTHONPATH and `threedgrut/model/model.py:25` (`import threedgut_tracer`) binds to the MI355X renderer unchanged."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
Tracer = _il.import_module("3dgrut_amd.gut_tracer").Tracer

__all__ = ["Tracer"]

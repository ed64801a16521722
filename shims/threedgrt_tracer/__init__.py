"""Shadow of the reference's `threedgrt_tracer` package (see shims/threedgut_tracer/__init__.py):
`threedgrut/model/model.py:23` (`import threedgrt_tracer`) binds to the MI355X software-BVH tracer."""
import importlib as _il
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
Tracer = _il.import_module("3dgrut_amd.grt_tracer").Tracer

__all__ = ["Tracer"]

"""`import gridpp` -> the MI355X implementation of gridpp's optimal-interpolation / neighbourhood path (gridpp_amd).

The reference's SWIG module is `%module gridpp` (swig/gridpp.i:1); with this package on the path a script written for the
reference runs unchanged on the functions gridpp_amd covers.  It never shadows an installed reference:
if another `gridpp` distribution is importable from a different sys.path entry, THAT one is loaded under this name (set
GRIDPP_USE_AMD=1 to take this implementation anyway, e.g. for an A/B run in one environment).
"""
import importlib.machinery as _mach
import importlib.util as _util
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
_other = None
if not _os.environ.get("GRIDPP_USE_AMD"):
    _paths = [p for p in _sys.path if _os.path.abspath(p or _os.getcwd()) != _here]
    try:
        _other = _mach.PathFinder.find_spec("gridpp", _paths)
    except (ImportError, ValueError):
        _other = None

if _other is not None and _other.origin and _os.path.abspath(_other.origin) != _os.path.abspath(__file__):
    _mod = _util.module_from_spec(_other)
    _sys.modules[__name__] = _mod
    _other.loader.exec_module(_mod)
else:
    import gridpp_amd as _impl
    from gridpp_amd import *          # noqa: F401,F403
    # (everything public, also the names `import *` skips when a module defines no __all__)
    globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("_")})
    __version__ = getattr(_impl, "__version__", None)
    implementation = "gridpp_amd"

"""Import shim: makes the `alac.net_amd/` directory importable as the package `alac.net_amd`.

The product package lives in the directory literally named `alac.net_amd/` (a dot is not a
legal Python package name), so this tiny namespace package registers it under that dotted name.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "alac.net_amd")
if "alac.net_amd" not in _sys.modules:
    _spec = _ilu.spec_from_file_location(
        "alac.net_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
    )
    net_amd = _ilu.module_from_spec(_spec)
    _sys.modules["alac.net_amd"] = net_amd
    _spec.loader.exec_module(net_amd)
else:
    net_amd = _sys.modules["alac.net_amd"]

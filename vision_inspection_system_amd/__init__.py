"""Importable alias for the hyphenated package directory ``vision-inspection-system_amd/``.

Python cannot import a directory whose name contains ``-``; this shim makes
``import vision_inspection_system_amd`` resolve submodules from
``vision-inspection-system_amd/`` (the directory the build contract names).
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "vision-inspection-system_amd")
if not _os.path.isdir(_real):  # pragma: no cover
    raise ImportError(f"package directory missing: {_real}")
__path__.insert(0, _real)
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f

"""Importable alias for the product package, whose directory is `gp-nerf_amd/` (a hyphen is not
valid in a Python module name).  `import gp_nerf_amd` executes `gp-nerf_amd/__init__.py` and
resolves sub-modules (`gp_nerf_amd.projection`, ...) from that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gp-nerf_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
